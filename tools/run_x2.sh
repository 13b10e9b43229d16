set -e
mkdir -p gpurun_out
timeout -k 10 150 python -u tools/time_stages.py --config 2 --steps 10
timeout -k 10 150 python -u tools/time_stages.py --config 5 --steps 10
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/x2_tests.log 2>&1 || { tail -30 gpurun_out/x2_tests.log; exit 1; }
tail -2 gpurun_out/x2_tests.log
