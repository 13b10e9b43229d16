set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "dict or Dict or fuzz or oneshot" > gpurun_out/r3_tests3.log 2>&1 || { tail -30 gpurun_out/r3_tests3.log; exit 1; }
tail -2 gpurun_out/r3_tests3.log
timeout -k 10 200 python tools/time_stages.py --config 5 --steps 10 2>&1 | tail -2
