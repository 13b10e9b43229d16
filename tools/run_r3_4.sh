set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_tests4.log 2>&1 || { tail -30 gpurun_out/r3_tests4.log; exit 1; }
tail -2 gpurun_out/r3_tests4.log
for c in 2 5 3 2m; do echo "config $c"; timeout -k 10 200 python tools/time_stages.py --config $c --steps 10 2>&1 | tail -2; done
