set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_tests_7.log 2>&1 || { tail -30 gpurun_out/r3_tests_7.log; exit 1; }
tail -2 gpurun_out/r3_tests_7.log
bash tools/run_r3_prof.sh
