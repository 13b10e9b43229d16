set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_tests6.log 2>&1 || { tail -30 gpurun_out/r3_tests6.log; exit 1; }
tail -2 gpurun_out/r3_tests6.log
timeout -k 10 280 python -u tools/fuzz.py --corrupt --rounds 4 --frames 400 --seed 61 > gpurun_out/r3_fuzz6a.log 2>&1 || { tail -20 gpurun_out/r3_fuzz6a.log; exit 1; }
tail -1 gpurun_out/r3_fuzz6a.log
timeout -k 10 280 python -u tools/fuzz.py --rounds 3 --frames 400 --seed 62 > gpurun_out/r3_fuzz6b.log 2>&1 || { tail -20 gpurun_out/r3_fuzz6b.log; exit 1; }
tail -1 gpurun_out/r3_fuzz6b.log
for c in 2 3; do echo "config $c"; timeout -k 10 200 python tools/time_stages.py --config $c --steps 10 2>&1 | tail -2; done
