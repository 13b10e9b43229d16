set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_tests5.log 2>&1 || { tail -30 gpurun_out/r3_tests5.log; exit 1; }
tail -2 gpurun_out/r3_tests5.log
timeout -k 10 280 python -u tools/fuzz.py --corrupt --rounds 4 --frames 400 --seed 51 > gpurun_out/r3_fuzz5a.log 2>&1 || { tail -20 gpurun_out/r3_fuzz5a.log; exit 1; }
tail -1 gpurun_out/r3_fuzz5a.log
timeout -k 10 280 python -u tools/fuzz.py --dict --corrupt --rounds 4 --frames 400 --seed 52 > gpurun_out/r3_fuzz5b.log 2>&1 || { tail -20 gpurun_out/r3_fuzz5b.log; exit 1; }
tail -1 gpurun_out/r3_fuzz5b.log
timeout -k 10 280 python -u tools/fuzz.py --records --corrupt --rounds 3 --frames 1000 --seed 53 > gpurun_out/r3_fuzz5c.log 2>&1 || { tail -20 gpurun_out/r3_fuzz5c.log; exit 1; }
tail -1 gpurun_out/r3_fuzz5c.log
timeout -k 10 300 python bench.py --config 5 --no-cpu > gpurun_out/r3_bench5.json 2>/dev/null; python3 -c "
import json; j=json.load(open('gpurun_out/r3_bench5.json')); print(j['value'], j['one_shot_latency'])"
