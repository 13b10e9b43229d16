set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_tests_final.log 2>&1 || { tail -30 gpurun_out/r3_tests_final.log; exit 1; }
tail -2 gpurun_out/r3_tests_final.log
timeout -k 10 100 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err
python3 -c "
import json; j=json.load(open('gpurun_out/r3_bench_final.json')); print(j['value'], j['ms_per_step'], j['roofline'], j['batches_in_flight'], j['one_shot_latency'], j['cpu_baseline']['value'], j['cpu_baseline']['libzstd']['value'])"
