set -e
mkdir -p gpurun_out
for c in 2 2m 3 5; do timeout -k 10 400 python bench.py --config $c --no-cpu > gpurun_out/r3_bench_c$c.json 2> gpurun_out/r3_bench_c$c.err; python3 -c "
import json; j=json.load(open('gpurun_out/r3_bench_c$c.json')); print('$c', j['value'], j['ms_per_step'], j['stage_ms'], j.get('one_shot_latency'))"; done
timeout -k 10 600 python bench.py --config 4 --no-cpu --steps 5 --warmup 1 > gpurun_out/r3_bench_c4.json 2> gpurun_out/r3_bench_c4.err; python3 -c "
import json; j=json.load(open('gpurun_out/r3_bench_c4.json')); print('4', j['value'], j['ms_per_step'], j['stage_ms'])"
