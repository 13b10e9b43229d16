set -e
mkdir -p gpurun_out
timeout -k 10 200 python -u -m pytest tests/test_gpu_bufferless.py -m gpu -x -q --timeout 120 --timeout-method thread -k "long_frame or misuse or digested" > gpurun_out/r3_bl2.log 2>&1 || { tail -30 gpurun_out/r3_bl2.log; exit 1; }
tail -2 gpurun_out/r3_bl2.log
for c in 2m 3 5 4; do
  timeout -k 10 400 python bench.py --config $c --no-cpu > gpurun_out/r3_final_c$c.json 2> gpurun_out/r3_final_c$c.err || { tail -5 gpurun_out/r3_final_c$c.err; exit 1; }
  python3 -c "
import json; j=json.load(open('gpurun_out/r3_final_c$c.json')); print('$c', j['value'], j['ms_per_step'], j['stage_ms'], j.get('batches_in_flight'))"
done
