set -e
mkdir -p gpurun_out
for n in 1024 2048 4096 8192 16384; do
  timeout -k 10 500 python bench.py --frames $n --no-cpu --no-latency --steps 10 > gpurun_out/r3_frames_$n.json 2> gpurun_out/r3_frames_$n.err || { tail -5 gpurun_out/r3_frames_$n.err; exit 1; }
  python3 -c "
import json; j=json.load(open('gpurun_out/r3_frames_$n.json')); print($n, j['value'], j['ms_per_step'], j['stage_ms'])"
done
