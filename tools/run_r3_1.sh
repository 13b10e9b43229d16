set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_tests1.log 2>&1 || { tail -30 gpurun_out/r3_tests1.log; exit 1; }
tail -3 gpurun_out/r3_tests1.log
timeout -k 10 200 python bench.py --no-cpu > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err
cat gpurun_out/r3_bench0.json
for v in A B C D; do
  echo "variant $v"
  ZSTDGPU_LIB=tools/variants/libzstdgpu_$v.so timeout -k 10 120 python tools/time_stages.py --config 2 --steps 10 2>&1 | tail -2
done
echo base
timeout -k 10 120 python tools/time_stages.py --config 2 --steps 10 2>&1 | tail -2
