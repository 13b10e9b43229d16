set -e
mkdir -p gpurun_out
for v in ship xb xc xd; do
  if [ $v = ship ]; then unset ZSTDGPU_LIB; else export ZSTDGPU_LIB=$PWD/tools/variants/libzstdgpu_$v.so; fi
  echo "== $v"
  timeout -k 10 150 python -u tools/time_stages.py --config 2 --steps 10
done
