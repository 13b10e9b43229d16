set -e
mkdir -p gpurun_out
for lib in "" tools/variants/libzstdgpu_xxh8.so tools/variants/libzstdgpu_xxh32.so; do
  echo "== lib: ${lib:-default}"
  ZSTDGPU_LIB=$lib timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "xxh64 or checksum or many_frames" 2>&1 | tail -3
  ZSTDGPU_LIB=$lib timeout -k 10 200 python tools/time_stages.py --config 3
done
