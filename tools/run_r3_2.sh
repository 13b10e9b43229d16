set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_tests2.log 2>&1 || { tail -30 gpurun_out/r3_tests2.log; exit 1; }
tail -3 gpurun_out/r3_tests2.log
echo base; timeout -k 10 120 python tools/time_stages.py --config 2 --steps 10 2>&1 | tail -2
echo cw4; ZSTDGPU_LIB=tools/variants/libzstdgpu_cw4.so timeout -k 10 120 python tools/time_stages.py --config 2 --steps 10 2>&1 | tail -2
ZSTDGPU_LIB=tools/variants/libzstdgpu_cw4.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "batch or stage or fixture or many" > gpurun_out/r3_tests2_cw4.log 2>&1 || { tail -30 gpurun_out/r3_tests2_cw4.log; exit 1; }
tail -3 gpurun_out/r3_tests2_cw4.log
