set -e
mkdir -p gpurun_out
timeout -k 10 280 python -u tools/fuzz.py --bufferless --corrupt --rounds 3 --frames 300 --seed 41 > gpurun_out/r3_fuzz_bl_corrupt.log 2>&1 || { tail -20 gpurun_out/r3_fuzz_bl_corrupt.log; exit 1; }
tail -2 gpurun_out/r3_fuzz_bl_corrupt.log
timeout -k 10 280 python -u tools/fuzz.py --bufferless --rounds 3 --frames 300 --seed 42 --max-kb 1500 > gpurun_out/r3_fuzz_bl_valid.log 2>&1 || { tail -20 gpurun_out/r3_fuzz_bl_valid.log; exit 1; }
tail -2 gpurun_out/r3_fuzz_bl_valid.log
timeout -k 10 280 python -u tools/fuzz.py --bufferless --dict --corrupt --rounds 3 --frames 300 --seed 43 > gpurun_out/r3_fuzz_bl_dict.log 2>&1 || { tail -20 gpurun_out/r3_fuzz_bl_dict.log; exit 1; }
tail -2 gpurun_out/r3_fuzz_bl_dict.log
