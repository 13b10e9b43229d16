set -e
mkdir -p gpurun_out
timeout -k 10 280 python -u tools/fuzz.py --dict --corrupt --rounds 6 --frames 400 --seed 31 > gpurun_out/r3_fuzz_dict_corrupt.log 2>&1 || { tail -20 gpurun_out/r3_fuzz_dict_corrupt.log; exit 1; }
tail -2 gpurun_out/r3_fuzz_dict_corrupt.log
timeout -k 10 200 python -u tools/fuzz.py --dict --rounds 5 --frames 400 --seed 32 > gpurun_out/r3_fuzz_dict.log 2>&1 || { tail -20 gpurun_out/r3_fuzz_dict.log; exit 1; }
tail -2 gpurun_out/r3_fuzz_dict.log
timeout -k 10 280 python -u tools/fuzz.py --corrupt --rounds 5 --frames 400 --seed 33 > gpurun_out/r3_fuzz_corrupt.log 2>&1 || { tail -20 gpurun_out/r3_fuzz_corrupt.log; exit 1; }
tail -2 gpurun_out/r3_fuzz_corrupt.log
timeout -k 10 200 python -u tools/fuzz.py --rounds 4 --frames 400 --seed 34 > gpurun_out/r3_fuzz_valid.log 2>&1 || { tail -20 gpurun_out/r3_fuzz_valid.log; exit 1; }
tail -2 gpurun_out/r3_fuzz_valid.log
