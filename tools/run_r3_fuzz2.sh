set -e
mkdir -p gpurun_out
timeout -k 10 280 python -u tools/fuzz.py --records --corrupt --rounds 6 --frames 1000 --seed 41 > gpurun_out/r3_fuzz_rec_corrupt.log 2>&1 || { tail -20 gpurun_out/r3_fuzz_rec_corrupt.log; exit 1; }
tail -2 gpurun_out/r3_fuzz_rec_corrupt.log
timeout -k 10 200 python -u tools/fuzz.py --records --rounds 4 --frames 4000 --seed 42 > gpurun_out/r3_fuzz_rec.log 2>&1 || { tail -20 gpurun_out/r3_fuzz_rec.log; exit 1; }
tail -2 gpurun_out/r3_fuzz_rec.log
