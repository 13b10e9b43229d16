set -e
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 290 python -u tools/fuzz.py "$@" > gpurun_out/r3_soak2_$name.log 2>&1 || { tail -20 gpurun_out/r3_soak2_$name.log; exit 1; }; echo "$name: $(tail -1 gpurun_out/r3_soak2_$name.log) $(grep -c '^round' gpurun_out/r3_soak2_$name.log) rounds"; }
run blvalid --bufferless --rounds 10 --frames 400 --seed 201 --max-kb 1500
run blcorrupt --bufferless --corrupt --rounds 10 --frames 400 --seed 202
run bldict --bufferless --dict --rounds 10 --frames 400 --seed 203
run bldictcorrupt --bufferless --dict --corrupt --rounds 10 --frames 400 --seed 204
run valid --rounds 16 --frames 400 --seed 205
run corrupt --corrupt --rounds 10 --frames 400 --seed 206
run reccorrupt --records --corrupt --rounds 6 --frames 1000 --seed 207
