set -e
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 290 python -u tools/fuzz.py "$@" > gpurun_out/r3_soak_$name.log 2>&1 || { tail -20 gpurun_out/r3_soak_$name.log; exit 1; }; echo "$name: $(tail -1 gpurun_out/r3_soak_$name.log) $(grep -c '^round' gpurun_out/r3_soak_$name.log) rounds"; }
run valid --rounds 20 --frames 400 --seed 101
run corrupt --corrupt --rounds 12 --frames 400 --seed 102
run dict --dict --rounds 12 --frames 400 --seed 103
run dictcorrupt --dict --corrupt --rounds 12 --frames 400 --seed 104
run rec --records --rounds 6 --frames 4000 --seed 105
run reccorrupt --records --corrupt --rounds 8 --frames 1000 --seed 106
