set -e
for c in 2 5 3 2m; do
  bash tools/profile.sh r03_config$c $c
  python3 tools/summarize_profile.py gpurun_out/prof_r03_config$c r03_config$c $c > gpurun_out/r03_config${c}_summary.out 2>&1
  cp profiles/r03_config${c}_summary.json profiles/r03_config${c}_summary.txt gpurun_out/
  tail -14 profiles/r03_config${c}_summary.txt
done
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err
cat gpurun_out/r3_bench1.json
