#!/bin/bash
# Per-kernel average durations (rocprofv3 kernel trace): tools/kstats.sh -> stdout
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kstats
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-latency > gpurun_out/kstats.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/kstats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %4s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
