#!/bin/bash
# Waves in flight per kernel: tools/occ_pass.sh <config>  (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES, one pass)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cfg=${1:-5}
rm -rf gpurun_out/occ; mkdir -p gpurun_out/occ
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/occ -- python3 tools/time_stages.py --config $cfg --steps 2 > gpurun_out/occ.log 2>&1 || { tail -5 gpurun_out/occ.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/occ/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
