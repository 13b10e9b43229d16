#!/bin/bash
# Instruction mix per kernel (one rocprofv3 counter pass): tools/instmix.sh <tag> -> gpurun_out/mix_<tag>.txt
set -u
tag=${1:-x}
out=gpurun_out/mix_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $out/pmc.log 2>&1 || exit 1
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
with open(out + ".txt", "w") as fo:
    for k in acc:
        n = max(cnt[k], 1)
        line = k + " launches=%d " % n + " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(acc[k].items()))
        print(line); fo.write(line + "\n")
PY
