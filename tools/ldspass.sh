cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ldsp
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/ldsp -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-latency > gpurun_out/ldsp.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/ldsp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if not m: continue
        acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_LDS": cnt[m.group(1)] += 1
for k in ("k_literals", "k_sequences", "k_execute"):
    n = max(cnt[k], 1)
    print(k, " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(acc[k].items())))
PY
