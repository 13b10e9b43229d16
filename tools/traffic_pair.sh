#!/bin/bash
# HBM-side traffic of the execute kernel with and without its scattered match-source fetches, same session:
#   tools/traffic_pair.sh <library built with -DZG_X_NOFAR>   -> stdout: per build FETCH_SIZE / WRITE_SIZE / duration of k_execute
# (tools/time_stages.py decodes config 2 without checking the output, which the knock-out build gets wrong on purpose)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
nofar=$1
for build in shipping nofar; do
  if [ $build = nofar ]; then export ZSTDGPU_LIB=$nofar; else unset ZSTDGPU_LIB; fi
  out=gpurun_out/traffic_$build
  rm -rf $out; mkdir -p $out
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/time_stages.py --config 2 --steps 3 > $out/stats.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/time_stages.py --config 2 --steps 3 > $out/fetch.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/time_stages.py --config 2 --steps 3 > $out/write.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/req -- python3 tools/time_stages.py --config 2 --steps 3 > $out/req.log 2>&1 || echo "(request-size pass not available)"
done
python3 - <<'PY'
import csv, glob, re, collections
for build in ("shipping", "nofar"):
    base = "gpurun_out/traffic_" + build
    line = {}
    for f in glob.glob(base + "/stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_execute" in r["Name"] and "wf" not in r["Name"]:
                line["avg_ms"] = float(r["AverageNs"]) / 1e6
    for sub in ("fetch", "write", "req"):
        acc = collections.defaultdict(list)
        for f in glob.glob(base + "/" + sub + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_execute" in r["Kernel_Name"] and "wf" not in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in acc.items():
            line[c] = sum(v) / len(v)
    print(build, " ".join("%s=%.6g" % kv for kv in sorted(line.items())))
PY
