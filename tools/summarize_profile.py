#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into profiles/<tag>_summary.json and a readable .txt.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced streaming reads (half the bytes); the kernels here read narrow and
unaligned, which is an uncalibrated pattern, so both the raw and the doubled read figure are kept
and the doubled one (upper bound) is what bench.py reports as traffic."""
import collections
import csv
import re
import glob
import json
import os
import sys


def per_kernel(path, pick):
    rows = list(csv.DictReader(open(max(glob.glob(os.path.join(path, "*", "*_counter_collection.csv")), key=os.path.getmtime))))   # newest run
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "zg::" in r["Kernel_Name"]:
            name = re.search(r"(k_\w+)", r["Kernel_Name"]).group(1)
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items() if c in pick or not pick} for k, d in agg.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    cfg = sys.argv[3] if len(sys.argv) > 3 else "2"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench      # the digest of the kernel sources: bench.py takes HBM traffic from a summary only when it matches
    out = {"tag": tag, "config": cfg, "command": f"python3 bench.py --config {cfg} --steps 5 --warmup 2 --no-cpu --no-latency",
           "kernel_source_digest": bench.kernel_source_digest(), "kernels": {}}
    stats = list(csv.DictReader(open(max(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime))))
    for r in stats:
        if "zg::" in r["Name"]:
            name = re.search(r"(k_\w+)", r["Name"]).group(1)
            out["kernels"][name] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                    "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
    fetch = per_kernel(os.path.join(src, "fetch"), {"FETCH_SIZE"})
    write = per_kernel(os.path.join(src, "write"), {"WRITE_SIZE"})
    lds = per_kernel(os.path.join(src, "lds"), set())
    for k, d in out["kernels"].items():
        f = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024
        w = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
        d["fetch_bytes_raw"] = f
        d["fetch_bytes_x2"] = 2 * f
        d["write_bytes"] = w
        d["hbm_traffic_bytes"] = 2 * f + w
        c = lds.get(k, {})
        if c.get("SQ_WAVES"):
            d["waves"] = c["SQ_WAVES"]
            d["lds_bank_conflict_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0)
            d["lds_active_cycles"] = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
            d["wave_cycles_x4"] = c.get("SQ_WAVE_CYCLES", 0.0) * 4
            d["busy_cycles"] = c.get("SQ_BUSY_CYCLES", 0.0)
            wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            d["share_active"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
            d["share_wait_memory_or_lds"] = c.get("SQ_WAIT_ANY", 0.0) / wc
            d["share_issue_stall"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wc
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    with open(os.path.join(root, "profiles", f"{tag}_summary.json"), "w") as fjs:
        json.dump(out, fjs, indent=1)
    with open(os.path.join(root, "profiles", f"{tag}_summary.txt"), "w") as ft:
        ft.write(f"rocprofv3 summary {tag}: {out['command']} (1 x MI355X), kernel sources {out['kernel_source_digest']}\n")
        ft.write("kernel            calls   avg_ms   fetch_MB(raw)  fetch_MB(x2)  write_MB  lds_conflict/lds_active  active  wait  stall\n")
        for k, d in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["avg_ns"]):
            ft.write("%-16s %6d %8.4f %13.1f %13.1f %9.1f   %10.0f/%-12.0f %6.2f %5.2f %5.2f\n" % (
                k, d["calls"], d["avg_ns"] / 1e6, d["fetch_bytes_raw"] / 1e6, d["fetch_bytes_x2"] / 1e6, d["write_bytes"] / 1e6,
                d.get("lds_bank_conflict_cycles", 0), d.get("lds_active_cycles", 0), d.get("share_active", 0),
                d.get("share_wait_memory_or_lds", 0), d.get("share_issue_stall", 0)))
    print(open(os.path.join(root, "profiles", f"{tag}_summary.txt")).read())


if __name__ == "__main__":
    main()
