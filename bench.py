#!/usr/bin/env python3
"""Benchmark of the batched MI355X zstd decoder on BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the whole stage chain (literals -> sequences -> layout -> execute -> checksum)
over one batch of synthetic frames, inputs and outputs resident in HBM.  Workload at every N:
config[1] of BASELINE.json per GPU -- 4096 independent level-3 frames of 128 KiB content (one
128 KB block each), half text-like and half binary records, seeded by global frame index.
Frames shard across ranks with no data-path collective (weak scaling: every rank decodes its own
4096 frames); the only collectives are the timing barrier and a max over ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "decompressed GB/s (node) on batched 128KB-block frames; % of HBM roofline"
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak
FRAME_BYTES = 131072


def make_frames(first_seed: int, n: int, level: int):
    """n compressed frames + their plain bytes.  Compressed with the system libzstd when the box has it;
    otherwise the committed 128 KB fixtures are replicated (and the JSON says so)."""
    from zstd_amd import corpus
    if corpus.SystemZstd.available():
        z = corpus.SystemZstd()
        plain = [corpus.mixed(first_seed + i, FRAME_BYTES) for i in range(n)]
        frames = [z.compress(p, level) for p in plain]
        return frames, plain, f"synthetic (seeded generators, compressed with libzstd {z.version} level {level})"
    import glob
    names = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*128k_l3_c0.zst")))
    tmpl = [open(f, "rb").read() for f in names]
    frames = [tmpl[i % len(tmpl)] for i in range(n)]
    return frames, None, f"synthetic (no libzstd on this box: {len(tmpl)} committed level-3 frames replicated x{n // len(tmpl)})"


def cpu_baseline(frames, seconds: float = 12.0, threads: int = 16):
    """The CPU oracle (a C port of the reference's decoder) on this box's host cores, `threads`
    threads over frames, on the same frames for about `seconds` of wall time."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from zstd_amd import api
    L = oracle.lib()
    threads = max(1, min(threads, os.cpu_count() or 1))
    src, fin, dst_bytes = api.pack_frames(frames, [FRAME_BYTES] * len(frames))
    dst = np.zeros(dst_bytes, dtype=np.uint8)
    parts = np.array_split(np.arange(len(frames)), threads)

    def work(idx):
        if len(idx) == 0:
            return 0
        so = np.ascontiguousarray(fin["srcOff"][idx]); ss = np.ascontiguousarray(fin["srcSize"][idx])
        do = np.ascontiguousarray(fin["dstOff"][idx]); dc = np.ascontiguousarray(fin["dstCapacity"][idx])
        r = L.oracle_decompress_batch(dst.ctypes.data, do.ctypes.data, dc.ctypes.data, src.ctypes.data, so.ctypes.data, ss.ctypes.data, len(idx))
        assert not L.oracle_isError(r)
        return r

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, parts))   # warm
        reps, total = 0, 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            total += sum(ex.map(work, parts))
            reps += 1
        dt = time.perf_counter() - t0
    return {"value": round(total / dt / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"{len(frames)} frames of the workload x {reps} passes, {threads} threads over frames, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    import torch
    from zstd_amd import api

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU decode path)"
    # Rehearsal on a box with fewer GPUs than ranks: BENCH_REHEARSAL=1 puts every rank on GPU 0 and uses gloo for the
    # two collectives (RCCL refuses two ranks on one device).  The driver's runs never set it.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = args.frames
    from zstd_amd import sharding
    frames, plain, data_note = make_frames(sharding.shard_seed_range(rank, n).start, n, args.level)
    src, fin, dst_bytes = api.pack_frames(frames, [FRAME_BYTES] * n)
    c_bytes = int(fin["srcSize"].sum())
    d_bytes = n * FRAME_BYTES

    dctx = api.DCtx(local_rank)
    t0 = time.perf_counter()
    batch = api.Batch(dctx, src, fin)
    t_desc_first = time.perf_counter() - t0       # includes the first allocation of the scratch buffers
    batch.close()
    t0 = time.perf_counter()
    batch = api.Batch(dctx, src, fin)             # steady state: the context hands the big buffers over
    t_desc = time.perf_counter() - t0
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    # the same descriptors built by the device kernels from the source in HBM (no host walk over the compressed bytes)
    t_desc_dev = None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bdev = api.Batch(dctx, None, fin, dev_src_ptr=d_src.data_ptr(), dev_src_bytes=src.nbytes)
        t_desc_dev = time.perf_counter() - t0
        assert bdev.descriptor_digest == batch.descriptor_digest
        bdev.close()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        batch.decode(d_src.data_ptr(), d_dst.data_ptr(), stream)
    batch.set_profiling(True)   # HIP events around every kernel, on the launch stream
    stage = np.zeros(api.NUM_STAGES)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.decode(d_src.data_ptr(), d_dst.data_ptr(), stream)
        stage += np.array(batch.stage_ms())      # waits for this step's last event; steps are serial on one stream anyway
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(dist, elapsed, device="cpu" if rehearsal else "cuda")
    stage /= args.steps

    # ---- verification, outside the timed region: verdicts, sizes, and every output byte via XXH64
    st = batch.status()
    assert not st["errorCode"].any(), "decode errors: %s" % st["errorCode"][st["errorCode"] != 0][:8]
    assert (st["regenSize"] == FRAME_BYTES).all()
    if plain is not None:
        d_plain = torch.from_numpy(np.frombuffer(b"".join(plain), dtype=np.uint8).copy()).cuda()
        want = api.xxh64_batch(dctx, d_plain.data_ptr(), np.arange(n, dtype=np.uint64) * FRAME_BYTES, [FRAME_BYTES] * n)
        got = api.xxh64_batch(dctx, d_dst.data_ptr(), fin["dstOff"], st["regenSize"])
        assert np.array_equal(want, got), "output differs from the original corpus"
        probe = [0, n // 2, n - 1]
        host = d_dst.cpu().numpy()
        for i in probe:
            o = int(fin[i]["dstOff"])
            assert host[o:o + FRAME_BYTES].tobytes() == plain[i]
        del d_plain, host

    # ---- PCIe-inclusive figure (reported aside, never the headline): pinned host buffers, H2D + decode + D2H
    e2e = None
    if rank == 0:
        h_src = torch.from_numpy(src).pin_memory()
        h_out = torch.empty(dst_bytes, dtype=torch.uint8, pin_memory=True)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d_src.copy_(h_src, non_blocking=True)
            batch.decode(d_src.data_ptr(), d_dst.data_ptr(), stream)
            h_out.copy_(d_dst, non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        e2e = d_bytes / best / 1e9
        del h_src, h_out

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * d_bytes * args.steps / elapsed / 1e9
    dom = int(np.argmax(stage[:5]))
    dom_ms = float(stage[dom])
    achieved = (c_bytes + d_bytes) / (dom_ms * 1e-3) / 1e9
    dom_name = api.lib().zstdgpu_stage_kernel_name(dom).decode()
    traffic = None     # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (same workload)
    summ = os.path.join(ROOT, "profiles", "r01_summary.json")
    if os.path.exists(summ) and n == 4096 and args.level == 3:
        k = json.load(open(summ))["kernels"].get(dom_name)
        if k:
            traffic = int(k["hbm_traffic_bytes"])
    out = {
        "metric": METRIC, "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": data_note,
        "config": {"workload": f"BASELINE config 2: {n} level-{args.level} frames x 128 KiB content per GPU (one 128 KB block each), "
                               "50% text-like / 50% binary records, no checksum; frames sharded over ranks, no collective",
                   "frames_per_gpu": n, "compressed_bytes_per_gpu": c_bytes, "decompressed_bytes_per_gpu": d_bytes,
                   "parallelism": f"frame-sharded x{world}"},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": c_bytes + d_bytes, "kernel_ms": round(dom_ms, 4)},
        "stage_ms": {api.STAGE_NAMES[i]: round(float(stage[i]), 4) for i in range(api.NUM_STAGES)},
        "whole_chain_frac_of_hbm_peak": round((c_bytes + d_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
        "host_descriptor_build_ms": round(t_desc * 1e3, 2), "device_descriptor_build_ms": round(t_desc_dev * 1e3, 2), "host_descriptor_build_first_ms": round(t_desc_first * 1e3, 2),
        "pcie_inclusive_GBps": round(e2e, 2) if e2e else None,
        "scratch_bytes": batch.scratch_bytes,
    }
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(frames)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
