#!/usr/bin/env python3
"""Benchmark of the batched MI355X zstd decoder on BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--config 2|2m|3|4|5]

A step = one pass of the whole stage chain (literals -> sequences -> layout -> execute -> checksum) over one batch
of synthetic frames, inputs and outputs resident in HBM.  Configurations (BASELINE.json `configs`):

  2   (default, the one the metric is quoted on) 4096 level-3 frames of 128 KiB per GPU, one block each, half
      text-like and half binary records, no checksum; weak scaling: every rank decodes its own 4096 frames
  2m  the same bytes as 512 frames of 1 MiB (eight blocks each: tables repeated, window carried across blocks)
  3   4096 level-19 frames of 128 KiB with XXH64 checksums
  4   a template set of mixed frames (text / binary, levels 1-19, checksums on and off) replicated to --total-gb of
      output; strong scaling: the frames are split over the ranks by size (zstd_amd.sharding.partition_frames)
  5   4 KB records compressed with a trained 110 KB dictionary, --frames of them (default 65536)

Frames shard across ranks with no data-path collective; the only collectives are the timing barrier and a max over
ranks.  With --gpus N > 1 and no launcher (WORLD_SIZE unset) the script starts its N ranks itself, before anything
touches a GPU.  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "decompressed GB/s (node) on batched 128KB-block frames; % of HBM roofline"
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak
KERNEL_SOURCES = ("zstd_amd/csrc/kernels.hip", "zstd_amd/csrc/descriptors.hip", "zstd_amd/csrc/zstdgpu_internal.h")


def kernel_source_digest() -> str:
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def host_threads() -> int:
    """Worker threads for host-side work: the CPUs this process may use, at most 16 (a one-GPU box's share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def compress_all(plains, levels, checksums, dictionary=None):
    """Compress with the system libzstd on host threads (ctypes releases the GIL)."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from zstd_amd import corpus
    tls = {}

    def one(i):
        z = tls.setdefault(threading.get_ident(), corpus.SystemZstd())
        return z.compress(plains[i], levels[i], checksum=checksums[i], dictionary=dictionary)
    with ThreadPoolExecutor(host_threads()) as ex:
        return list(ex.map(one, range(len(plains))))


def make_workload(cfg: str, rank: int, world: int, n_frames: int, level: int, total_gb: float):
    """Returns dict(frames, plain, caps, dictionary, note, workload, replicas, scaling)."""
    from zstd_amd import corpus, sharding
    have_z = corpus.SystemZstd.available()
    zver = corpus.SystemZstd().version if have_z else None
    if cfg in ("2", "3"):
        n = n_frames or 4096
        lvl = level or (3 if cfg == "2" else 19)
        ck = cfg == "3"
        seeds = sharding.shard_seed_range(rank, n)
        if have_z:
            plain = [corpus.mixed(s, 131072) for s in seeds]
            frames = compress_all(plain, [lvl] * n, [ck] * n)
            note = f"synthetic (seeded generators, compressed with libzstd {zver} level {lvl})"
        else:
            import glob
            names = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*128k_l3_c0.zst" if cfg == "2" else "*128k_l19_c1.zst")))
            tmpl = [open(f, "rb").read() for f in names]
            frames = [tmpl[i % len(tmpl)] for i in range(n)]
            plain = None
            note = f"synthetic (no libzstd on this box: {len(tmpl)} committed frames replicated x{n // len(tmpl)})"
        what = (f"BASELINE config 2: {n} level-{lvl} frames x 128 KiB content per GPU (one 128 KB block each), 50% text-like / 50% binary records, no checksum"
                if cfg == "2" else
                f"BASELINE config 3: {n} level-{lvl} frames x 128 KiB content per GPU, 50% text-like / 50% binary records, XXH64 checksum on")
        return dict(frames=frames, plain=plain, caps=[131072] * n, dictionary=None, note=note, replicas=1, scaling="weak",
                    workload=what + "; frames sharded over ranks, no collective")
    assert have_z, f"--config {cfg} needs the system libzstd to make its frames"
    if cfg == "2m":
        n = n_frames or 512
        lvl = level or 3
        plain = [corpus.mixed(30_000 + s, 1 << 20) for s in sharding.shard_seed_range(rank, n)]
        frames = compress_all(plain, [lvl] * n, [False] * n)
        return dict(frames=frames, plain=plain, caps=[1 << 20] * n, dictionary=None, replicas=1, scaling="weak",
                    note=f"synthetic (seeded generators, compressed with libzstd {zver} level {lvl})",
                    workload=f"BASELINE config 2, multi-block variant: {n} level-{lvl} frames x 1 MiB per GPU (eight 128 KB blocks each); frames sharded over ranks, no collective")
    if cfg == "4":
        t = n_frames or 4096      # (a launch of 4096 frames fills the chip: 16 blocks per compute unit in every stage)
        lv = [(1, 3, 3, 5, 9, 19)[i % 6] for i in range(t)]
        ck = [bool((i // 6) & 1) for i in range(t)]
        plain_all = [corpus.mixed(40_000 + i, 131072) for i in range(t)]
        frames_all = compress_all(plain_all, lv, ck)
        costs = [len(f) + 131072 for f in frames_all]
        mine = sharding.partition_frames(costs, world)[rank]
        replicas = max(1, int(round(total_gb * 1e9 / (t * 131072))))
        return dict(frames=[frames_all[i] for i in mine], plain=[plain_all[i] for i in mine], caps=[131072] * len(mine), dictionary=None,
                    replicas=replicas, scaling="strong",
                    note=f"synthetic (seeded generators, compressed with libzstd {zver}; {t} distinct frames replicated x{replicas})",
                    workload=f"BASELINE config 4: {t} template frames x 128 KiB (text / binary, levels 1,3,3,5,9,19, checksum on for half) replicated x{replicas} = "
                             f"{t * replicas * 131072 / 1e9:.1f} GB of output, split over {world} rank(s) by size with no collective")
    if cfg == "5":
        n = n_frames or 65536
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_fixtures
        z = corpus.SystemZstd()
        recs = gen_fixtures.records(77 + rank, 1000 + n)
        d = z.train_dictionary(gen_fixtures.records(77, 2000)[:1000], 112640)      # the same dictionary on every rank
        plain = recs[1000:]
        frames = compress_all(plain, [level or 3] * n, [False] * n, dictionary=d)
        return dict(frames=frames, plain=plain, caps=[4096] * n, dictionary=d, replicas=1, scaling="weak",
                    note=f"synthetic (seeded 4 KB records, dictionary trained by ZDICT_trainFromBuffer of libzstd {zver}, level {level or 3})",
                    workload=f"BASELINE config 5: {n} frames of 4 KB records per GPU compressed with a trained {len(d)}-byte dictionary; frames sharded over ranks, no collective")
    raise SystemExit(f"unknown --config {cfg}")


def cpu_baseline(frames, caps, dictionary, budget_s: float = 20.0):
    """The decoders a host could run instead, timed on this box's cores on the same frames: the CPU oracle (a plain C
    restatement of the reference's decoder; the reference itself is Free Pascal and cannot be built here) with one
    thread and with every core this process may use, and the system libzstd the same way when the box has one."""
    from concurrent.futures import ThreadPoolExecutor
    import threading
    from oracle import oracle
    from zstd_amd import api, corpus
    L = oracle.lib()
    threads = host_threads()
    src, fin, dst_bytes = api.pack_frames(frames, caps)
    dst = np.zeros(dst_bytes, dtype=np.uint8)
    d = dictionary or b""

    def oracle_part(idx):
        if len(idx) == 0:
            return 0
        if not d:
            so = np.ascontiguousarray(fin["srcOff"][idx]); ss = np.ascontiguousarray(fin["srcSize"][idx])
            do = np.ascontiguousarray(fin["dstOff"][idx]); dc = np.ascontiguousarray(fin["dstCapacity"][idx])
            r = L.oracle_decompress_batch(dst.ctypes.data, do.ctypes.data, dc.ctypes.data, src.ctypes.data, so.ctypes.data, ss.ctypes.data, len(idx))
            assert not L.oracle_isError(r)
            return r
        total = 0
        for i in idx:
            r = L.oracle_decompress_usingDict(dst.ctypes.data + int(fin["dstOff"][i]), int(fin["dstCapacity"][i]),
                                              src.ctypes.data + int(fin["srcOff"][i]), int(fin["srcSize"][i]), d, len(d))
            assert not L.oracle_isError(r)
            total += r
        return total

    tls = {}

    def zstd_part(idx):
        z = tls.setdefault(threading.get_ident(), corpus.SystemZstd())
        total = 0
        for i in idx:
            sp = src.ctypes.data + int(fin["srcOff"][i]); dp = dst.ctypes.data + int(fin["dstOff"][i])
            if d:
                r = z.lib.ZSTD_decompress_usingDict(z._dctx, dp, int(fin["dstCapacity"][i]), sp, int(fin["srcSize"][i]), d, len(d))
            else:
                r = z.lib.ZSTD_decompressDCtx(z._dctx, dp, int(fin["dstCapacity"][i]), sp, int(fin["srcSize"][i]))
            assert not z.lib.ZSTD_isError(r)
            total += r
        return total

    def timed(fn, nthreads, seconds):
        parts = [p for p in np.array_split(np.arange(len(frames)), nthreads)]
        with ThreadPoolExecutor(nthreads) as ex:
            reps, total = 0, 0
            t0 = time.perf_counter()
            while reps == 0 or time.perf_counter() - t0 < seconds:
                total += sum(ex.map(fn, parts))
                reps += 1
            dt = time.perf_counter() - t0
        return total / dt / 1e9, reps, dt

    legs = [("port", oracle_part)]
    if corpus.SystemZstd.available():
        legs.append(("libzstd", zstd_part))
    per = budget_s / (2 * len(legs))
    # one thread runs over a bounded sample of the frames (about `per` seconds at ~0.5 GB/s), all threads over all of them
    res = {}
    sample1 = max(1, min(len(frames), int(per * 0.5e9 / max(1, sum(caps) / len(caps)))))
    idx1 = np.arange(sample1)
    for kind, fn in legs:
        t0 = time.perf_counter(); reps1 = 0; tot1 = 0
        while reps1 == 0 or time.perf_counter() - t0 < per:
            tot1 += fn(idx1); reps1 += 1
        dt1 = time.perf_counter() - t0
        vN, repsN, dtN = timed(fn, threads, per)
        res[kind] = {"one_thread_GBps": round(tot1 / dt1 / 1e9, 3), "all_threads_GBps": round(vN, 3), "threads": threads,
                     "sample": f"1 thread: first {sample1} frames x {reps1} passes, {dt1:.1f} s; {threads} threads over frames: all {len(frames)} frames x {repsN} passes, {dtN:.1f} s"}
    out = {"value": res["port"]["all_threads_GBps"], "unit": "GB/s", "cores": threads, "kind": "port",
           "sample": res["port"]["sample"], "one_thread_value": res["port"]["one_thread_GBps"], "cpu_model": cpu_model(),
           "cores_of_the_box": os.cpu_count(), "cores_this_process_may_use": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
           "cores_note": "`cores` = worker threads used: the CPUs this process may use, capped at 16 (the share of a one-GPU job on the test boxes)",
           "note": "the oracle is a plain C restatement of the reference's decoder, not a tuned one: the system libzstd (same algorithm, upstream's tuned C) is the figure a host would really get"}
    if "libzstd" in res:
        from zstd_amd import corpus as c2
        out["libzstd"] = dict(res["libzstd"], kind="libzstd", version=c2.SystemZstd().version, value=res["libzstd"]["all_threads_GBps"], cores=threads, unit="GB/s")
    return out


def one_shot_latency(dctx, calls: int = 200):
    """Per-call cost of the one-shot drop-ins on a small frame: a committed 4 KB record compressed with the committed 16 KB
    dictionary, host buffers in and out.  `rebuilt`: zstdgpu_decompress_usingDDict as a caller of the reference's
    ZSTD_decompress_usingDDict would use it (frame walk, a batch made and destroyed, two copies, every call).  `reused`:
    the same frame as a batch prepared once, decoded on device buffers and its verdict read back, every call."""
    import torch
    from zstd_amd import api
    gold = os.path.join(ROOT, "tests", "golden")
    manifest = json.load(open(os.path.join(gold, "manifest.json")))
    e = {x["name"]: x for x in manifest["frames"]}["rec00_dict"]
    fr = open(os.path.join(gold, e["file"]), "rb").read()
    dd = api.DDict(dctx, open(os.path.join(gold, manifest["dicts"]["dict16k"]["file"]), "rb").read())
    want = api.ZSTD_decompress_usingDDict(dctx, fr, e["capacity"], dd)
    assert len(want) == e["capacity"]
    t0 = time.perf_counter()
    for _ in range(calls):
        api.ZSTD_decompress_usingDDict(dctx, fr, e["capacity"], dd)
    rebuilt = (time.perf_counter() - t0) / calls
    src, fin, dst_bytes = api.pack_frames([fr], [e["capacity"]])
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
    b = api.Batch(dctx, src, fin, dd)
    stream = torch.cuda.current_stream().cuda_stream
    b.decode(d_src.data_ptr(), d_dst.data_ptr(), stream)
    assert int(b.status()[0]["errorCode"]) == 0 and d_dst.cpu().numpy().tobytes()[: len(want)] == want
    t0 = time.perf_counter()
    for _ in range(calls):
        b.decode(d_src.data_ptr(), d_dst.data_ptr(), stream)
        b.status()
    reused = (time.perf_counter() - t0) / calls
    b.close()
    # the entry point that allocates nothing, recorded once into a HIP graph and replayed: device walk + stage chain as one
    # graph launch, the verdict read back from device memory after every replay
    d_fin = torch.from_numpy(fin.view(np.uint8).copy()).cuda()
    d_st = torch.zeros(api.FRAME_STATUS_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_dst.zero_()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        api.decode_frames_noalloc(dctx, d_src.data_ptr(), src.nbytes, d_fin.data_ptr(), 1, d_dst.data_ptr(), d_st.data_ptr(), dd, torch.cuda.current_stream().cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    assert int(d_st.cpu().numpy().view(api.FRAME_STATUS_DTYPE)[0]["errorCode"]) == 0 and d_dst.cpu().numpy().tobytes()[: len(want)] == want
    t0 = time.perf_counter()
    for _ in range(calls):
        g.replay()
        d_st.cpu()
    graph = (time.perf_counter() - t0) / calls
    del g
    dd.close()
    return {"frame": "tests/golden rec00_dict (4096 B record, 774 B compressed, 16 KB dictionary)", "calls": calls,
            "decompress_usingDDict_us_per_call": round(rebuilt * 1e6, 1), "prepared_batch_decode_and_status_us_per_call": round(reused * 1e6, 1),
            "noalloc_graph_replay_and_status_us_per_call": round(graph * 1e6, 1)}


def bufferless_latency(dctx):
    """Per-call cost of the bufferless interface (zstdgpu_decompressContinue) on a committed frame of eight full blocks
    (1 MiB of text, level 1), host buffers in and out: every compressed block is one pass of the stage chain."""
    import ctypes
    from zstd_amd import api
    gold = os.path.join(ROOT, "tests", "golden")
    e = {x["name"]: x for x in json.load(open(os.path.join(gold, "manifest.json")))["frames"]}["text1m_l1"]
    fr = open(os.path.join(gold, e["file"]), "rb").read()
    buf = ctypes.create_string_buffer(e["capacity"])
    base = ctypes.addressof(buf)

    def once():
        assert not api.ZSTD_isError(dctx.decompress_begin())
        pos = op = blocks = 0
        t_blocks = 0.0
        while True:
            n = dctx.next_src_size_to_decompress()
            if n == 0:
                break
            is_block = dctx.next_input_type() in (2, 3)
            t0 = time.perf_counter()
            r = dctx.decompress_continue(base + op, e["capacity"] - op, fr[pos:pos + n])
            if is_block:
                t_blocks += time.perf_counter() - t0
                blocks += 1
            assert not api.ZSTD_isError(r)
            pos += n
            op += r
        return op, blocks, t_blocks

    once()
    t0 = time.perf_counter()
    reps, tb, nb, size = 5, 0.0, 0, 0
    for _ in range(reps):
        size, b, t = once()
        tb += t
        nb += b
    whole = (time.perf_counter() - t0) / reps
    assert size == e["size"]
    return {"frame": "tests/golden text1m_l1 (1 MiB, eight blocks)", "us_per_block_call": round(tb / nb * 1e6, 1),
            "frame_ms": round(whole * 1e3, 2), "GBps": round(size / whole / 1e9, 3)}


def spawn_ranks(args) -> int:
    """--gpus N > 1 without a launcher: start the N ranks as fresh processes (nothing here has touched a GPU)."""
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2", choices=["2", "2m", "3", "4", "5"])
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (config 4: template frames in total); 0 = the configuration's own count")
    ap.add_argument("--level", type=int, default=0, help="compression level; 0 = the configuration's own")
    ap.add_argument("--total-gb", type=float, default=64.0, help="config 4: output bytes of the whole job, in GB")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-shot latency leg (profiling runs: its hundreds of tiny launches would swamp the per-kernel averages)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    from zstd_amd import api, sharding

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

    w = make_workload(args.config, rank, world, args.frames, args.level, args.total_gb)
    frames, plain, caps, replicas = w["frames"], w["plain"], w["caps"], w["replicas"]
    n = len(frames)
    src, fin, dst_bytes = api.pack_frames(frames, caps)
    c_bytes = int(fin["srcSize"].sum()) * replicas
    d_bytes = int(sum(caps)) * replicas
    rep_stride = (dst_bytes + 255) & ~255

    dctx = api.DCtx(local_rank)
    ddict = api.DDict(dctx, w["dictionary"]) if w["dictionary"] is not None else None
    t0 = time.perf_counter()
    batch = api.Batch(dctx, src, fin, ddict)
    t_desc_first = time.perf_counter() - t0       # includes the first allocation of the scratch buffers
    batch.close()
    t0 = time.perf_counter()
    batch = api.Batch(dctx, src, fin, ddict)      # steady state: the context hands the big buffers over
    t_desc = time.perf_counter() - t0
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(rep_stride * replicas, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    # the same descriptors built by the device kernels from the source in HBM (no host walk over the compressed bytes)
    t_desc_dev = None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bdev = api.Batch(dctx, None, fin, ddict, dev_src_ptr=d_src.data_ptr(), dev_src_bytes=src.nbytes)
        t_desc_dev = time.perf_counter() - t0
        assert bdev.descriptor_digest == batch.descriptor_digest
        bdev.close()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def one_step():
        for r in range(replicas):
            batch.decode(d_src.data_ptr(), d_dst.data_ptr() + r * rep_stride, stream)

    for _ in range(args.warmup):
        one_step()
    batch.set_profiling(True)   # HIP events around every kernel, on the launch stream
    stage = np.zeros(api.NUM_STAGES)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
        stage += np.array(batch.stage_ms())      # waits for this step's last launch; launches are serial on one stream anyway
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(dist, elapsed, device="cpu" if rehearsal else "cuda")
    stage /= args.steps                           # per launch (config 4: the last replica's launch of every step)

    # ---- verification, outside the timed region: verdicts, sizes, and every output byte via XXH64
    st = batch.status()
    assert not st["errorCode"].any(), "decode errors: %s" % st["errorCode"][st["errorCode"] != 0][:8]
    assert (st["regenSize"] == np.array(caps, dtype=np.uint64)).all()
    if plain is not None:
        sizes = np.array([len(p) for p in plain], dtype=np.uint64)
        d_plain = torch.from_numpy(np.frombuffer(b"".join(plain), dtype=np.uint8).copy()).cuda()
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        want = api.xxh64_batch(dctx, d_plain.data_ptr(), offs, sizes)
        for r in sorted({0, replicas - 1}):
            got = api.xxh64_batch(dctx, d_dst.data_ptr() + r * rep_stride, fin["dstOff"], st["regenSize"])
            assert np.array_equal(want, got), "output differs from the original corpus"
        host = d_dst[:dst_bytes].cpu().numpy()
        for i in (0, n // 2, n - 1):
            o = int(fin[i]["dstOff"])
            assert host[o:o + len(plain[i])].tobytes() == plain[i]
        del d_plain, host

    # ---- several batches in flight (reported aside, never the headline): the same frames as two / three batch objects with
    # their own scratch and output, launched in turn on streams of their own, so that the compute units a kernel's short
    # chains leave early are taken by another batch's kernels instead of idling until its longest chain ends
    two = None
    if replicas == 1 and not args.no_latency:
        two = {}
        extra = [(api.Batch(dctx, src, fin, ddict), torch.zeros(rep_stride, dtype=torch.uint8, device="cuda"), torch.cuda.Stream()) for _ in range(2)]
        for nfl in (2, 3):
            lanes = [(batch, d_dst, None)] + extra[: nfl - 1]

            def launch_all():
                for b_, dst_, s_ in lanes:
                    b_.decode(d_src.data_ptr(), dst_.data_ptr(), s_.cuda_stream if s_ is not None else stream)
            rounds = max(2, args.steps // nfl)
            for _ in range(2):
                launch_all()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(rounds):
                launch_all()
            torch.cuda.synchronize()
            dtn = time.perf_counter() - t0
            for b_, dst_, s_ in lanes[1:]:
                assert not b_.status()["errorCode"].any() and torch.equal(d_dst[:dst_bytes], dst_[:dst_bytes])
            two[f"{nfl}_batches"] = {"GBps": round(nfl * rounds * d_bytes / dtn / 1e9, 2), "launches": nfl * rounds, "ms_per_launch": round(dtn / (nfl * rounds) * 1e3, 4)}
        for b_, dst_, s_ in extra:
            b_.close()
        del extra

    # ---- PCIe-inclusive figure (reported aside, never the headline): pinned host buffers, H2D + decode + D2H
    e2e = e2e_pipe = None
    if rank == 0 and replicas == 1:
        h_src = torch.from_numpy(src).pin_memory()
        h_out = torch.empty(dst_bytes, dtype=torch.uint8, pin_memory=True)
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d_src.copy_(h_src, non_blocking=True)
            batch.decode(d_src.data_ptr(), d_dst.data_ptr(), stream)
            h_out.copy_(d_dst[:dst_bytes], non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        e2e = d_bytes / best / 1e9
        # the same with three batches in flight on streams of their own, each with its own device and pinned host buffers:
        # one batch's upload and another's download run beside the decode of a third (what a caller with host data would do)
        if not args.no_latency:
            slots = []
            for _ in range(3):
                s_ = torch.cuda.Stream()
                slots.append((api.Batch(dctx, src, fin, ddict), torch.empty_like(d_src), torch.empty(rep_stride, dtype=torch.uint8, device="cuda"),
                              torch.empty(dst_bytes, dtype=torch.uint8, pin_memory=True), s_))

            def pipe_round():
                for b_, ds_, dd_, ho_, s_ in slots:
                    with torch.cuda.stream(s_):
                        ds_.copy_(h_src, non_blocking=True)
                        b_.decode(ds_.data_ptr(), dd_.data_ptr(), s_.cuda_stream)
                        ho_.copy_(dd_[:dst_bytes], non_blocking=True)
            pipe_round()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                pipe_round()
            torch.cuda.synchronize()
            e2e_pipe = 9 * d_bytes / (time.perf_counter() - t0) / 1e9
            assert torch.equal(slots[2][3], h_out)
            for b_, *_ in slots:
                b_.close()
            del slots
        del h_src, h_out

    # job totals: under weak scaling every rank carries the same load; under strong scaling the shards differ by a frame or so
    totals = torch.tensor([float(c_bytes), float(d_bytes)], dtype=torch.float64, device="cpu" if (rehearsal or dist is None) else "cuda")
    if dist is not None:
        dist.all_reduce(totals)
    job_c, job_d = float(totals[0]), float(totals[1])
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = job_d * args.steps / elapsed / 1e9
    dom = int(np.argmax(stage[:5]))
    dom_ms = float(stage[dom])
    launch_cd = (c_bytes + d_bytes) / replicas            # algorithmic bytes of ONE launch on this rank
    achieved = launch_cd / (dom_ms * 1e-3) / 1e9
    dom_name = api.lib().zstdgpu_stage_kernel_name(dom).decode()
    if dom_name == "k_execute" and n <= 1024:
        dom_name = "k_execute_wf"     # (batches of up to 1024 frames of many sequences: the eight-wave kernel)
    # HBM bytes per launch of that kernel from the rocprofv3 PMC passes -- only when the committed summary was taken on this
    # configuration from these very kernel sources; a profile of other code says nothing about this run
    traffic, traffic_from = None, None
    digest = kernel_source_digest()
    for tag in ("r03", "r02"):
        summ = os.path.join(ROOT, "profiles", f"{tag}_config{args.config}_summary.json")
        if os.path.exists(summ) and not args.frames and not args.level:
            js = json.load(open(summ))
            if dom_name == "k_execute":       # the execute stage is two kernels, of which one does the work of a uniform batch
                dom_name = max(("k_execute", "k_execute_wf"), key=lambda kn: js["kernels"].get(kn, {}).get("avg_ns", 0.0))
            k = js["kernels"].get(dom_name)
            if k and js.get("kernel_source_digest") == digest:
                traffic = int(k["hbm_traffic_bytes"])
                traffic_from = os.path.relpath(summ, ROOT)
    out = {
        "metric": METRIC, "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": w["scaling"], "vs_baseline": None,
        "dtype": "u8", "data": w["note"],
        "config": {"workload": w["workload"], "frames_on_rank0": n, "replicas_per_step": replicas,
                   "compressed_bytes_job": int(job_c), "decompressed_bytes_job": int(job_d),
                   "parallelism": f"frame-sharded x{world}"},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_from": traffic_from,
                     "algorithmic_bytes_per_launch": int(launch_cd), "kernel_ms": round(dom_ms, 4)},
        "stage_ms": {api.STAGE_NAMES[i]: round(float(stage[i]), 4) for i in range(api.NUM_STAGES)},
        "whole_chain_frac_of_hbm_peak": round((job_c + job_d) / world / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
        "host_descriptor_build_ms": round(t_desc * 1e3, 2), "device_descriptor_build_ms": round(t_desc_dev * 1e3, 2), "host_descriptor_build_first_ms": round(t_desc_first * 1e3, 2),
        "pcie_inclusive_GBps": round(e2e, 2) if e2e else None, "pcie_inclusive_3_in_flight_GBps": round(e2e_pipe, 2) if e2e_pipe else None,
        "batches_in_flight": two,
        "scratch_bytes": batch.scratch_bytes, "kernel_source_digest": digest,
    }
    if world == 1 and not args.no_latency:
        out["one_shot_latency"] = one_shot_latency(dctx)
        out["bufferless"] = bufferless_latency(dctx)
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(frames, caps, w["dictionary"])
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
