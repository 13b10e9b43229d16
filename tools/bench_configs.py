#!/usr/bin/env python3
"""Side measurements on BASELINE.json's other configurations (not the bench.py contract line):
   C2t / C2b  the benchmark shape (4096 level-3 frames of 128 KiB) on one corpus only: text-like, binary records
   C2m  the same shape as 512 frames of eight blocks (1 MiB each): tables repeated and windows carried across blocks
   C3  level-19 frames of 128 KiB with checksums
   C5  4 KB records compressed with a trained dictionary, many small frames
Prints one line per configuration: decompressed GB/s and per-stage milliseconds."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name, frames, capacities, plain, dictionary=None, steps=5):
    import torch
    from zstd_amd import api
    n = len(frames)
    src, fin, dst_bytes = api.pack_frames(frames, capacities)
    dctx = api.DCtx(0)
    ddict = api.DDict(dctx, dictionary) if dictionary is not None else None
    t0 = time.perf_counter()
    batch = api.Batch(dctx, src, fin, ddict)
    t_first = time.perf_counter() - t0
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    batch.close()
    t0 = time.perf_counter()
    batch = api.Batch(dctx, src, fin, ddict)      # steady state: the context hands the big buffers over
    t_host = time.perf_counter() - t0
    t_dev = 0.0
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bdev = api.Batch(dctx, None, fin, ddict, dev_src_ptr=d_src.data_ptr(), dev_src_bytes=src.nbytes)
        t_dev = time.perf_counter() - t0
        assert bdev.descriptor_digest == batch.descriptor_digest
        bdev.close()
    for _ in range(2):
        batch.decode(d_src.data_ptr(), d_dst.data_ptr(), stream)
    batch.set_profiling(True)
    torch.cuda.synchronize()
    stage = np.zeros(api.NUM_STAGES)
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.decode(d_src.data_ptr(), d_dst.data_ptr(), stream)
        stage += np.array(batch.stage_ms())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = batch.status()
    assert not st["errorCode"].any(), st["errorCode"][st["errorCode"] != 0][:8]
    sizes = np.array([len(p) for p in plain], dtype=np.uint64)
    assert (st["regenSize"] == sizes).all()
    d_plain = torch.from_numpy(np.frombuffer(b"".join(plain), dtype=np.uint8).copy()).cuda()
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    want = api.xxh64_batch(dctx, d_plain.data_ptr(), offs, sizes)
    got = api.xxh64_batch(dctx, d_dst.data_ptr(), fin["dstOff"], st["regenSize"])
    assert np.array_equal(want, got), "output differs from the original"
    d = int(sizes.sum())
    stage /= steps
    print(f"{name}: {n} frames, {d / 1e6:.1f} MB out, {int(fin['srcSize'].sum()) / 1e6:.1f} MB in: {d / dt / 1e9:.2f} GB/s "
          f"({dt * 1e3:.3f} ms/step; " + " ".join(f"{api.STAGE_NAMES[i]} {stage[i]:.3f}" for i in range(api.NUM_STAGES)) +
          f"; descriptors: host walk {t_host * 1e3:.2f} ms (first {t_first * 1e3:.1f}), device walk {t_dev * 1e3:.2f} ms)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c2-frames", type=int, default=4096)
    ap.add_argument("--c3-frames", type=int, default=1024)
    ap.add_argument("--c5-frames", type=int, default=65536)
    args = ap.parse_args()
    from zstd_amd import corpus
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_fixtures
    z = corpus.SystemZstd()
    if args.c2_frames:
        from concurrent.futures import ThreadPoolExecutor
        import threading
        tls = {}

        def comp(pl):
            return tls.setdefault(threading.get_ident(), corpus.SystemZstd()).compress(pl, 3)
        for name, gen in (("C2t text-like only", corpus.text_like), ("C2b binary records only", corpus.binary_records)):
            plain = [gen(20_000 + i, 131072) for i in range(args.c2_frames)]
            with ThreadPoolExecutor(16) as ex:
                frames = list(ex.map(comp, plain))
            run(name, frames, [131072] * len(frames), plain)
        plain = [corpus.mixed(30_000 + i, 1 << 20) for i in range(args.c2_frames // 8)]
        with ThreadPoolExecutor(16) as ex:
            frames = list(ex.map(comp, plain))
        run("C2m 1 MiB frames of eight blocks", frames, [1 << 20] * len(frames), plain)
    plain = [corpus.mixed(10_000 + i, 131072) for i in range(args.c3_frames)]
    frames = [z.compress(p, 19, True) for p in plain]
    run("C3 level 19 + checksum", frames, [131072] * len(frames), plain)
    recs = gen_fixtures.records(77, 1000 + 4096)
    d = z.train_dictionary(recs[:1000], 112640)
    base = [z.compress(r, 3, dictionary=d) for r in recs[1000:]]
    rep = (args.c5_frames + len(base) - 1) // len(base)
    frames = (base * rep)[:args.c5_frames]
    plain = (recs[1000:] * rep)[:args.c5_frames]
    run(f"C5 4 KB records, {len(d)} B dictionary ({len(base)} distinct frames replicated)", frames, [4096] * len(frames), plain, dictionary=d)


if __name__ == "__main__":
    main()
