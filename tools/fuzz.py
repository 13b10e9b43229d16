#!/usr/bin/env python3
"""Differential fuzz of the GPU decoder: seeded inputs of many shapes and sizes, compressed with the system libzstd
at random levels / window sizes / flags, decoded in one batch (host walk and device walk) and compared with the
plain bytes through XXH64 (oracle's XXH64 on the CPU).  Frames that fail are written to gpurun_out/ for replay.

    python tools/fuzz.py [--rounds R] [--frames N] [--seed S]
"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zstd_amd import api, corpus  # noqa: E402
from oracle import oracle  # noqa: E402


def gen_plain(rng: np.random.Generator, n: int) -> bytes:
    kind = int(rng.integers(0, 9))
    seed = int(rng.integers(0, 1 << 30))
    if kind == 0:
        return corpus.text_like(seed, n)
    if kind == 1:
        return corpus.binary_records(seed, n)
    if kind == 2:       # noise
        return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    if kind == 3:       # few symbols, skewed
        k = int(rng.integers(2, 40))
        p = rng.random(k) ** 3
        return rng.choice(k, size=n, p=p / p.sum()).astype(np.uint8).tobytes()
    if kind == 4:       # periodic with a random period, a few flipped bytes
        period = int(rng.integers(1, 400))
        a = np.tile(rng.integers(0, 256, size=period, dtype=np.uint8), n // period + 1)[:n].copy()
        for _ in range(int(rng.integers(0, 20))):
            if n:
                a[int(rng.integers(0, n))] ^= 0x41
        return a.tobytes()
    if kind == 5:       # long zero runs with islands of noise: long matches and long literal runs
        a = np.zeros(n, dtype=np.uint8)
        pos = 0
        while pos < n:
            pos += int(rng.integers(1, 20000))
            ln = int(rng.integers(1, 6000))
            a[pos:pos + ln] = rng.integers(0, 256, size=max(0, min(ln, n - pos)), dtype=np.uint8)
            pos += ln
        return a.tobytes()
    if kind == 6:       # splices of a text: far repeats of every length
        base = np.frombuffer(corpus.text_like(seed, max(n, 1)), dtype=np.uint8)
        out = bytearray()
        while len(out) < n:
            s = int(rng.integers(0, max(1, n)))
            ln = int(rng.integers(1, 3000))
            out += base[s:s + ln].tobytes()
            if rng.random() < 0.3:
                out += rng.integers(0, 256, size=int(rng.integers(1, 200)), dtype=np.uint8).tobytes()
        return bytes(out[:n])
    if kind == 7:       # short words with a random byte after each: very many short sequences
        return corpus_many(rng, n)
    return corpus.mixed(seed, n)[: n // 2] + corpus.mixed(seed + 1, n)[: n - n // 2]


def corpus_many(rng, n):
    words = [rng.integers(0, 256, size=int(rng.integers(3, 6)), dtype=np.uint8).tobytes() for _ in range(int(rng.integers(4, 80)))]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(0, len(words)))]
        out.append(int(rng.integers(0, 256)))
    return bytes(out[:n])


def run(args, dctx=None) -> int:
    """Runs the rounds `args` asks for (rounds, frames, seed, max_kb, corrupt); returns the number of mismatches."""
    import torch
    assert torch.cuda.is_available()
    dctx = dctx or api.DCtx(0)
    tls = {}
    use_dict = getattr(args, "dict", False) or getattr(args, "records", False)
    records = getattr(args, "records", False)
    dictionary, ddict = None, None
    rec_pool = None
    if records:
        # batches of small structured records, all compressed against one dictionary trained on such records: every block
        # takes its Huffman and sequence tables from the dictionary, the shape that gets the kernels' shared-table variants
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_fixtures
        zt = corpus.SystemZstd()
        rec_pool = gen_fixtures.records(1000 + args.seed, 1000 + args.rounds * args.frames)
        dictionary = zt.train_dictionary(rec_pool[:1000], 112640)
        ddict = api.DDict(dctx, dictionary)
    elif use_dict:
        # a dictionary trained on text and record samples; with --dict every frame is compressed with it (some without
        # a dictionary ID in the header) and sizes stay small, where a dictionary matters
        zt = corpus.SystemZstd()
        rs = np.random.default_rng(args.seed)
        samples = [corpus.mixed(int(rs.integers(0, 1 << 30)), int(rs.integers(500, 6000))) for _ in range(1500)]
        dictionary = zt.train_dictionary(samples, int(rs.choice([8192, 30000, 112640])))
        ddict = api.DDict(dctx, dictionary)

    def compress(job):
        import threading
        z = tls.setdefault(threading.get_ident(), corpus.SystemZstd())
        plain, level, ck, cs, wlog = job
        if use_dict:
            return z.compress(plain, level, checksum=ck, content_size=cs, window_log=wlog, dictionary=dictionary, dict_id=len(plain) % 7 != 0)
        return z.compress(plain, level, checksum=ck, content_size=cs, window_log=wlog)

    bad = 0
    for r in range(args.rounds):
        rng = np.random.default_rng(args.seed * 1000 + r)
        t0 = time.time()
        jobs = []
        for i in range(args.frames):
            n = int(np.exp(rng.uniform(0, np.log((16 if use_dict else args.max_kb) * 1024))))
            if rng.random() < 0.05:
                n = int(rng.integers(0, 40))
            plain = gen_plain(rng, n)
            if records:
                plain = rec_pool[1000 + r * args.frames + i]
            level = int(rng.choice([-5, -1, 1, 1, 2, 3, 3, 3, 4, 5, 6, 7, 9, 12, 15, 19]))
            if records:
                level = 3
            if level >= 12 and n > 96 * 1024:
                level = 3
            wlog = int(rng.integers(10, 24)) if rng.random() < 0.3 else 0
            jobs.append((plain, level, bool(rng.integers(0, 2)), bool(rng.random() < 0.8), wlog))
        with ThreadPoolExecutor(16) as ex:
            frames = list(ex.map(compress, jobs))
        caps = [len(j[0]) for j in jobs]
        wantErr = [0] * len(jobs)
        wantLen = list(caps)
        oracle_diffs = []
        if args.corrupt:
            damaged = []
            for fr in frames:
                a = bytearray(fr)
                for _ in range(int(rng.integers(1, 4))):
                    hi = len(a) if rng.random() < 0.5 else min(len(a), 200)
                    a[int(rng.integers(0, hi))] = int(rng.integers(0, 256))
                if rng.random() < 0.1:
                    a = a[:int(rng.integers(0, len(a) + 1))]
                damaged.append(bytes(a))
            frames = damaged

            def expect(k):
                # the verdict of the system libzstd (1.4.8, the release the reference transliterates) where the box has it,
                # and the oracle must agree with it; frames whose magic selects the system library's decoders of the pre-1.0
                # formats (the reference has none) are judged by the oracle alone
                import threading
                out, err = oracle.decompress(frames[k], caps[k], dictionary)
                legacy = len(frames[k]) >= 4 and frames[k][1:4] == b"\xb5\x2f\xfd" and 0x22 <= frames[k][0] <= 0x27
                if corpus.SystemZstd.available() and not legacy:
                    z = tls.setdefault(threading.get_ident(), corpus.SystemZstd())
                    zout, zerr = z.decompress(frames[k], caps[k], dictionary)
                    if (zerr, zout) != (err, out):
                        print(f"ORACLE DIFFERS FROM LIBZSTD round {r} frame {k}: libzstd {zerr} oracle {err}", flush=True)
                        oracle_diffs.append(k)
                    out, err = zout, zerr
                return (oracle.xxh64(out) if not err else 0, err, len(out) if not err else 0)
            with ThreadPoolExecutor(16) as ex:
                exp = list(ex.map(expect, range(len(frames))))
            want = [e[0] for e in exp]; wantErr = [e[1] for e in exp]; wantLen = [e[2] for e in exp]
            bad += len(oracle_diffs)
        else:
            want = [oracle.xxh64(j[0]) for j in jobs]
        src, fin, dst_bytes = api.pack_frames(frames, caps)
        d_src = torch.from_numpy(src).cuda()
        t1 = time.time()
        digests = []
        for device_walk in (False, True):
            b = api.Batch(dctx, None, fin, ddict, dev_src_ptr=d_src.data_ptr(), dev_src_bytes=src.nbytes) if device_walk else api.Batch(dctx, src, fin, ddict)
            digests.append(b.descriptor_digest)
            d_dst = torch.zeros(max(dst_bytes, 1), dtype=torch.uint8, device="cuda")
            b.decode(d_src.data_ptr(), d_dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
            st = b.status()
            got = api.xxh64_batch(dctx, d_dst.data_ptr(), fin["dstOff"], st["regenSize"])
            for i in range(len(jobs)):
                if args.corrupt and int(st[i]["errorCode"]) == 72 and wantErr[i] != 72:
                    continue      # a batch entry must be exactly one frame: damage that moves the frame's end is srcSize_wrong
                                  # here before anything is decoded; the one-shot comparison below covers those frames
                ok = int(st[i]["errorCode"]) == wantErr[i] and (wantErr[i] != 0 or (int(st[i]["regenSize"]) == wantLen[i] and int(got[i]) == want[i]))
                if not ok:
                    bad += 1
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    fn = os.path.join(ROOT, "gpurun_out", f"fuzz_fail_s{args.seed}_r{r}_{i}.zst")
                    with open(fn, "wb") as f:
                        f.write(frames[i])
                    print(f"MISMATCH round {r} frame {i} walk={'device' if device_walk else 'host'} err={int(st[i]['errorCode'])} "
                          f"regen={int(st[i]['regenSize'])} want={wantLen[i]} wantErr={wantErr[i]} level={jobs[i][1]} wlog={jobs[i][4]} -> {fn}", flush=True)
            b.close()
        if args.corrupt:          # the one-shot entry point, frame by frame: verdict and bytes exactly as the oracle's
            for i in range(len(jobs)):
                try:
                    out = api.ZSTD_decompress_usingDDict(dctx, frames[i], caps[i], ddict) if ddict else api.ZSTD_decompressDCtx(dctx, frames[i], caps[i])
                    gerr, gh, gl = 0, oracle.xxh64(out), len(out)
                except api.ZstdError as ex:
                    gerr, gh, gl = ex.code, 0, 0
                if gerr != wantErr[i] or (not gerr and (gh != want[i] or gl != wantLen[i])):
                    bad += 1
                    fn = os.path.join(ROOT, "gpurun_out", f"fuzz_fail_s{args.seed}_r{r}_{i}.zst")
                    os.makedirs(os.path.dirname(fn), exist_ok=True)
                    with open(fn, "wb") as f:
                        f.write(frames[i])
                    print(f"ONE-SHOT MISMATCH round {r} frame {i} err={gerr} wantErr={wantErr[i]} len={gl} wantLen={wantLen[i]} level={jobs[i][1]} -> {fn}", flush=True)
        if getattr(args, "bufferless", False):
            # the same frames (damaged or not) block by block through zstdgpu_decompressContinue: every call's size, kind and
            # result and the bytes against the system libzstd driven by the same loop (tests/continue_driver.py)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import continue_driver as cd
            sysbe = cd.SystemBackend()
            for i in range(len(jobs)):
                got = cd.drive(cd.GpuBackend(dctx), frames[i], caps[i], dictionary, keep_steps=True)
                ref = cd.drive(sysbe, frames[i], caps[i], dictionary, keep_steps=True)
                if got["trace"] != ref["trace"] or got["out"] != ref["out"]:
                    bad += 1
                    fn = os.path.join(ROOT, "gpurun_out", f"fuzz_fail_bl_s{args.seed}_r{r}_{i}.zst")
                    os.makedirs(os.path.dirname(fn), exist_ok=True)
                    with open(fn, "wb") as f:
                        f.write(frames[i])
                    print(f"BUFFERLESS MISMATCH round {r} frame {i}: {cd.first_difference(got['trace'], ref['trace'])} cap={caps[i]} level={jobs[i][1]} -> {fn}", flush=True)
        if digests[0] != digests[1]:
            bad += 1
            print(f"DIGEST MISMATCH round {r}", flush=True)
        print(f"round {r}: {len(jobs)} frames, {sum(caps) / 1e6:.1f} MB plain, compress {t1 - t0:.1f}s, decode+check {time.time() - t1:.1f}s, bad so far {bad}", flush=True)
    if ddict:
        ddict.close()
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-kb", type=int, default=400)
    ap.add_argument("--corrupt", action="store_true", help="damage every frame (1-3 random bytes) and compare verdicts and bytes with the system libzstd (and the CPU oracle with it)")
    ap.add_argument("--dict", action="store_true", help="small frames compressed with a trained dictionary")
    ap.add_argument("--records", action="store_true", help="batches of 4 KB structured records compressed against a dictionary trained on such records (level 3)")
    ap.add_argument("--bufferless", action="store_true", help="also decode every frame block by block through zstdgpu_decompressContinue and compare each call with the system libzstd's")
    bad = run(ap.parse_args())
    print("FUZZ", "FAILED" if bad else "OK")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
