#!/usr/bin/env python3
"""Differential fuzz of the CPU oracle against the system libzstd (1.4.8, the release the reference transliterates),
on the CPU only: seeded inputs of the shapes of tools/fuzz.py, compressed by libzstd, then damaged (1-3 random bytes,
sometimes truncated); the oracle's verdict -- error code, or size and bytes -- must equal libzstd's on every frame.

    python tools/fuzz_cpu.py [--rounds R] [--frames N] [--seed S] [--workers W] [--save DIR]

Frames on which the two differ are written to --save (default gpurun_out/divergent) with both verdicts in their names.
"""
import argparse
import hashlib
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def damage(rng, fr: bytes) -> bytes:
    a = bytearray(fr)
    mode = rng.random()
    if mode < 0.15 and len(a) > 12:
        # aim at the tail of the frame: the end of a sequence bitstream, the checksum
        for _ in range(int(rng.integers(1, 3))):
            a[len(a) - 1 - int(rng.integers(0, min(len(a), 24)))] = int(rng.integers(0, 256))
    else:
        for _ in range(int(rng.integers(1, 4))):
            hi = len(a) if rng.random() < 0.5 else min(len(a), 200)
            if hi:
                a[int(rng.integers(0, hi))] = int(rng.integers(0, 256))
    if rng.random() < 0.1:
        a = a[:int(rng.integers(0, len(a) + 1))]
    return bytes(a)


def is_legacy_magic(fr: bytes) -> bool:
    return len(fr) >= 4 and fr[1:4] == b"\xb5\x2f\xfd" and 0x22 <= fr[0] <= 0x27


def long_round(job):
    """Frames that zstd 1.4.8 hands to its prefetching sequence loop: window above 2^24 and far offsets."""
    seed, r, n_variants = job
    from zstd_amd import corpus
    from oracle import oracle
    z = corpus.SystemZstd()
    rng = np.random.default_rng(seed * 7919 + r)
    chunk = rng.integers(0, 256, size=1 << 16, dtype=np.uint8).tobytes()
    text = corpus.text_like(int(rng.integers(0, 1 << 30)), 9 << 20)
    # the same 64 KB chunks recur more than 8 MiB apart: offsets with 23 and more extra bits
    plain = chunk + text + chunk + corpus.binary_records(int(rng.integers(0, 1 << 30)), 9 << 20) + chunk + text[: 1 << 20]
    fr = z.compress(plain, int(rng.choice([1, 3, 5])), checksum=bool(rng.integers(0, 2)), content_size=False, window_log=25)
    want, werr = z.decompress(fr, len(plain))
    assert werr == 0 and want == plain
    diffs, n_err = [], 0
    # where the blocks are: damage the tail of block bitstreams (where a sequence stream runs dry or is left unfinished)
    pos, ends = 6, []
    while pos + 3 <= len(fr):
        h = fr[pos] | (fr[pos + 1] << 8) | (fr[pos + 2] << 16)
        size = 1 if ((h >> 1) & 3) == 1 else h >> 3
        pos += 3 + size
        ends.append(pos)
        if h & 1:
            break
    for v in range(n_variants):
        a = bytearray(fr)
        for _ in range(int(rng.integers(1, 3))):
            e = ends[int(rng.integers(0, len(ends)))]
            a[e - 1 - int(rng.integers(0, 6))] = int(rng.integers(0, 256))
        dmg = bytes(a)
        want, werr = z.decompress(dmg, len(plain))
        got, gerr = oracle.decompress(dmg, len(plain))
        n_err += 1 if werr else 0
        if werr != gerr or (not werr and want != got):
            diffs.append((dmg, len(plain), werr, gerr, None, 0, False))
    return r, n_variants, n_err, diffs


def one_round(job):
    seed, r, n_frames, max_kb, use_dict, valid = job
    import fuzz as gpu_fuzz      # the input generators
    from zstd_amd import corpus
    from oracle import oracle
    z = corpus.SystemZstd()
    rng = np.random.default_rng(seed * 100003 + r)
    dictionary = None
    if use_dict:
        rs = np.random.default_rng(seed)
        samples = [corpus.mixed(int(rs.integers(0, 1 << 30)), int(rs.integers(500, 6000))) for _ in range(600)]
        dictionary = z.train_dictionary(samples, int(rs.choice([8192, 30000])))
    diffs = []
    n_err = 0
    for i in range(n_frames):
        n = int(np.exp(rng.uniform(0, np.log((16 if use_dict else max_kb) * 1024))))
        if rng.random() < 0.05:
            n = int(rng.integers(0, 40))
        plain = gpu_fuzz.gen_plain(rng, n)
        level = int(rng.choice([-5, -1, 1, 1, 2, 3, 3, 3, 4, 5, 6, 7, 9, 12, 15, 19]))
        if level >= 12 and n > 64 * 1024:
            level = 3
        wlog = int(rng.integers(10, 24)) if rng.random() < 0.3 else 0
        if use_dict:
            fr = z.compress(plain, level, checksum=bool(rng.integers(0, 2)), content_size=bool(rng.random() < 0.8), window_log=wlog,
                            dictionary=dictionary, dict_id=n % 7 != 0)
        else:
            fr = z.compress(plain, level, checksum=bool(rng.integers(0, 2)), content_size=bool(rng.random() < 0.8), window_log=wlog)
        if not valid:
            fr = damage(rng, fr)
        cap = n + (int(rng.integers(0, 64)) if rng.random() < 0.2 else 0)
        want, werr = z.decompress(fr, cap, dictionary)
        got, gerr = oracle.decompress(fr, cap, dictionary)
        n_err += 1 if werr else 0
        if is_legacy_magic(fr):
            # The system library is built with its decoders of the pre-1.0 formats (magic 0xFD2FB522..27); the reference has
            # none (ZSTD_LEGACY_SUPPORT = 0, zstd_decompressf.pas:41) and answers prefix_unknown: not a decoder difference.
            continue
        if werr != gerr or (not werr and want != got):
            diffs.append((fr, cap, werr, gerr, None if werr else hashlib.sha256(want).hexdigest()[:12], level, use_dict))
    return r, n_frames, n_err, diffs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=50)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-kb", type=int, default=96)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--dict", action="store_true")
    ap.add_argument("--valid", action="store_true", help="leave the frames undamaged")
    ap.add_argument("--long", action="store_true", help="20 MB frames with a 32 MiB window and far matches, damaged at the ends of blocks")
    ap.add_argument("--save", default=os.path.join(ROOT, "gpurun_out", "divergent"))
    args = ap.parse_args()
    from zstd_amd import build
    build.build_corpus(); build.build_oracle()
    t0 = time.time()
    total = errs = 0
    ndiff = 0
    jobs = [(args.seed, r, args.frames, args.max_kb, args.dict, args.valid) for r in range(args.rounds)]
    fn_round = one_round
    if args.long:
        fn_round = long_round
        jobs = [(args.seed, r, args.frames) for r in range(args.rounds)]
    with ProcessPoolExecutor(args.workers) as ex:
        for r, n, n_err, diffs in ex.map(fn_round, jobs):
            total += n; errs += n_err
            for k, (fr, cap, werr, gerr, sha, level, ud) in enumerate(diffs):
                os.makedirs(args.save, exist_ok=True)
                fn = os.path.join(args.save, f"s{args.seed}_r{r}_{k}_cap{cap}_zstd{werr}_oracle{gerr}{'_dict' if ud else ''}.zst")
                with open(fn, "wb") as f:
                    f.write(fr)
                ndiff += 1
                print(f"DIFF round {r}: libzstd {werr} oracle {gerr} cap {cap} level {level} len {len(fr)} -> {fn}", flush=True)
            if r % 10 == 9:
                print(f"round {r}: {total} frames so far ({errs} rejected by libzstd), {ndiff} differences, {time.time() - t0:.0f}s", flush=True)
    print(f"{total} frames, {errs} rejected by libzstd, {ndiff} differences, {time.time() - t0:.0f}s")
    print("FUZZ_CPU", "DIFFERENCES" if ndiff else "OK")
    sys.exit(1 if ndiff else 0)


if __name__ == "__main__":
    main()
