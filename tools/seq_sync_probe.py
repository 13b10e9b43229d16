"""Can a block's sequence bitstream be decoded in parts?  CPU study (no GPU): builds tools/seq_sync_probe.c against the
oracle and runs it over frames of the benchmark shapes (config 2: level 3, config 3: level 19; 128 KiB of text-like
or binary records per frame, compressed here by the system libzstd).

    python tools/seq_sync_probe.py [--frames 64] [--starts 8]

Prints one JSON line per (config, corpus): starts, how many fell into step with the true decode, and after how many
sequences (quantiles), for starts at a true sequence boundary with arbitrary states (mode 0) and at an arbitrary bit
position (mode 1)."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zstd_amd import build, corpus  # noqa: E402


def load():
    out_dir = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "seq_sync_probe.so")
    src = os.path.join(ROOT, "tools", "seq_sync_probe.c")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-o", so, src])
    L = ctypes.CDLL(so)
    L.probe_begin.argtypes = [ctypes.c_uint, ctypes.c_ulonglong]
    L.probe_frame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    L.probe_frame.restype = ctypes.c_size_t
    L.probe_end.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
    L.probe_end.restype = ctypes.c_ulonglong
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64, help="frames per (config, corpus)")
    ap.add_argument("--starts", type=int, default=8, help="evenly spaced starts per block and mode (K: K-1 starts)")
    args = ap.parse_args()
    build.build_corpus()
    L = load()
    z = corpus.SystemZstd()
    n = 128 * 1024
    dst = ctypes.create_string_buffer(n)
    for cfg, level in (("2", 3), ("3", 19)):
        for kind, gen, seeds in (("text", corpus.text_like, range(0, 2 * args.frames, 2)), ("binary", corpus.binary_records, range(1, 2 * args.frames, 2))):
            L.probe_begin(args.starts, 12345)
            for s in seeds:
                plain = gen(s, n)
                fr = z.compress(plain, level, checksum=(cfg == "3"))
                r = L.probe_frame(dst, n, fr, len(fr))
                assert r == n and dst.raw == plain
            out = (ctypes.c_ulonglong * 16)()
            seqs = ctypes.c_ulonglong()
            blocks = L.probe_end(out, ctypes.byref(seqs))
            line = {"config": cfg, "level": level, "corpus": kind, "blocks": int(blocks), "sequences_per_block": int(seqs.value // max(1, blocks))}
            for m, name in ((0, "start_on_boundary"), (1, "start_anywhere")):
                st, sy, ch, steps, q10, q50, q90, mx = [int(out[8 * m + i]) for i in range(8)]
                line[name] = {"starts": st, "fell_into_step": sy, "share": round(sy / max(1, st), 4),
                              "speculative_sequences_decoded": steps, "boundaries_with_only_the_cursor_right": ch,
                              "sequences_until_in_step_q10_q50_q90_max": [q10, q50, q90, mx]}
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
