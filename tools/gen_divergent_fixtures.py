#!/usr/bin/env python3
"""Damaged frames on which the decoders that could claim to be "the reference" disagree, committed with every verdict
under tests/golden/divergent/:

  zstd148     what the system libzstd 1.4.8 returns (error code, or size + sha256 of the bytes)
  as_written  what the port's text prescribes where it differs from 1.4.8's dispatch: always the prefetching sequence loop
              (decompress/zstd_decompress_block.pas:1541), no decoders of the pre-1.0 formats (ZSTD_LEGACY_SUPPORT = 0)
  x1_only     the same frame with the single-symbol Huffman decoder forced (what a decoder without HUF_selectDecoder says)

Classes looked for: a sequence stream that runs dry before its last sequence (1.4.8 keeps reading its last container and
may accept, overflow the output, or fail the checksum; the prefetching loop says corruption_detected); a sequence
stream that is not used up (1.4.8 rejects, the prefetching loop does not look); a literal stream only the
double-symbol decoder's last-symbol rule accepts (huf_decompress.pas:923-940); a first frame with the magic number of
a pre-1.0 format (the system library was built with those decoders).

The rule this build follows -- oracle and GPU alike -- is zstd 1.4.8's, as dispatched by 1.4.8, without legacy formats.

    python tools/gen_divergent_fixtures.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "tests", "golden", "divergent")


def verdict(out, err):
    return {"error": err} if err else {"size": len(out), "sha256": hashlib.sha256(out).hexdigest()}


def main():
    import fuzz as gen
    import fuzz_cpu
    from oracle import oracle
    from zstd_amd import corpus
    z = corpus.SystemZstd()
    assert z.version == "1.4.8", z.version
    os.makedirs(OUT, exist_ok=True)
    want = {"dry_accepted": 3, "dry_dst_too_small": 2, "dry_checksum": 2, "unfinished_stream": 3, "huf_last_symbol": 3, "legacy_magic": 2}
    found = {k: [] for k in want}
    rng = np.random.default_rng(20240611)
    tries = 0
    while any(len(found[k]) < want[k] for k in want) and tries < 400000:
        tries += 1
        n = int(np.exp(rng.uniform(np.log(64), np.log(24 * 1024))))
        plain = gen.gen_plain(rng, n)
        level = int(rng.choice([1, 3, 3, 5, 9, 19]))
        fr = z.compress(plain, level, checksum=bool(rng.integers(0, 2)), content_size=bool(rng.random() < 0.8))
        fr = fuzz_cpu.damage(rng, fr)
        cap = n
        w_out, w_err = z.decompress(fr, cap)
        oracle.set_body(oracle.BODY_AS_ZSTD_148); oracle.set_huf(0)
        o_out, o_err = oracle.decompress(fr, cap)
        oracle.set_body(oracle.BODY_ALWAYS_PREFETCHING)
        l_out, l_err = oracle.decompress(fr, cap)
        oracle.set_body(oracle.BODY_AS_ZSTD_148); oracle.set_huf(1)
        x_out, x_err = oracle.decompress(fr, cap)
        oracle.set_huf(0)
        legacy = fuzz_cpu.is_legacy_magic(fr)
        if not legacy:
            assert (w_err, w_out) == (o_err, o_out), "oracle differs from libzstd: run tools/fuzz_cpu.py"
        cls = None
        if legacy and w_err != o_err:
            cls = "legacy_magic"
        elif (x_err, x_out) != (o_err, o_out):
            cls = "huf_last_symbol"
        elif (l_err, l_out) != (o_err, o_out):
            if l_err == 20:
                cls = {0: "dry_accepted", 70: "dry_dst_too_small", 22: "dry_checksum"}.get(o_err)
            elif o_err == 20:
                cls = "unfinished_stream"
        if cls is None or len(found[cls]) >= want[cls] or len(fr) > 12000:
            continue
        name = f"{cls}_{len(found[cls])}"
        with open(os.path.join(OUT, name + ".zst"), "wb") as f:
            f.write(fr)
        found[cls].append({"name": name, "file": name + ".zst", "class": cls, "capacity": cap, "csize": len(fr),
                           "zstd148": verdict(w_out, w_err), "build": verdict(o_out, o_err),
                           "as_written": verdict(l_out, l_err) if not legacy else {"error": 10}, "x1_only": verdict(x_out, x_err)})
    entries = [e for k in want for e in found[k]]
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump({"generator": "tools/gen_divergent_fixtures.py", "libzstd": z.version, "frames": entries}, f, indent=1)
        f.write("\n")
    print({k: len(v) for k, v in found.items()}, "after", tries, "frames")


if __name__ == "__main__":
    main()
