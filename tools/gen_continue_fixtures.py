#!/usr/bin/env python3
"""Writes tests/golden/continue_traces.json: what the system libzstd (1.4.8) answers when the frames of the golden
manifests are decoded through its bufferless interface (ZSTD_decompressBegin[_usingDict], ZSTD_nextSrcSizeToDecompress,
ZSTD_nextInputType, ZSTD_decompressContinue), driven by tests/continue_driver.py.

Per frame and variant: the number of calls, a SHA-256 over the (input type, bytes handed over, result) of every call, the
first error with the call it came from, the bytes consumed and the SHA-256 of the output.  Variants: "plain"; "rawchunk"
(raw blocks handed over 1000 bytes at a time) for the frames that hold raw blocks.

Run from the repository root: python tools/gen_continue_fixtures.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import continue_driver as cd  # noqa: E402
from zstd_amd import corpus  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def has_raw_block(data: bytes) -> bool:
    """Whether a (sound, single) zstd1 frame holds a raw block of more than 1000 bytes."""
    try:
        if len(data) < 6 or data[:4] != b"\x28\xb5\x2f\xfd":
            return False
        fhd = data[4]
        pos = 5 + (0 if (fhd >> 5) & 1 else 1) + (0, 1, 2, 4)[fhd & 3] + ((0, 2, 4, 8)[fhd >> 6] or (1 if (fhd >> 5) & 1 else 0))
        while pos + 3 <= len(data):
            h = data[pos] | (data[pos + 1] << 8) | (data[pos + 2] << 16)
            typ, size = (h >> 1) & 3, h >> 3
            if typ == 0 and size > 1000:
                return True
            if typ == 3:
                return False
            pos += 3 + (1 if typ == 1 else size)
            if h & 1:
                return False
    except IndexError:
        pass
    return False


def main():
    z = corpus.SystemZstd()
    out = {"generator": "tools/gen_continue_fixtures.py", "libzstd": z.version, "entries": []}
    dicts = json.load(open(os.path.join(GOLD, "manifest.json")))["dicts"]

    def read_dict(name):
        return open(os.path.join(GOLD, dicts[name]["file"]), "rb").read() if name else None

    sets = [("", json.load(open(os.path.join(GOLD, "manifest.json")))["frames"]),
            ("params", json.load(open(os.path.join(GOLD, "params", "manifest.json")))["frames"]),
            ("divergent", json.load(open(os.path.join(GOLD, "divergent", "manifest.json")))["frames"])]
    for sub, frames in sets:
        for e in frames:
            data = open(os.path.join(GOLD, sub, e["file"]), "rb").read()
            d = read_dict(e.get("dict"))
            variants = [("plain", 0)]
            if has_raw_block(data):
                variants.append(("rawchunk", 1000))
            for vname, chunk in variants:
                be = cd.SystemBackend(magicless=bool(e.get("format", 0)), ignore_checksum=bool(e.get("ignore_checksum", 0)))
                r = cd.drive(be, data, e["capacity"], d, raw_chunk=chunk)
                out["entries"].append({"set": sub, "name": e["name"], "variant": vname, "raw_chunk": chunk, **r})
    with open(os.path.join(GOLD, "continue_traces.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
        f.write("\n")
    n_err = sum(1 for x in out["entries"] if x["err"])
    print("wrote %d entries (%d end in an error) from libzstd %s" % (len(out["entries"]), n_err, z.version))


if __name__ == "__main__":
    main()
