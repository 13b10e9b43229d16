#!/usr/bin/env python3
"""Generate tests/golden/queries.json: what the system libzstd.so.1.4.8 answers to the header / size queries
(ZSTD_isFrame, ZSTD_frameHeaderSize, ZSTD_getFrameHeader, ZSTD_getFrameContentSize, ZSTD_getDecompressedSize,
ZSTD_findFrameCompressedSize, ZSTD_findDecompressedSize, ZSTD_decompressBound) for every committed fixture and for
prefixes of some of them.  Run in the build container only, after tools/gen_fixtures.py; the output is committed.

    python tools/gen_query_fixtures.py
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zstd_amd import corpus  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


class FrameHeader(ctypes.Structure):
    _fields_ = [("frameContentSize", ctypes.c_ulonglong), ("windowSize", ctypes.c_ulonglong), ("blockSizeMax", ctypes.c_uint),
                ("frameType", ctypes.c_uint), ("headerSize", ctypes.c_uint), ("dictID", ctypes.c_uint), ("checksumFlag", ctypes.c_uint)]


def main():
    z = corpus.SystemZstd()
    assert z.version == "1.4.8", z.version
    L = z.lib
    vp, sz, ull = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_ulonglong
    L.ZSTD_isFrame.argtypes = [vp, sz]; L.ZSTD_isFrame.restype = ctypes.c_uint
    L.ZSTD_frameHeaderSize.argtypes = [vp, sz]; L.ZSTD_frameHeaderSize.restype = sz
    L.ZSTD_getFrameHeader.argtypes = [ctypes.POINTER(FrameHeader), vp, sz]; L.ZSTD_getFrameHeader.restype = sz
    for n in ("ZSTD_getFrameContentSize", "ZSTD_getDecompressedSize", "ZSTD_findDecompressedSize", "ZSTD_decompressBound"):
        getattr(L, n).argtypes = [vp, sz]; getattr(L, n).restype = ull
    L.ZSTD_findFrameCompressedSize.argtypes = [vp, sz]; L.ZSTD_findFrameCompressedSize.restype = sz

    def size_or_error(r):
        return {"error": L.ZSTD_getErrorCode(r)} if L.ZSTD_isError(r) else {"value": r}

    def query(data: bytes):
        buf = ctypes.create_string_buffer(data, max(len(data), 1))
        n = len(data)
        h = FrameHeader()
        r = L.ZSTD_getFrameHeader(ctypes.byref(h), buf, n)
        q = {
            "isFrame": L.ZSTD_isFrame(buf, n),
            "frameHeaderSize": size_or_error(L.ZSTD_frameHeaderSize(buf, n)),
            "getFrameHeader": size_or_error(r),
            "getFrameContentSize": L.ZSTD_getFrameContentSize(buf, n),
            "getDecompressedSize": L.ZSTD_getDecompressedSize(buf, n),
            "findFrameCompressedSize": size_or_error(L.ZSTD_findFrameCompressedSize(buf, n)),
            "findDecompressedSize": L.ZSTD_findDecompressedSize(buf, n),
            "decompressBound": L.ZSTD_decompressBound(buf, n),
        }
        if not L.ZSTD_isError(r) and r == 0:
            q["header"] = {f: int(getattr(h, f)) for f, _ in FrameHeader._fields_}
        return q

    with open(os.path.join(GOLD, "manifest.json")) as f:
        frames = json.load(f)["frames"]
    out = {"generator": "tools/gen_query_fixtures.py", "libzstd": z.version, "whole": {}, "prefixes": {}}
    for e in frames:
        with open(os.path.join(GOLD, e["file"]), "rb") as f:
            data = f.read()
        out["whole"][e["name"]] = query(data)
    for name in ("text4096_l3_c1", "skippable_mix", "skippable_only", "rec00_dict", "text300k_l3_nocsize", "concat3", "bin128k_l19_c1"):
        e = next(x for x in frames if x["name"] == name)
        with open(os.path.join(GOLD, e["file"]), "rb") as f:
            data = f.read()
        out["prefixes"][name] = {str(cut): query(data[:cut]) for cut in list(range(0, 24)) + [len(data) - 1]}
    with open(os.path.join(GOLD, "queries.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print(len(out["whole"]), "whole +", sum(len(v) for v in out["prefixes"].values()), "prefix queries,",
          os.path.getsize(os.path.join(GOLD, "queries.json")) // 1024, "KB")


if __name__ == "__main__":
    main()
