"""The C-ABI library: loads, exports every symbol include/zstdgpu.h declares, and its host-only
functions (header queries, descriptor builder, error helpers) behave like the reference's.
No GPU work happens here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import FRAMES, FRAME_IDS, ROOT, read_frame
from oracle import oracle
from zstd_amd import api


def header_symbols():
    text = open(os.path.join(ROOT, "include", "zstdgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zstdgpu_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = api.lib()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(api.EXPORTS) == syms


def test_error_helpers():
    L = api.lib()
    assert L.zstdgpu_versionNumber() == 10408
    assert not L.zstdgpu_isError(0) and not L.zstdgpu_isError(1 << 40)
    e = (1 << 64) - 20
    assert L.zstdgpu_isError(e) and L.zstdgpu_getErrorCode(e) == 20
    assert L.zstdgpu_getErrorName(e) == b"Corrupted block detected"
    assert not L.zstdgpu_isError((1 << 64) - 121)


@pytest.mark.parametrize("e", FRAMES, ids=FRAME_IDS)
def test_header_queries_match_oracle(e):
    fr = read_frame(e)
    OL = oracle.lib()
    assert api.lib().zstdgpu_getFrameContentSize(fr, len(fr)) == OL.oracle_getFrameContentSize(fr, len(fr))
    assert api.lib().zstdgpu_findFrameCompressedSize(fr, len(fr)) == OL.oracle_findFrameCompressedSize(fr, len(fr))


def test_build_descriptors_walks_blocks():
    seen = set()
    for name in ("mixed_rle_raw_text", "rle300k", "raw200k"):
        e = next(x for x in FRAMES if x["name"] == name)
        fr = read_frame(e)
        src, fin, _ = api.pack_frames([fr], [e["capacity"]])
        fds, bds, nb = api.build_descriptors(src, fin)
        assert fds[0].hostError == 0 and fds[0].nBlocks == nb and fds[0].contentSize == e["size"]
        seen |= {bds[i].blockType for i in range(nb)}
        assert bds[nb - 1].lastBlock == 1 and all(bds[i].lastBlock == 0 for i in range(nb - 1))
        # payloads tile the frame: header, then 3-byte block headers between payloads, then the checksum
        pos = int(fin[0]["srcOff"]) + len(fr) - 4
        assert int(bds[nb - 1].srcOff) + bds[nb - 1].cSize == pos
        for i in range(nb - 1):
            assert int(bds[i].srcOff) + bds[i].cSize + 3 == int(bds[i + 1].srcOff)
        assert fds[0].checksumFlag == 1
        assert fds[0].checksum == int.from_bytes(fr[-4:], "little")
    assert seen == {0, 1, 2}          # raw, RLE and compressed blocks all met


def test_build_descriptors_flags_host_errors():
    good = read_frame(next(x for x in FRAMES if x["name"] == "text1000_l3_c1"))
    cases = [good[:4], b"\x00" * 20, good[:len(good) // 2], good + b"xx"]
    src, fin, _ = api.pack_frames(cases, [1000] * len(cases))
    fds, bds, nb = api.build_descriptors(src, fin)
    assert fds[0].hostError == 72                     # shorter than the smallest frame
    assert fds[1].hostError == 10                     # bad magic
    assert fds[2].hostError == 0 and bds[fds[2].firstBlock + fds[2].nBlocks - 1].blockType == 3   # walk stops inside the frame
    assert fds[3].hostError == 72                     # bytes after the frame


# ------------------------------------------------------------------ header / size queries against libzstd's answers
import json  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "queries.json")) as _f:
    QUERIES = json.load(_f)


def _check_queries(data: bytes, q, tag):
    def size_or_error(call):
        try:
            return {"value": call()}
        except api.ZstdError as ex:
            return {"error": ex.code}

    assert api.ZSTD_isFrame(data) == q["isFrame"], tag
    assert size_or_error(lambda: api.ZSTD_frameHeaderSize(data)) == q["frameHeaderSize"], tag
    try:
        r, hdr = api.ZSTD_getFrameHeader(data)
        assert {"value": r} == q["getFrameHeader"], tag
        if r == 0:
            assert hdr == q["header"], tag
    except api.ZstdError as ex:
        assert {"error": ex.code} == q["getFrameHeader"], tag
    assert api.ZSTD_getFrameContentSize(data) == q["getFrameContentSize"], tag
    assert api.ZSTD_getDecompressedSize(data) == q["getDecompressedSize"], tag
    assert size_or_error(lambda: api.ZSTD_findFrameCompressedSize(data)) == q["findFrameCompressedSize"], tag
    assert api.ZSTD_findDecompressedSize(data) == q["findDecompressedSize"], tag
    assert api.ZSTD_decompressBound(data) == q["decompressBound"], tag


def test_queries_match_libzstd_on_every_fixture():
    """ZSTD_isFrame, ZSTD_frameHeaderSize, ZSTD_getFrameHeader, ZSTD_getFrameContentSize, ZSTD_getDecompressedSize,
    ZSTD_findFrameCompressedSize, ZSTD_findDecompressedSize, ZSTD_decompressBound: libzstd 1.4.8's recorded answers."""
    assert len(QUERIES["whole"]) == len(FRAMES)
    for e in FRAMES:
        _check_queries(read_frame(e), QUERIES["whole"][e["name"]], e["name"])


def test_queries_match_libzstd_on_prefixes():
    """The same on the first 0..23 bytes of some frames (bytes-wanted results, half-read skippable headers)."""
    for name, cuts in QUERIES["prefixes"].items():
        data = read_frame(next(x for x in FRAMES if x["name"] == name))
        for cut, q in cuts.items():
            _check_queries(data[:int(cut)], q, f"{name}[:{cut}]")
