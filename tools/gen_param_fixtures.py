#!/usr/bin/env python3
"""Generate tests/golden/params/: frames whose decoding depends on a decoder parameter
(ZSTD_d_format, ZSTD_d_forceIgnoreChecksum, ZSTD_d_windowLogMax), with the verdict of the system
libzstd.so.1.4.8 under that parameter (one-shot ZSTD_decompressDCtx / ZSTD_decompress_usingDict, or
ZSTD_decompressStream for the entries marked "stream").  Run in the build container only; outputs
are committed.

    python tools/gen_param_fixtures.py
"""
import ctypes
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zstd_amd import corpus  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "params")
KB = 1024
C_FORMAT = 10                      # ZSTD_c_experimentalParam2 = ZSTD_c_format
D_WINDOWLOGMAX, D_FORMAT, D_IGNORECHECKSUM = 100, 1000, 1002


class InBuf(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


class OutBuf(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        os.remove(os.path.join(OUT, f))
    z = corpus.SystemZstd()
    assert z.version == "1.4.8", z.version
    L = z.lib
    L.ZSTD_DCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.ZSTD_DCtx_setParameter.restype = ctypes.c_size_t
    L.ZSTD_DCtx_reset.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ZSTD_DCtx_reset.restype = ctypes.c_size_t
    L.ZSTD_decompressDCtx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    L.ZSTD_decompressDCtx.restype = ctypes.c_size_t
    L.ZSTD_decompressStream.argtypes = [ctypes.c_void_p, ctypes.POINTER(OutBuf), ctypes.POINTER(InBuf)]
    L.ZSTD_decompressStream.restype = ctypes.c_size_t

    def compress(data, level=3, checksum=False, magicless=False, window_log=0, dictionary=None):
        L.ZSTD_CCtx_reset(z._cctx, 3)
        for p, v in ((z.C_LEVEL, level), (z.C_CHECKSUM, int(checksum)), (C_FORMAT, int(magicless))):
            assert not L.ZSTD_isError(L.ZSTD_CCtx_setParameter(z._cctx, p, v)), (p, v)
        if window_log:
            assert not L.ZSTD_isError(L.ZSTD_CCtx_setParameter(z._cctx, z.C_WINDOWLOG, window_log))
        if dictionary:
            assert not L.ZSTD_isError(L.ZSTD_CCtx_loadDictionary(z._cctx, dictionary, len(dictionary)))
        cap = L.ZSTD_compressBound(len(data))
        out = ctypes.create_string_buffer(cap)
        n = L.ZSTD_compress2(z._cctx, out, cap, data, len(data))
        assert not L.ZSTD_isError(n)
        return out.raw[:n]

    def set_params(fmt, ignore, wlm=0):
        L.ZSTD_DCtx_reset(z._dctx, 3)
        assert not L.ZSTD_isError(L.ZSTD_DCtx_setParameter(z._dctx, D_FORMAT, fmt))
        assert not L.ZSTD_isError(L.ZSTD_DCtx_setParameter(z._dctx, D_IGNORECHECKSUM, ignore))
        if wlm:
            assert not L.ZSTD_isError(L.ZSTD_DCtx_setParameter(z._dctx, D_WINDOWLOGMAX, wlm))

    def decode(frame, capacity, fmt, ignore, dictionary=None):
        set_params(fmt, ignore)
        out = ctypes.create_string_buffer(max(capacity, 1))
        if dictionary:
            n = L.ZSTD_decompress_usingDict(z._dctx, out, capacity, frame, len(frame), dictionary, len(dictionary))
        else:
            n = L.ZSTD_decompressDCtx(z._dctx, out, capacity, frame, len(frame))
        L.ZSTD_DCtx_reset(z._dctx, 3)
        if L.ZSTD_isError(n):
            return None, L.ZSTD_getErrorCode(n)
        return out.raw[:n], 0

    def decode_stream(frame, capacity, fmt, ignore, wlm, chunk):
        """Feeds `chunk` bytes per call (0: everything in one call, which lets libzstd take its single-pass shortcut)."""
        set_params(fmt, ignore, wlm)
        out = ctypes.create_string_buffer(max(capacity, 1))
        src = ctypes.create_string_buffer(frame, len(frame))
        ob = OutBuf(ctypes.cast(out, ctypes.c_void_p), capacity, 0)
        ib = InBuf(ctypes.cast(src, ctypes.c_void_p), min(chunk, len(frame)) if chunk else len(frame), 0)
        err = 0
        for _ in range(1 << 16):
            r = L.ZSTD_decompressStream(z._dctx, ctypes.byref(ob), ctypes.byref(ib))
            if L.ZSTD_isError(r):
                err = L.ZSTD_getErrorCode(r)
                break
            if ib.pos == ib.size and ib.size < len(frame):
                ib.size = min(ib.size + chunk, len(frame))
                continue
            if r == 0 and ib.pos == ib.size:
                break
            if ib.pos == ib.size and ob.pos == ob.size:
                break
        L.ZSTD_DCtx_reset(z._dctx, 3)
        return (None, err) if err else (out.raw[:ob.pos], 0)

    manifest = {"generator": "tools/gen_param_fixtures.py", "libzstd": z.version, "frames": []}

    def add(name, frame, capacity, fmt=0, ignore=0, dictionary=None, stream_wlm=None, chunk=0, note=""):
        if stream_wlm is not None:
            out, err = decode_stream(frame, capacity, fmt, ignore, stream_wlm, chunk)
        else:
            out, err = decode(frame, capacity, fmt, ignore, dictionary)
        fn = name + ".zst"
        with open(os.path.join(OUT, fn), "wb") as f:
            f.write(frame)
        e = {"name": name, "file": fn, "capacity": capacity, "format": fmt, "ignore_checksum": ignore, "note": note}
        if dictionary:
            e["dict"] = "dict16k"
        if stream_wlm is not None:
            e["stream_window_log_max"] = stream_wlm
            e["stream_chunk"] = chunk
        if err:
            e["error"] = err
        else:
            e["size"] = len(out)
            e["sha256"] = hashlib.sha256(out).hexdigest()
        manifest["frames"].append(e)
        return out, err

    text = corpus.text_like(701, 128 * KB)
    binr = corpus.binary_records(702, 40 * KB)
    big = corpus.mixed(703, 300 * KB)
    ml_text = compress(text, 3, True, magicless=True)
    ml_bin = compress(binr, 19, False, magicless=True)
    ml_big = compress(big, 3, True, magicless=True)
    ml_empty = compress(b"", 3, False, magicless=True)
    out, err = add("ml_text128k_c1", ml_text, len(text), fmt=1, note="magicless frame, checksum")
    assert out == text
    add("ml_bin40k_l19", ml_bin, len(binr), fmt=1, note="magicless frame, level 19")
    add("ml_mixed300k_c1", ml_big, len(big), fmt=1, note="magicless, 3 blocks")
    add("ml_empty", ml_empty, 0, fmt=1, note="magicless empty frame")
    out, err = add("ml_concat", ml_text + ml_bin + ml_empty + ml_big, len(text) + len(binr) + len(big), fmt=1, note="four magicless frames back to back")
    assert out == text + binr + big
    add("ml_as_zstd1", ml_text, len(text), fmt=0, note="magicless frame decoded as zstd1 -> error")
    add("zstd1_as_ml", compress(text, 3, True), len(text), fmt=1, note="zstd1 frame decoded as magicless -> error")
    add("ml_cap_small", ml_text, len(text) - 1, fmt=1, note="magicless, capacity too small")
    for cut in (0, 1, 2, 3, 4, 5, 6, len(ml_bin) // 2, len(ml_bin) - 1):
        add(f"ml_cut{cut}", ml_bin[:cut], len(binr), fmt=1, note="truncated magicless frame")
    add("ml_trailing", ml_bin + b"\x00\x00", len(binr), fmt=1, note="magicless frame + 2 stray bytes")
    skip = b"\x50\x2A\x4D\x18" + (5).to_bytes(4, "little") + b"hello"
    add("ml_skippable_mix", skip + ml_bin + skip, len(binr), fmt=1, note="skippable frames keep their magic in magicless mode")

    with open(os.path.join(ROOT, "tests", "golden", "dict16k.bin"), "rb") as f:
        d = f.read()
    rec = corpus.binary_records(704, 4096)
    out, err = add("ml_rec_dict", compress(rec, 3, True, magicless=True, dictionary=d), 4096, fmt=1, dictionary=d, note="magicless + dictionary")
    assert out == rec

    # checksum policy
    ck = compress(text, 3, True)
    bad = bytearray(ck); bad[-1] ^= 0x55
    add("ck_bad_validate", bytes(bad), len(text), ignore=0, note="damaged checksum trailer, validated -> checksum_wrong")
    out, err = add("ck_bad_ignore", bytes(bad), len(text), ignore=1, note="damaged checksum trailer, ignored")
    assert out == text
    add("ck_good_ignore", ck, len(text), ignore=1, note="good checksum, ignored")
    add("ck_cut_ignore", ck[:-2], len(text), ignore=1, note="trailer truncated: still an error when ignored")
    mlbad = bytearray(ml_text); mlbad[-2] ^= 1
    add("ck_ml_bad_ignore", bytes(mlbad), len(text), fmt=1, ignore=1, note="magicless + damaged trailer, ignored")
    two = bytes(bad) + ck
    add("ck_concat_ignore", two, 2 * len(text), ignore=1, note="two frames, first with damaged trailer, ignored")

    # streaming window limit: frames without a content size carry a window descriptor (byte 5), patched here to
    # announce a larger window than the encoder needed (decoders accept any window that covers the offsets used)
    def with_window_log(frame, wlog):
        assert frame[4] & 0x20 == 0          # not single-segment
        fr = bytearray(frame); fr[5] = (wlog - 10) << 3
        return bytes(fr)
    wide = with_window_log(z.compress(corpus.text_like(705, 600 * KB), 3, True, content_size=False), 22)
    add("st_wlog22_default", wide, 600 * KB, stream_wlm=0, note="stream, window 4 MiB under the default limit")
    add("st_wlog22_max20", wide, 600 * KB, stream_wlm=20, chunk=1000, note="stream, window 4 MiB over windowLogMax 20 -> windowTooLarge")
    add("st_wlog22_max21_onecall", wide, 600 * KB, stream_wlm=21, note="stream, window 4 MiB over windowLogMax 21, whole frame in one call")
    add("st_wlog22_max22", wide, 600 * KB, stream_wlm=22, chunk=50000, note="stream, window 4 MiB at windowLogMax 22")
    wide28 = with_window_log(z.compress(corpus.text_like(706, 20 * KB), 3, False, content_size=False), 28)
    add("st_wlog28_default", wide28, 20 * KB, stream_wlm=0, chunk=7, note="stream, window 256 MiB over the default limit -> windowTooLarge")
    add("st_wlog28_max28", wide28, 20 * KB, stream_wlm=28, chunk=7, note="stream, window 256 MiB allowed")
    add("os_wlog28", wide28, 20 * KB, note="one-shot: no window limit")
    known = z.compress(corpus.text_like(707, 600 * KB), 3, True)
    add("st_known_max19_onecall", known, 600 * KB, stream_wlm=19, note="stream, content size known, whole frame in one call with room for it: decoded in one pass, the window limit does not apply")
    add("st_known_max19_chunks", known, 600 * KB, stream_wlm=19, chunk=65536, note="stream, same frame in chunks: window 600 KiB over windowLogMax 19 -> windowTooLarge")
    add("st_ml_text", ml_text, len(text), fmt=1, stream_wlm=0, note="stream, magicless")
    add("st_ml_text_chunks", ml_text + ml_bin, len(text) + len(binr), fmt=1, stream_wlm=0, chunk=3001, note="stream, two magicless frames in chunks")
    add("st_ck_bad_ignore", bytes(bad), len(text), ignore=1, stream_wlm=0, chunk=4096, note="stream, damaged trailer ignored")
    add("st_ck_bad_validate", bytes(bad), len(text), ignore=0, stream_wlm=0, chunk=4096, note="stream, damaged trailer -> checksum_wrong")

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"{len(manifest['frames'])} fixtures, {total / 1e6:.2f} MB")
    for e in manifest["frames"]:
        print(e["name"], e.get("error", "ok"), e.get("size", ""))


if __name__ == "__main__":
    main()
