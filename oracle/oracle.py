"""ctypes loader for the CPU oracle (test infrastructure; never on the product path)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libzstd_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        sz, vp, u64, u32 = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32
        L.oracle_decompress.argtypes = [vp, sz, vp, sz]; L.oracle_decompress.restype = sz
        L.oracle_decompress_usingDict.argtypes = [vp, sz, vp, sz, vp, sz]; L.oracle_decompress_usingDict.restype = sz
        L.oracle_decompress_advanced.argtypes = [vp, sz, vp, sz, vp, sz, ctypes.c_uint, ctypes.c_uint]; L.oracle_decompress_advanced.restype = sz
        L.oracle_decompress_traced.argtypes = [vp, sz, vp, sz, vp, sz, vp]; L.oracle_decompress_traced.restype = sz
        L.oracle_isError.argtypes = [sz]; L.oracle_isError.restype = ctypes.c_uint
        L.oracle_getErrorCode.argtypes = [sz]; L.oracle_getErrorCode.restype = ctypes.c_uint
        L.oracle_xxh64.argtypes = [vp, sz, u64]; L.oracle_xxh64.restype = u64
        L.oracle_getFrameContentSize.argtypes = [vp, sz]; L.oracle_getFrameContentSize.restype = u64
        L.oracle_findFrameCompressedSize.argtypes = [vp, sz]; L.oracle_findFrameCompressedSize.restype = sz
        L.oracle_trace_new.restype = vp
        L.oracle_trace_free.argtypes = [vp]
        for n in ("nBlocks",): getattr(L, "oracle_trace_" + n).argtypes = [vp]; getattr(L, "oracle_trace_" + n).restype = u32
        for n in ("nSeqs", "litBytes"): getattr(L, "oracle_trace_" + n).argtypes = [vp]; getattr(L, "oracle_trace_" + n).restype = u64
        for n in ("blkLitSize", "blkNbSeq", "blkRegen", "lits", "seqs"):
            getattr(L, "oracle_trace_" + n).argtypes = [vp]; getattr(L, "oracle_trace_" + n).restype = vp
        L.oracle_default_table.argtypes = [ctypes.c_int, vp, vp, vp, vp]; L.oracle_default_table.restype = ctypes.c_uint
        L.oracle_set_huf.argtypes = [ctypes.c_int]; L.oracle_set_huf.restype = None
        L.oracle_set_body.argtypes = [ctypes.c_int]; L.oracle_set_body.restype = None
        L.oracle_decompress_batch.argtypes = [vp, vp, vp, vp, vp, vp, u32]; L.oracle_decompress_batch.restype = sz
        _lib = L
    return _lib


def decompress(frame: bytes, capacity: int, dictionary: bytes = None, magicless: bool = False, ignore_checksum: bool = False):
    """Returns (bytes, 0) on success or (None, zstd error code)."""
    L = lib()
    out = ctypes.create_string_buffer(max(capacity, 1))
    d = dictionary or b""
    if magicless or ignore_checksum:
        r = L.oracle_decompress_advanced(out, capacity, frame, len(frame), d if d else None, len(d), int(magicless), int(ignore_checksum))
    else:
        r = L.oracle_decompress_usingDict(out, capacity, frame, len(frame), d if d else None, len(d))
    if L.oracle_isError(r):
        return None, L.oracle_getErrorCode(r)
    return out.raw[:r], 0


def decompress_traced(frame: bytes, capacity: int, dictionary: bytes = None):
    """Decode and also return per-compressed-block intermediates:
    dict(out, err, blk_lit_size, blk_nb_seq, blk_regen, lits, seqs[n,3])."""
    L = lib()
    out = ctypes.create_string_buffer(max(capacity, 1))
    d = dictionary or b""
    t = L.oracle_trace_new()
    try:
        r = L.oracle_decompress_traced(out, capacity, frame, len(frame), d if d else None, len(d), t)
        nb = L.oracle_trace_nBlocks(t); ns = L.oracle_trace_nSeqs(t); nl = L.oracle_trace_litBytes(t)
        def arr(ptr, n, dt):
            if not n or not ptr:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()
        res = dict(
            blk_lit_size=arr(L.oracle_trace_blkLitSize(t), nb, np.uint32),
            blk_nb_seq=arr(L.oracle_trace_blkNbSeq(t), nb, np.uint32),
            blk_regen=arr(L.oracle_trace_blkRegen(t), nb, np.uint32),
            lits=arr(L.oracle_trace_lits(t), nl, np.uint8),
            seqs=arr(L.oracle_trace_seqs(t), ns * 3, np.uint32).reshape(-1, 3),
        )
        if L.oracle_isError(r):
            res.update(out=None, err=L.oracle_getErrorCode(r))
        else:
            res.update(out=out.raw[:r], err=0)
        return res
    finally:
        L.oracle_trace_free(t)


def default_table(which: int):
    """The oracle's predefined table `which` (0 LL, 1 OF, 2 ML) and code tables:
    dict(tableLog, cells [n][4] {nextState, nbAdditionalBits, nbBits, baseValue}, base, bits, norm)."""
    cells = np.zeros((64, 4), dtype=np.uint32); base = np.zeros(53, dtype=np.uint32); bits = np.zeros(53, dtype=np.uint32); norm = np.zeros(53, dtype=np.int32)
    log = lib().oracle_default_table(which, cells.ctypes.data, base.ctypes.data, bits.ctypes.data, norm.ctypes.data)
    n = (36, 32, 53)[which]
    return dict(tableLog=log, cells=cells[: 1 << log].tolist(), base=base[:n].tolist(), bits=bits[:n].tolist(), norm=norm[: (36, 29, 53)[which]].tolist())


BODY_AS_ZSTD_148, BODY_ALWAYS_PREFETCHING, BODY_NEVER_PREFETCHING = 0, 1, 2


def set_body(policy: int):
    """Which sequence loop the oracle runs (process-wide): as zstd 1.4.8 dispatches (default, what the fixtures pin), always
    the prefetching loop (the port as written, zstd_decompress_block.pas:1541), or never."""
    lib().oracle_set_body(policy)


def set_huf(policy: int):
    """0: the Huffman decoder HUF_selectDecoder picks (default); 1: always the single-symbol decoder."""
    lib().oracle_set_huf(policy)


def xxh64(data: bytes, seed: int = 0) -> int:
    return lib().oracle_xxh64(data, len(data), seed)
