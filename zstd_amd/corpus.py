"""Seeded synthetic corpora and (optional) access to the system libzstd as a *compressor*.

Nothing here is on the decode path.  The generators are the C functions in csrc/corpus.c;
`SystemZstd` wraps libzstd.so.1 (upstream zstd, same 1.4.8 the reference was ported from)
through ctypes and is used to make compressed frames and dictionaries for fixtures and
benchmarks, and as the parity pin of the CPU oracle.  The decoder never calls it.
"""
import ctypes
import ctypes.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CORPUS_SO = os.path.join(_HERE, "_corpus.so")
_corpus = None


def _lib():
    global _corpus
    if _corpus is None:
        if not os.path.exists(_CORPUS_SO):
            raise RuntimeError(f"{_CORPUS_SO} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(_CORPUS_SO)
        for fn in (lib.corpus_text, lib.corpus_binary):
            fn.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t]
            fn.restype = None
        _corpus = lib
    return _corpus


def text_like(seed: int, n: int) -> bytes:
    buf = np.empty(n, dtype=np.uint8)
    _lib().corpus_text(seed, buf.ctypes.data, n)
    return buf.tobytes()


def binary_records(seed: int, n: int) -> bytes:
    buf = np.empty(n, dtype=np.uint8)
    _lib().corpus_binary(seed, buf.ctypes.data, n)
    return buf.tobytes()


def mixed(seed: int, n: int) -> bytes:
    """Even seeds text-like, odd seeds binary records (the 50/50 mix of the benchmark configs)."""
    return text_like(seed, n) if seed % 2 == 0 else binary_records(seed, n)


class SystemZstd:
    """ctypes view of the system libzstd (compressor side + reference decoder)."""

    # ZSTD_cParameter values of zstd 1.4.x
    C_LEVEL, C_WINDOWLOG = 100, 101
    C_CONTENTSIZE, C_CHECKSUM, C_DICTID = 200, 201, 202

    def __init__(self):
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        self.lib = L = ctypes.CDLL(name)
        L.ZSTD_versionString.restype = ctypes.c_char_p
        L.ZSTD_versionNumber.restype = ctypes.c_uint
        L.ZSTD_compressBound.restype = ctypes.c_size_t
        L.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        L.ZSTD_isError.restype = ctypes.c_uint
        L.ZSTD_isError.argtypes = [ctypes.c_size_t]
        L.ZSTD_getErrorCode.restype = ctypes.c_int
        L.ZSTD_getErrorCode.argtypes = [ctypes.c_size_t]
        L.ZSTD_createCCtx.restype = ctypes.c_void_p
        L.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
        L.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
        L.ZSTD_CCtx_loadDictionary.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        L.ZSTD_CCtx_loadDictionary.restype = ctypes.c_size_t
        L.ZSTD_CCtx_reset.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ZSTD_CCtx_reset.restype = ctypes.c_size_t
        L.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.ZSTD_compress2.restype = ctypes.c_size_t
        L.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.ZSTD_decompress.restype = ctypes.c_size_t
        L.ZSTD_decompressDCtx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.ZSTD_decompressDCtx.restype = ctypes.c_size_t
        L.ZSTD_createDCtx.restype = ctypes.c_void_p
        L.ZSTD_freeDCtx.argtypes = [ctypes.c_void_p]
        L.ZSTD_decompress_usingDict.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.ZSTD_decompress_usingDict.restype = ctypes.c_size_t
        L.ZDICT_trainFromBuffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        L.ZDICT_trainFromBuffer.restype = ctypes.c_size_t
        self.version = L.ZSTD_versionString().decode()
        self._cctx = L.ZSTD_createCCtx()
        self._dctx = L.ZSTD_createDCtx()

    @staticmethod
    def available() -> bool:
        try:
            SystemZstd()
            return True
        except OSError:
            return False

    def compress(self, data: bytes, level: int = 3, checksum: bool = False, content_size: bool = True,
                 window_log: int = 0, dictionary: bytes = None, dict_id: bool = True) -> bytes:
        L = self.lib
        L.ZSTD_CCtx_reset(self._cctx, 3)  # session and parameters
        for p, v in ((self.C_LEVEL, level), (self.C_CHECKSUM, int(checksum)), (self.C_CONTENTSIZE, int(content_size)),
                     (self.C_DICTID, int(dict_id))):
            r = L.ZSTD_CCtx_setParameter(self._cctx, p, v)
            assert not L.ZSTD_isError(r), (p, v)
        if window_log:
            r = L.ZSTD_CCtx_setParameter(self._cctx, self.C_WINDOWLOG, window_log)
            assert not L.ZSTD_isError(r)
        if dictionary:
            r = L.ZSTD_CCtx_loadDictionary(self._cctx, dictionary, len(dictionary))
            assert not L.ZSTD_isError(r)
        cap = L.ZSTD_compressBound(len(data))
        out = ctypes.create_string_buffer(cap)
        n = L.ZSTD_compress2(self._cctx, out, cap, data, len(data))
        assert not L.ZSTD_isError(n), L.ZSTD_getErrorCode(n)
        return out.raw[:n]

    def decompress(self, frame: bytes, capacity: int, dictionary: bytes = None):
        """Returns (bytes, 0) or (None, error_code) from the system decoder."""
        L = self.lib
        out = ctypes.create_string_buffer(max(capacity, 1))
        if dictionary:
            n = L.ZSTD_decompress_usingDict(self._dctx, out, capacity, frame, len(frame), dictionary, len(dictionary))
        else:
            n = L.ZSTD_decompress(out, capacity, frame, len(frame))
        if L.ZSTD_isError(n):
            return None, L.ZSTD_getErrorCode(n)
        return out.raw[:n], 0

    def train_dictionary(self, samples, capacity: int = 112640) -> bytes:
        L = self.lib
        blob = b"".join(samples)
        sizes = (ctypes.c_size_t * len(samples))(*[len(s) for s in samples])
        out = ctypes.create_string_buffer(capacity)
        n = L.ZDICT_trainFromBuffer(out, capacity, blob, sizes, len(samples))
        assert not L.ZSTD_isError(n), "ZDICT_trainFromBuffer failed"
        return out.raw[:n]
