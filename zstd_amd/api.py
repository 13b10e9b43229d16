"""Host-side mirror of the reference's decoder interface over the C ABI of libzstdgpu.so.

Function names and argument meaning follow decompress/zstd_decompressf.pas (ZSTD_decompress,
ZSTD_decompressDCtx, ZSTD_decompress_usingDDict, ZSTD_getFrameContentSize,
ZSTD_findFrameCompressedSize, ZSTD_createDDict ...).  Everything here is plumbing: it loads the
shared library, marshals pointers and sizes, and raises `ZstdError` with the ZSTD_ErrorCode the
library returned.  There is no CPU fallback: without the built library (or without a GPU) the
calls fail loudly.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZSTDGPU_LIB") or os.path.join(_HERE, "libzstdgpu.so")   # ZSTDGPU_LIB: a diagnostic build of the same library

CONTENTSIZE_UNKNOWN = (1 << 64) - 1
CONTENTSIZE_ERROR = (1 << 64) - 2
NUM_STAGES = 6
STAGE_NAMES = ("literals", "sequences", "layout", "execute", "checksum", "total")

ERROR_NAMES = {
    0: "no_error", 1: "GENERIC", 10: "prefix_unknown", 12: "version_unsupported", 14: "frameParameter_unsupported",
    16: "frameParameter_windowTooLarge", 20: "corruption_detected", 22: "checksum_wrong", 30: "dictionary_corrupted",
    32: "dictionary_wrong", 64: "memory_allocation", 70: "dstSize_tooSmall", 72: "srcSize_wrong",
}


class ZstdError(Exception):
    def __init__(self, code: int):
        self.code = code
        super().__init__(f"zstd error {code} ({ERROR_NAMES.get(code, '?')})")


class FrameIn(ctypes.Structure):
    _fields_ = [("srcOff", ctypes.c_uint64), ("srcSize", ctypes.c_uint64), ("dstOff", ctypes.c_uint64), ("dstCapacity", ctypes.c_uint64)]


class FrameDesc(ctypes.Structure):
    _fields_ = [("srcOff", ctypes.c_uint64), ("srcSize", ctypes.c_uint64), ("dstOff", ctypes.c_uint64), ("dstCapacity", ctypes.c_uint64),
                ("contentSize", ctypes.c_uint64), ("windowSize", ctypes.c_uint64), ("firstBlock", ctypes.c_uint32),
                ("nBlocks", ctypes.c_uint32), ("dictID", ctypes.c_uint32), ("checksumFlag", ctypes.c_uint32),
                ("checksum", ctypes.c_uint32), ("hostError", ctypes.c_uint32)]


class BlockDesc(ctypes.Structure):
    _fields_ = [("srcOff", ctypes.c_uint64), ("cSize", ctypes.c_uint32), ("origSize", ctypes.c_uint32), ("frameIdx", ctypes.c_uint32),
                ("blockType", ctypes.c_uint8), ("lastBlock", ctypes.c_uint8), ("hostError", ctypes.c_uint16)]


FRAME_STATUS_DTYPE = np.dtype([("errorCode", np.uint32), ("xxh64Low32", np.uint32), ("regenSize", np.uint64)])
FRAME_IN_DTYPE = np.dtype([("srcOff", np.uint64), ("srcSize", np.uint64), ("dstOff", np.uint64), ("dstCapacity", np.uint64)])

_lib = None

# every symbol include/zstdgpu.h declares
EXPORTS = (
    "zstdgpu_isError", "zstdgpu_getErrorCode", "zstdgpu_getErrorName", "zstdgpu_versionNumber",
    "zstdgpu_create", "zstdgpu_destroy",
    "zstdgpu_getFrameContentSize", "zstdgpu_findFrameCompressedSize", "zstdgpu_getDictID_fromDict", "zstdgpu_getDictID_fromFrame",
    "zstdgpu_isFrame", "zstdgpu_frameHeaderSize", "zstdgpu_getFrameHeader", "zstdgpu_getDecompressedSize", "zstdgpu_findDecompressedSize",
    "zstdgpu_decompressBound",
    "zstdgpu_createDDict", "zstdgpu_freeDDict",
    "zstdgpu_decompress", "zstdgpu_decompressDCtx", "zstdgpu_decompress_usingDict", "zstdgpu_decompress_usingDDict",
    "zstdgpu_buildDescriptors", "zstdgpu_batch_create", "zstdgpu_batch_create_from_frames", "zstdgpu_batch_create_on_device",
    "zstdgpu_batch_descriptorDigest", "zstdgpu_batch_destroy",
    "zstdgpu_createDStream", "zstdgpu_freeDStream", "zstdgpu_initDStream", "zstdgpu_initDStream_usingDDict", "zstdgpu_resetDStream",
    "zstdgpu_decompressStream",
    "zstdgpu_dParam_getBounds", "zstdgpu_DCtx_setParameter", "zstdgpu_DCtx_getParameter", "zstdgpu_DCtx_refDDict", "zstdgpu_DCtx_reset",
    "zstdgpu_batch_decode", "zstdgpu_batch_status", "zstdgpu_batch_nFrames", "zstdgpu_batch_nBlocks", "zstdgpu_batch_srcBytes",
    "zstdgpu_batch_scratchBytes", "zstdgpu_batch_set_profiling", "zstdgpu_batch_stage_ms", "zstdgpu_stage_kernel_name",
    "zstdgpu_decode_frames_noalloc",
    "zstdgpu_xxh64Batch", "zstdgpu_predefinedTables", "zstdgpu_batch_blockInfo", "zstdgpu_batch_copyLiterals", "zstdgpu_batch_copySequences",
    "zstdgpu_test_setExecuteKernel", "zstdgpu_batch_kernelVariants",
    "zstdgpu_decompressBegin", "zstdgpu_decompressBegin_usingDict", "zstdgpu_decompressBegin_usingDDict",
    "zstdgpu_nextSrcSizeToDecompress", "zstdgpu_nextInputType", "zstdgpu_decompressContinue",
)


class FrameHeader(ctypes.Structure):
    """zstdgpu_frameHeader (ZSTD_frameHeader)."""
    _fields_ = [("frameContentSize", ctypes.c_ulonglong), ("windowSize", ctypes.c_ulonglong), ("blockSizeMax", ctypes.c_uint),
                ("frameType", ctypes.c_uint), ("headerSize", ctypes.c_uint), ("dictID", ctypes.c_uint), ("checksumFlag", ctypes.c_uint)]


class Bounds(ctypes.Structure):
    """zstdgpu_bounds (ZSTD_bounds)."""
    _fields_ = [("error", ctypes.c_size_t), ("lowerBound", ctypes.c_int), ("upperBound", ctypes.c_int)]


# zstdgpu_dParameter / zstdgpu_ResetDirective
ZSTD_d_windowLogMax, ZSTD_d_format, ZSTD_d_stableOutBuffer, ZSTD_d_forceIgnoreChecksum = 100, 1000, 1001, 1002
ZSTD_f_zstd1, ZSTD_f_zstd1_magicless = 0, 1
ZSTD_reset_session_only, ZSTD_reset_parameters, ZSTD_reset_session_and_parameters = 1, 2, 3


def lib():
    """Load libzstdgpu.so and declare its prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = ctypes.CDLL(LIB_PATH)
    sz, vp, u64, u32, ci = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    P = ctypes.POINTER

    def proto(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    proto("zstdgpu_isError", ctypes.c_uint, sz)
    proto("zstdgpu_getErrorCode", ctypes.c_uint, sz)
    proto("zstdgpu_getErrorName", ctypes.c_char_p, sz)
    proto("zstdgpu_versionNumber", ctypes.c_uint)
    proto("zstdgpu_create", vp, ci)
    proto("zstdgpu_destroy", None, vp)
    proto("zstdgpu_getFrameContentSize", ctypes.c_ulonglong, vp, sz)
    proto("zstdgpu_findFrameCompressedSize", sz, vp, sz)
    proto("zstdgpu_isFrame", ctypes.c_uint, vp, sz)
    proto("zstdgpu_frameHeaderSize", sz, vp, sz)
    proto("zstdgpu_getFrameHeader", sz, P(FrameHeader), vp, sz)
    proto("zstdgpu_getDecompressedSize", ctypes.c_ulonglong, vp, sz)
    proto("zstdgpu_findDecompressedSize", ctypes.c_ulonglong, vp, sz)
    proto("zstdgpu_decompressBound", ctypes.c_ulonglong, vp, sz)
    proto("zstdgpu_getDictID_fromDict", ctypes.c_uint, vp, sz)
    proto("zstdgpu_getDictID_fromFrame", ctypes.c_uint, vp, sz)
    proto("zstdgpu_createDDict", vp, vp, vp, sz)
    proto("zstdgpu_freeDDict", None, vp)
    proto("zstdgpu_decompress", sz, vp, sz, vp, sz)
    proto("zstdgpu_decompressDCtx", sz, vp, vp, sz, vp, sz)
    proto("zstdgpu_decompress_usingDict", sz, vp, vp, sz, vp, sz, vp, sz)
    proto("zstdgpu_decompress_usingDDict", sz, vp, vp, sz, vp, sz, vp)
    proto("zstdgpu_buildDescriptors", sz, vp, sz, vp, u32, vp, vp, sz)
    proto("zstdgpu_batch_create", vp, vp, vp, sz, vp, u32, vp, u32, vp)
    proto("zstdgpu_batch_create_from_frames", vp, vp, vp, sz, vp, u32, vp)
    proto("zstdgpu_batch_create_on_device", vp, vp, vp, sz, vp, u32, vp)
    proto("zstdgpu_batch_descriptorDigest", ctypes.c_uint64, vp)
    proto("zstdgpu_createDStream", vp, vp)
    proto("zstdgpu_freeDStream", None, vp)
    proto("zstdgpu_initDStream", sz, vp)
    proto("zstdgpu_initDStream_usingDDict", sz, vp, vp)
    proto("zstdgpu_resetDStream", sz, vp)
    proto("zstdgpu_decompressStream", sz, vp, vp, vp)
    proto("zstdgpu_dParam_getBounds", Bounds, ci)
    proto("zstdgpu_DCtx_setParameter", sz, vp, ci, ci)
    proto("zstdgpu_DCtx_getParameter", sz, vp, ci, P(ci))
    proto("zstdgpu_DCtx_refDDict", sz, vp, vp)
    proto("zstdgpu_DCtx_reset", sz, vp, ci)
    proto("zstdgpu_batch_destroy", None, vp)
    proto("zstdgpu_batch_decode", sz, vp, vp, vp, vp)
    proto("zstdgpu_batch_status", sz, vp, vp)
    proto("zstdgpu_decode_frames_noalloc", sz, vp, vp, sz, vp, u32, vp, vp, vp, vp)
    proto("zstdgpu_batch_nFrames", u32, vp)
    proto("zstdgpu_batch_nBlocks", u32, vp)
    proto("zstdgpu_batch_srcBytes", u64, vp)
    proto("zstdgpu_batch_scratchBytes", u64, vp)
    proto("zstdgpu_batch_set_profiling", None, vp, ci)
    proto("zstdgpu_batch_stage_ms", sz, vp, P(ctypes.c_float))
    proto("zstdgpu_stage_kernel_name", ctypes.c_char_p, ci)
    proto("zstdgpu_xxh64Batch", sz, vp, vp, vp, vp, u32, vp)
    proto("zstdgpu_predefinedTables", sz, vp, vp)
    proto("zstdgpu_batch_blockInfo", sz, vp, u32, P(u32), P(u32), P(u64), P(u32))
    proto("zstdgpu_batch_copyLiterals", sz, vp, vp, u32, vp, sz)
    proto("zstdgpu_batch_copySequences", sz, vp, u32, vp, sz)
    proto("zstdgpu_test_setExecuteKernel", sz, vp, ci)
    proto("zstdgpu_batch_kernelVariants", u32, vp)
    proto("zstdgpu_decompressBegin", sz, vp)
    proto("zstdgpu_decompressBegin_usingDict", sz, vp, vp, sz)
    proto("zstdgpu_decompressBegin_usingDDict", sz, vp, vp)
    proto("zstdgpu_nextSrcSizeToDecompress", sz, vp)
    proto("zstdgpu_nextInputType", ctypes.c_uint, vp)
    proto("zstdgpu_decompressContinue", sz, vp, vp, sz, vp, sz)
    _lib = L
    return L


def ZSTD_dParam_getBounds(param: int):
    """Returns (lowerBound, upperBound); raises ZstdError for an unknown parameter."""
    b = lib().zstdgpu_dParam_getBounds(param)
    _check(b.error)
    return b.lowerBound, b.upperBound


def ZSTD_isError(r: int) -> bool:
    return bool(lib().zstdgpu_isError(r))


def ZSTD_getErrorCode(r: int) -> int:
    return lib().zstdgpu_getErrorCode(r)


def _check(r: int) -> int:
    L = lib()
    if L.zstdgpu_isError(r):
        raise ZstdError(L.zstdgpu_getErrorCode(r))
    return r


# --------------------------------------------------------------------------- host-side queries
def ZSTD_getFrameContentSize(src: bytes) -> int:
    return lib().zstdgpu_getFrameContentSize(src, len(src))


def ZSTD_findFrameCompressedSize(src: bytes) -> int:
    return _check(lib().zstdgpu_findFrameCompressedSize(src, len(src)))


def ZSTD_isFrame(src: bytes) -> int:
    return lib().zstdgpu_isFrame(src, len(src))


def ZSTD_frameHeaderSize(src: bytes) -> int:
    return _check(lib().zstdgpu_frameHeaderSize(src, len(src)))


def ZSTD_getFrameHeader(src: bytes):
    """Returns (0, dict of the header fields), or (bytes wanted, None); raises ZstdError for an invalid header."""
    h = FrameHeader()
    r = _check(lib().zstdgpu_getFrameHeader(ctypes.byref(h), src, len(src)))
    return (0, {f: int(getattr(h, f)) for f, _ in FrameHeader._fields_}) if r == 0 else (r, None)


def ZSTD_getDecompressedSize(src: bytes) -> int:
    return lib().zstdgpu_getDecompressedSize(src, len(src))


def ZSTD_findDecompressedSize(src: bytes) -> int:
    return lib().zstdgpu_findDecompressedSize(src, len(src))


def ZSTD_decompressBound(src: bytes) -> int:
    return lib().zstdgpu_decompressBound(src, len(src))


def ZSTD_getDictID_fromDict(d: bytes) -> int:
    return lib().zstdgpu_getDictID_fromDict(d, len(d))


def ZSTD_getDictID_fromFrame(src: bytes) -> int:
    return lib().zstdgpu_getDictID_fromFrame(src, len(src))


# --------------------------------------------------------------------------- context / dictionary
class DCtx:
    """ZSTD_DCtx stand-in: one per GPU, not re-entrant."""

    def __init__(self, device: int = 0):
        self.h = lib().zstdgpu_create(device)
        if not self.h:
            raise RuntimeError(f"zstdgpu_create({device}) failed: no usable GPU")
        self.device = device

    # ZSTD_DCtx_setParameter / getParameter / refDDict / reset (decompress/zstd_decompressf.pas:1641-1826)
    def set_parameter(self, param: int, value: int):
        _check(lib().zstdgpu_DCtx_setParameter(self.h, param, value))

    def get_parameter(self, param: int) -> int:
        v = ctypes.c_int(0)
        _check(lib().zstdgpu_DCtx_getParameter(self.h, param, ctypes.byref(v)))
        return v.value

    def ref_ddict(self, ddict):
        self._ref = ddict      # keep it alive
        _check(lib().zstdgpu_DCtx_refDDict(self.h, ddict.h if ddict else None))

    def reset(self, directive: int = ZSTD_reset_session_and_parameters):
        _check(lib().zstdgpu_DCtx_reset(self.h, directive))

    EXECUTE_KERNELS = {"auto": 0, "one": 1, "eight": 2}

    def set_execute_kernel(self, kernel: str):
        """Tests and measurements: send every frame of later launches to one of the two execute kernels
        ("one" wave per frame, "eight" waves per frame), or leave the choice to the library ("auto")."""
        _check(lib().zstdgpu_test_setExecuteKernel(self.h, self.EXECUTE_KERNELS[kernel]))

    # bufferless, one block per call (decompress/zstd_decompressf.pas:995-1246, :1386-1452); results are raw size_t values
    # (a count or an error, test with ZSTD_isError) as in the reference, because a caller acts on either
    def decompress_begin(self, dictionary: bytes = None, ddict=None) -> int:
        L = lib()
        if ddict is not None:
            return L.zstdgpu_decompressBegin_usingDDict(self.h, ddict.h)
        if dictionary:
            return L.zstdgpu_decompressBegin_usingDict(self.h, dictionary, len(dictionary))
        return L.zstdgpu_decompressBegin(self.h)

    def next_src_size_to_decompress(self) -> int:
        return lib().zstdgpu_nextSrcSizeToDecompress(self.h)

    def next_input_type(self) -> int:
        return lib().zstdgpu_nextInputType(self.h)

    def decompress_continue(self, dst_addr: int, dst_capacity: int, src: bytes) -> int:
        return lib().zstdgpu_decompressContinue(self.h, dst_addr, dst_capacity, src, len(src))

    def close(self):
        if self.h:
            lib().zstdgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DDict:
    """ZSTD_DDict stand-in: dictionary copied to the GPU, entropy tables digested there once."""

    def __init__(self, dctx: DCtx, dictionary: bytes):
        self.dctx = dctx
        self.h = lib().zstdgpu_createDDict(dctx.h, dictionary, len(dictionary))
        if not self.h:
            raise ZstdError(64)

    def close(self):
        if self.h:
            lib().zstdgpu_freeDDict(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------- one-shot drop-ins (host buffers)
def ZSTD_decompressDCtx(dctx: DCtx, src: bytes, dstCapacity: int) -> bytes:
    out = ctypes.create_string_buffer(max(dstCapacity, 1))
    r = _check(lib().zstdgpu_decompressDCtx(dctx.h, out, dstCapacity, src, len(src)))
    return out.raw[:r]


def ZSTD_decompress(src: bytes, dstCapacity: int) -> bytes:
    out = ctypes.create_string_buffer(max(dstCapacity, 1))
    r = _check(lib().zstdgpu_decompress(out, dstCapacity, src, len(src)))
    return out.raw[:r]


def ZSTD_decompress_usingDict(dctx: DCtx, src: bytes, dstCapacity: int, dictionary: bytes) -> bytes:
    out = ctypes.create_string_buffer(max(dstCapacity, 1))
    r = _check(lib().zstdgpu_decompress_usingDict(dctx.h, out, dstCapacity, src, len(src), dictionary, len(dictionary or b"")))
    return out.raw[:r]


def ZSTD_decompress_usingDDict(dctx: DCtx, src: bytes, dstCapacity: int, ddict: DDict) -> bytes:
    out = ctypes.create_string_buffer(max(dstCapacity, 1))
    r = _check(lib().zstdgpu_decompress_usingDDict(dctx.h, out, dstCapacity, src, len(src), ddict.h if ddict else None))
    return out.raw[:r]


# --------------------------------------------------------------------------- batches (device buffers)
def build_descriptors(host_src: np.ndarray, frames_in: np.ndarray):
    """C mirror of the host descriptor builder.  Returns (FrameDesc array, BlockDesc array)."""
    L = lib()
    n = len(frames_in)
    fin = np.ascontiguousarray(frames_in, dtype=FRAME_IN_DTYPE)
    fds = (FrameDesc * max(n, 1))()
    nb = _check(L.zstdgpu_buildDescriptors(host_src.ctypes.data, host_src.nbytes, fin.ctypes.data, n, fds, None, 0))
    bds = (BlockDesc * max(nb, 1))()
    nb2 = _check(L.zstdgpu_buildDescriptors(host_src.ctypes.data, host_src.nbytes, fin.ctypes.data, n, fds, bds, nb))
    assert nb2 == nb
    return fds, bds, nb


class Batch:
    """A prepared batch of independent frames: descriptors and scratch live on the GPU, `decode`
    launches the stage chain on device buffers and allocates nothing."""

    def __init__(self, dctx: DCtx, host_src: np.ndarray, frames_in: np.ndarray, ddict: DDict = None, via_descriptors: bool = False,
                 dev_src_ptr: int = 0, dev_src_bytes: int = 0):
        """host_src: the concatenated frames in host memory (walked on the host), or None with dev_src_ptr / dev_src_bytes:
        the same bytes in device memory, walked by device kernels."""
        L = lib()
        self.dctx, self.ddict = dctx, ddict
        self.n = len(frames_in)
        fin = np.ascontiguousarray(frames_in, dtype=FRAME_IN_DTYPE)
        dd = ddict.h if ddict else None
        if host_src is None:
            self.h = L.zstdgpu_batch_create_on_device(dctx.h, dev_src_ptr, dev_src_bytes, fin.ctypes.data, self.n, dd)
            if not self.h:
                raise ZstdError(64)
            return
        host_src = np.ascontiguousarray(host_src, dtype=np.uint8)
        if via_descriptors:
            fds, bds, nb = build_descriptors(host_src, fin)
            self.h = L.zstdgpu_batch_create(dctx.h, host_src.ctypes.data, host_src.nbytes, fds, self.n, bds, nb, dd)
        else:
            self.h = L.zstdgpu_batch_create_from_frames(dctx.h, host_src.ctypes.data, host_src.nbytes, fin.ctypes.data, self.n, dd)
        if not self.h:
            raise ZstdError(64)

    @property
    def descriptor_digest(self):
        return lib().zstdgpu_batch_descriptorDigest(self.h)

    @property
    def n_blocks(self):
        return lib().zstdgpu_batch_nBlocks(self.h)

    @property
    def src_bytes(self):
        return lib().zstdgpu_batch_srcBytes(self.h)

    @property
    def scratch_bytes(self):
        return lib().zstdgpu_batch_scratchBytes(self.h)

    def decode(self, dev_src_ptr: int, dev_dst_ptr: int, stream: int = 0):
        _check(lib().zstdgpu_batch_decode(self.h, dev_src_ptr, dev_dst_ptr, stream or None))

    def status(self) -> np.ndarray:
        st = np.zeros(max(self.n, 1), dtype=FRAME_STATUS_DTYPE)
        _check(lib().zstdgpu_batch_status(self.h, st.ctypes.data))
        return st[: self.n]

    def set_profiling(self, on: bool):
        lib().zstdgpu_batch_set_profiling(self.h, int(on))

    def stage_ms(self):
        ms = (ctypes.c_float * NUM_STAGES)()
        _check(lib().zstdgpu_batch_stage_ms(self.h, ms))
        return list(ms)

    def block_info(self, bi: int):
        L = lib()
        a, b, c, d = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64(), ctypes.c_uint32()
        _check(L.zstdgpu_batch_blockInfo(self.h, bi, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)))
        return dict(litSize=a.value, nbSeq=b.value, regen=c.value, error=d.value)

    def literals(self, dev_src_ptr: int, bi: int) -> bytes:
        out = ctypes.create_string_buffer(1 << 17)
        r = _check(lib().zstdgpu_batch_copyLiterals(self.h, dev_src_ptr, bi, out, 1 << 17))
        return out.raw[:r]

    def sequences(self, bi: int) -> np.ndarray:
        cap = 0x7F00 + 0xFFFF + 1
        out = np.zeros((cap, 3), dtype=np.uint32)
        r = _check(lib().zstdgpu_batch_copySequences(self.h, bi, out.ctypes.data, cap))
        return out[:r].copy()

    def close(self):
        if self.h:
            lib().zstdgpu_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


NOALLOC_MAX_FRAMES, NOALLOC_MAX_SRC_BYTES = 256, 4 << 20


def decode_frames_noalloc(dctx: DCtx, dev_src_ptr: int, src_bytes: int, dev_frames_ptr: int, n_frames: int,
                          dev_dst_ptr: int, dev_status_ptr: int, ddict: DDict = None, stream: int = 0):
    """zstdgpu_decode_frames_noalloc: everything in device memory, nothing allocated, nothing awaited."""
    _check(lib().zstdgpu_decode_frames_noalloc(dctx.h, dev_src_ptr, src_bytes, dev_frames_ptr, n_frames, dev_dst_ptr, dev_status_ptr,
                                               ddict.h if ddict else None, stream or None))


def xxh64_batch(dctx: DCtx, dev_base_ptr: int, offsets, lengths) -> np.ndarray:
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    out = np.zeros(len(off), dtype=np.uint64)
    _check(lib().zstdgpu_xxh64Batch(dctx.h, dev_base_ptr, off.ctypes.data, ln.ctypes.data, len(off), out.ctypes.data))
    return out


def predefined_tables(dctx: DCtx):
    """The GPU's predefined LL / OF / ML decoding tables: three lists of [nextState, nbAdditionalBits, nbBits, baseValue] cells."""
    out = np.zeros((3, 64, 4), dtype=np.uint32)
    _check(lib().zstdgpu_predefinedTables(dctx.h, out.ctypes.data))
    return [out[0].tolist(), out[1][:32].tolist(), out[2].tolist()]


def pack_frames(frames, capacities=None, align: int = 16):
    """Concatenate compressed frames into one source buffer (each frame `align`-aligned) and lay
    their outputs out back to back.  Returns (host_src uint8 array, FrameIn array, dst_bytes)."""
    n = len(frames)
    fin = np.zeros(n, dtype=FRAME_IN_DTYPE)
    so = do = 0
    for i, fr in enumerate(frames):
        cap = len(fr) if capacities is None else capacities[i]
        fin[i] = (so, len(fr), do, cap)
        so += (len(fr) + align - 1) // align * align
        do += (cap + align - 1) // align * align
    src = np.zeros(so + 64, dtype=np.uint8)
    for i, fr in enumerate(frames):
        o = int(fin[i]["srcOff"])
        src[o:o + len(fr)] = np.frombuffer(fr, dtype=np.uint8)
    return src, fin, do + 64


class _InBuffer(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


class _OutBuffer(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


class DStream:
    """ZSTD_DStream over the batched path (zstdgpu_decompressStream)."""

    def __init__(self, dctx: DCtx, ddict: DDict = None):
        self.dctx, self.ddict = dctx, ddict
        self.h = lib().zstdgpu_createDStream(dctx.h)
        if not self.h:
            raise ZstdError(64)
        _check(lib().zstdgpu_initDStream_usingDDict(self.h, ddict.h) if ddict else lib().zstdgpu_initDStream(self.h))

    def reset(self):
        _check(lib().zstdgpu_resetDStream(self.h))

    def decompress_stream(self, data: bytes, out_room: int):
        """One call of ZSTD_decompressStream: returns (return value, bytes consumed, bytes produced)."""
        src = ctypes.create_string_buffer(data, len(data)) if data else ctypes.create_string_buffer(1)
        dst = ctypes.create_string_buffer(max(out_room, 1))
        ib = _InBuffer(ctypes.cast(src, ctypes.c_void_p), len(data), 0)
        ob = _OutBuffer(ctypes.cast(dst, ctypes.c_void_p), out_room, 0)
        r = _check(lib().zstdgpu_decompressStream(self.h, ctypes.byref(ob), ctypes.byref(ib)))
        return r, ib.pos, dst.raw[:ob.pos]

    def decompress_stream_into(self, data: bytes, dst, size: int, pos: int):
        """The same into a caller-held buffer `dst` (ctypes array) at `pos`, as ZSTD_d_stableOutBuffer wants it:
        returns (return value, bytes consumed, new pos)."""
        src = ctypes.create_string_buffer(data, len(data)) if data else ctypes.create_string_buffer(1)
        ib = _InBuffer(ctypes.cast(src, ctypes.c_void_p), len(data), 0)
        ob = _OutBuffer(ctypes.cast(dst, ctypes.c_void_p), size, pos)
        r = _check(lib().zstdgpu_decompressStream(self.h, ctypes.byref(ob), ctypes.byref(ib)))
        return r, ib.pos, ob.pos

    def close(self):
        if self.h:
            lib().zstdgpu_freeDStream(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
