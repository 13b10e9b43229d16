"""Drives a bufferless decoder (ZSTD_decompressBegin / ZSTD_nextSrcSizeToDecompress / ZSTD_nextInputType /
ZSTD_decompressContinue, decompress/zstd_decompressf.pas:995-1246, :1386-1421) over one input and records what it did.

The same loop runs against three back ends: the system libzstd 1.4.8 (tools/gen_continue_fixtures.py writes its answers
to tests/golden/continue_traces.json), the CPU oracle, and the GPU library.  Test infrastructure only.
"""
import ctypes
import hashlib

MAX_STEPS = 200000


class Backend:
    """begin(dictionary) -> error code; next_size(); next_type(); cont(dst_addr, cap, src_bytes) -> (value, error code)."""


class SystemBackend(Backend):
    D_FORMAT, D_IGNORE_CHECKSUM = 1000, 1002      # ZSTD_d_format, ZSTD_d_forceIgnoreChecksum (zstd.h 1.4.8)

    def __init__(self, magicless=False, ignore_checksum=False):
        from zstd_amd import corpus
        self.L = L = corpus.SystemZstd().lib
        sz, vp = ctypes.c_size_t, ctypes.c_void_p
        L.ZSTD_decompressBegin.argtypes = [vp]; L.ZSTD_decompressBegin.restype = sz
        L.ZSTD_decompressBegin_usingDict.argtypes = [vp, vp, sz]; L.ZSTD_decompressBegin_usingDict.restype = sz
        L.ZSTD_nextSrcSizeToDecompress.argtypes = [vp]; L.ZSTD_nextSrcSizeToDecompress.restype = sz
        L.ZSTD_nextInputType.argtypes = [vp]; L.ZSTD_nextInputType.restype = ctypes.c_int
        L.ZSTD_decompressContinue.argtypes = [vp, vp, sz, vp, sz]; L.ZSTD_decompressContinue.restype = sz
        L.ZSTD_DCtx_setParameter.argtypes = [vp, ctypes.c_int, ctypes.c_int]; L.ZSTD_DCtx_setParameter.restype = sz
        self.dctx = L.ZSTD_createDCtx()
        if magicless:
            assert not L.ZSTD_isError(L.ZSTD_DCtx_setParameter(self.dctx, self.D_FORMAT, 1))
        if ignore_checksum:
            assert not L.ZSTD_isError(L.ZSTD_DCtx_setParameter(self.dctx, self.D_IGNORE_CHECKSUM, 1))
        self._dict = None

    def __del__(self):
        try:
            self.L.ZSTD_freeDCtx(self.dctx)
        except Exception:
            pass

    def _code(self, r):
        return self.L.ZSTD_getErrorCode(r) if self.L.ZSTD_isError(r) else 0

    def begin(self, dictionary):
        self._dict = ctypes.create_string_buffer(dictionary, len(dictionary)) if dictionary else None      # referenced, not copied: kept alive here
        r = self.L.ZSTD_decompressBegin_usingDict(self.dctx, self._dict, len(dictionary)) if dictionary else self.L.ZSTD_decompressBegin(self.dctx)
        return self._code(r)

    def next_size(self):
        return self.L.ZSTD_nextSrcSizeToDecompress(self.dctx)

    def next_type(self):
        return self.L.ZSTD_nextInputType(self.dctx)

    def cont(self, dst_addr, cap, src):
        r = self.L.ZSTD_decompressContinue(self.dctx, dst_addr, cap, src, len(src))
        c = self._code(r)
        return (0 if c else r), c


class OracleBackend(Backend):
    def __init__(self, magicless=False, ignore_checksum=False):
        from oracle import oracle
        self.L = L = oracle.lib()
        sz, vp = ctypes.c_size_t, ctypes.c_void_p
        L.oracle_createDCtx.restype = vp
        L.oracle_freeDCtx.argtypes = [vp]
        L.oracle_DCtx_setFormat.argtypes = [vp, ctypes.c_uint, ctypes.c_uint]
        L.oracle_decompressBegin_usingDict.argtypes = [vp, vp, sz]; L.oracle_decompressBegin_usingDict.restype = sz
        L.oracle_nextSrcSizeToDecompress.argtypes = [vp]; L.oracle_nextSrcSizeToDecompress.restype = sz
        L.oracle_nextInputType.argtypes = [vp]; L.oracle_nextInputType.restype = ctypes.c_uint
        L.oracle_decompressContinue.argtypes = [vp, vp, sz, vp, sz]; L.oracle_decompressContinue.restype = sz
        self.c = L.oracle_createDCtx()
        L.oracle_DCtx_setFormat(self.c, int(magicless), int(ignore_checksum))

    def __del__(self):
        try:
            self.L.oracle_freeDCtx(self.c)
        except Exception:
            pass

    def _code(self, r):
        return self.L.oracle_getErrorCode(r) if self.L.oracle_isError(r) else 0

    def begin(self, dictionary):
        d = dictionary or b""
        return self._code(self.L.oracle_decompressBegin_usingDict(self.c, d if d else None, len(d)))

    def next_size(self):
        return self.L.oracle_nextSrcSizeToDecompress(self.c)

    def next_type(self):
        return self.L.oracle_nextInputType(self.c)

    def cont(self, dst_addr, cap, src):
        r = self.L.oracle_decompressContinue(self.c, dst_addr, cap, src, len(src))
        c = self._code(r)
        return (0 if c else r), c


class GpuBackend(Backend):
    """The library under test, through its C ABI (zstd_amd.api is a ctypes layer)."""

    def __init__(self, dctx, use_ddict=False):
        from zstd_amd import api
        self.api, self.dctx, self.use_ddict = api, dctx, use_ddict
        self._dd = None

    def _code(self, r):
        return self.api.ZSTD_getErrorCode(r) if self.api.ZSTD_isError(r) else 0

    def begin(self, dictionary):
        if dictionary and self.use_ddict:
            if self._dd is None:
                self._dd = self.api.DDict(self.dctx, dictionary)
            return self._code(self.dctx.decompress_begin(ddict=self._dd))
        return self._code(self.dctx.decompress_begin(dictionary))

    def next_size(self):
        return self.dctx.next_src_size_to_decompress()

    def next_type(self):
        return self.dctx.next_input_type()

    def cont(self, dst_addr, cap, src):
        r = self.dctx.decompress_continue(dst_addr, cap, src)
        c = self._code(r)
        return (0 if c else r), c


def drive(be: Backend, data: bytes, capacity: int, dictionary: bytes = None, raw_chunk: int = 0, keep_steps: bool = False):
    """Decode `data` (one or more frames) into one contiguous buffer of `capacity` bytes the way a caller of the
    bufferless interface does: ask for the next size, hand over exactly that many bytes (a raw block in pieces of
    `raw_chunk` bytes when that is set; whatever is left when the input ends early), advance.  Returns a dict:
    steps, trace_sha256 (over "type,size,result;" per call), err {step, code} or None, consumed, out_size, out_sha256."""
    buf = ctypes.create_string_buffer(max(capacity, 1))
    base = ctypes.addressof(buf)
    pos, op = 0, 0
    trace = []
    err = None
    btype = None
    frames = 0
    while pos < len(data) and err is None and len(trace) < MAX_STEPS:
        c = be.begin(dictionary)
        trace.append(("B", 0, -c if c else 0))
        if c:
            err = {"step": len(trace) - 1, "code": c}
            break
        frames += 1
        while len(trace) < MAX_STEPS:
            want = be.next_size()
            typ = be.next_type()
            if want == 0:
                break
            n = want
            if raw_chunk and typ in (2, 3) and btype == 0:
                n = min(n, raw_chunk)
            n = min(n, len(data) - pos)
            piece = data[pos:pos + n]
            r, c = be.cont(base + op, capacity - op, piece)
            trace.append((typ, n, -c if c else r))
            if c:
                err = {"step": len(trace) - 1, "code": c}
                break
            if typ == 1 and n == 3:
                btype = (piece[0] >> 1) & 3
            pos += n
            op += r
            if n == 0:      # nothing left to give and the decoder still wants input
                err = {"step": len(trace) - 1, "code": 0}
                break
    text = "".join("%s,%d,%d;" % t for t in trace)
    out = buf.raw[:op]
    res = {"steps": len(trace), "trace_sha256": hashlib.sha256(text.encode()).hexdigest(), "err": err, "consumed": pos,
           "out_size": op, "out_sha256": hashlib.sha256(out).hexdigest(), "frames": frames}
    if keep_steps:
        res["trace"] = trace
        res["out"] = out
    return res


def first_difference(a, b):
    """Index and entries of the first step where two kept traces differ (for failure messages)."""
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i, x, y
    if len(a) != len(b):
        i = min(len(a), len(b))
        return i, (a[i] if i < len(a) else None), (b[i] if i < len(b) else None)
    return None
