"""The entry point that allocates nothing (zstdgpu_decode_frames_noalloc) against the batch path it is a variant of
(zstdgpu_batch_create_on_device + zstdgpu_batch_decode): same verdicts, same bytes -- bit-exact, byte and integer work
only -- on two calls in a row, the second with a smaller input, and the limits are refused."""
import numpy as np
import pytest

from conftest import FRAMES, read_frame, read_dict
from zstd_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dctx():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    d = api.DCtx(0)
    yield d
    d.close()


def _reference(dctx, src, fin, dst_bytes, ddict=None):
    """The existing entry points: a batch built on the device from the same device buffers."""
    import torch
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.full((dst_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
    b = api.Batch(dctx, None, fin, ddict, dev_src_ptr=d_src.data_ptr(), dev_src_bytes=src.nbytes)
    b.decode(d_src.data_ptr(), d_dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = b.status().copy()
    out = d_dst.cpu().numpy()
    b.close()
    return st, out


def _noalloc(dctx, src, fin, dst_bytes, ddict=None):
    import torch
    d_src = torch.from_numpy(src).cuda()
    d_fin = torch.from_numpy(fin.view(np.uint8).copy()).cuda()
    d_dst = torch.full((dst_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
    d_st = torch.full((max(len(fin), 1) * api.FRAME_STATUS_DTYPE.itemsize,), 0xEE, dtype=torch.uint8, device="cuda")
    import gc
    gc.collect()        # (what the collector frees during the call would show as a change too)
    torch.cuda.synchronize()
    # free bytes of the device (hipMemGetInfo) see the library's own hipMalloc, which torch's allocator counters do not
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    api.decode_frames_noalloc(dctx, d_src.data_ptr(), src.nbytes, d_fin.data_ptr(), len(fin), d_dst.data_ptr(), d_st.data_ptr(),
                              ddict, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before
    st = d_st.cpu().numpy().view(api.FRAME_STATUS_DTYPE)[: len(fin)].copy()
    return st, d_dst.cpu().numpy()


def _entries(pred):
    return [e for e in FRAMES if "dict" not in e and e["csize"] > 0 and pred(e)
            and not e["name"].startswith(("concat", "skippable", "rec_nodict"))]


def _compare(dctx, entries, ddict=None):
    frames = [read_frame(e) for e in entries]
    caps = [e["capacity"] for e in entries]
    src, fin, dst_bytes = api.pack_frames(frames, caps)
    assert src.nbytes <= api.NOALLOC_MAX_SRC_BYTES and len(fin) <= api.NOALLOC_MAX_FRAMES
    want_st, want_out = _reference(dctx, src, fin, dst_bytes, ddict)
    got_st, got_out = _noalloc(dctx, src, fin, dst_bytes, ddict)
    assert np.array_equal(got_st, want_st)
    assert np.array_equal(got_out, want_out)
    return want_st


def test_noalloc_two_calls_second_smaller(dctx):
    """A mixed batch of valid and damaged frames (several blocks, checksums, every literal and table mode), then a
    smaller one: the second call must not see anything of the first in the shared scratch."""
    # within the entry point's limits: valid and damaged fixtures of at most 128 KB, about 1 MiB of output of each kind
    pick = []
    for damaged in (False, True):
        total = 0
        for e in sorted(_entries(lambda e: e["capacity"] <= (1 << 17) and ("error" in e) == damaged), key=lambda e: e["capacity"]):
            if total + e["capacity"] > (1 << 20) or len(pick) >= 200:
                continue
            pick.append(e)
            total += e["capacity"]
    st = _compare(dctx, pick)
    assert (st["errorCode"] == 0).any() and (st["errorCode"] != 0).any()
    assert not (st["errorCode"] == 64).any()          # the batch fits the scratch: no frame is refused for room
    small = [e for e in pick if "error" not in e][::5][:5]
    assert len(small) >= 3
    st2 = _compare(dctx, small)
    assert (st2["errorCode"] == 0).all()


def test_noalloc_with_dictionary(dctx):
    d = read_dict("dict16k")
    dd = api.DDict(dctx, d)
    entries = [e for e in FRAMES if e.get("dict") == "dict16k"]
    _compare(dctx, entries, dd)
    dd.close()


def test_noalloc_under_stream_capture(dctx):
    """The advertised use: the call is recorded into a HIP graph (where an allocation is an error that ends the
    capture) and the graph replayed twice into a destination refilled in between; verdicts and bytes as the batch path's."""
    import torch
    entries = [e for e in _entries(lambda e: e["capacity"] <= (1 << 16))][:40]
    assert len(entries) >= 20
    frames = [read_frame(e) for e in entries]
    src, fin, dst_bytes = api.pack_frames(frames, [e["capacity"] for e in entries])
    want_st, want_out = _reference(dctx, src, fin, dst_bytes)
    d_src = torch.from_numpy(src).cuda()
    d_fin = torch.from_numpy(fin.view(np.uint8).copy()).cuda()
    d_dst = torch.full((dst_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
    d_st = torch.full((len(fin) * api.FRAME_STATUS_DTYPE.itemsize,), 0xEE, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):       # (capture mode "global": any allocating HIP call on any thread fails the capture)
        api.decode_frames_noalloc(dctx, d_src.data_ptr(), src.nbytes, d_fin.data_ptr(), len(fin), d_dst.data_ptr(), d_st.data_ptr(),
                                  None, torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        d_dst.fill_(0x5A)
        d_st.fill_(0xEE)
        g.replay()
        torch.cuda.synchronize()
        st = d_st.cpu().numpy().view(api.FRAME_STATUS_DTYPE)[: len(fin)]
        assert np.array_equal(st, want_st)
        assert np.array_equal(d_dst.cpu().numpy(), want_out)


def test_noalloc_limits(dctx):
    import torch
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = buf.data_ptr()
    with pytest.raises(api.ZstdError) as ex:
        api.decode_frames_noalloc(dctx, p, 1 << 10, p, api.NOALLOC_MAX_FRAMES + 1, p, p)
    assert ex.value.code == 42
    with pytest.raises(api.ZstdError) as ex:
        api.decode_frames_noalloc(dctx, p, api.NOALLOC_MAX_SRC_BYTES + 1, p, 1, p, p)
    assert ex.value.code == 42


def test_noalloc_contents_beyond_the_scratch(dctx):
    """More blocks than the block array holds: known only on the device, where every frame of the call is refused
    with memory_allocation and nothing is written."""
    import struct
    import torch
    # a frame of 1100 empty raw blocks: magic, descriptor (single segment, content size 0), block headers
    body = b"".join(struct.pack("<I", (0 << 1) | (1 if i == 1099 else 0))[:3] for i in range(1100))
    fr = struct.pack("<I", 0xFD2FB528) + bytes([0x20, 0x00]) + body
    src, fin, dst_bytes = api.pack_frames([fr, fr], [16, 16])
    got_st, got_out = _noalloc(dctx, src, fin, dst_bytes)
    assert [int(x) for x in got_st["errorCode"]] == [64, 64]
    assert (got_out == 0x5A).all()
    # and the scratch is usable again afterwards
    e = next(x for x in FRAMES if x["name"] == "text1000_l3_c1")
    st = _compare(dctx, [e])
    assert int(st[0]["errorCode"]) == 0
