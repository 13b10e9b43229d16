"""Parity of the HIP decoder with the CPU oracle, through the C ABI.  Bit-exact: byte, integer and
index work only.  Expected results come from the oracle run on the same input and from the golden
manifest (libzstd 1.4.8's verdicts)."""
import hashlib

import numpy as np
import pytest

from conftest import DIVERGENT_FRAMES, DIVERGENT_IDS, FRAMES, FRAME_IDS, read_divergent_frame, read_frame, read_dict, regenerate, verdict_of
from oracle import oracle
from zstd_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dctx():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    d = api.DCtx(0)
    yield d
    d.close()


def dev(arr: np.ndarray):
    import torch
    return torch.from_numpy(arr).cuda()


# ------------------------------------------------------------------ one-shot drop-ins
@pytest.mark.parametrize("e", FRAMES, ids=FRAME_IDS)
def test_oneshot_matches_oracle_and_manifest(dctx, e):
    fr = read_frame(e)
    d = read_dict(e.get("dict"))
    want, werr = oracle.decompress(fr, e["capacity"], d)
    try:
        if d is not None:
            got = api.ZSTD_decompress_usingDict(dctx, fr, e["capacity"], d)
        else:
            got = api.ZSTD_decompressDCtx(dctx, fr, e["capacity"])
        gerr = 0
    except api.ZstdError as ex:
        got, gerr = None, ex.code
    assert gerr == werr == e.get("error", 0)
    if not gerr:
        assert got == want
        assert hashlib.sha256(got).hexdigest() == e["sha256"]


@pytest.mark.parametrize("e", DIVERGENT_FRAMES, ids=DIVERGENT_IDS)
def test_divergent_fixtures_follow_zstd_148(dctx, e):
    """Damaged frames on which zstd 1.4.8, the port as written and a single-symbol-only decoder disagree
    (tests/golden/divergent/): the GPU follows 1.4.8 -- stale-container reads of a dry sequence stream, the
    double-symbol decoder's last-symbol rule -- like the oracle."""
    fr = read_divergent_frame(e)
    try:
        got = verdict_of(api.ZSTD_decompressDCtx(dctx, fr, e["capacity"]), 0)
    except api.ZstdError as ex:
        got = verdict_of(None, ex.code)
    assert got == e["build"]
    assert got == verdict_of(*oracle.decompress(fr, e["capacity"]))


def test_zstd_decompress_creates_its_own_context():
    e = next(x for x in FRAMES if x["name"] == "text4096_l3_c1")
    assert api.ZSTD_decompress(read_frame(e), e["capacity"]) == regenerate(e["source"])


def test_oneshot_input_beyond_the_fixed_scratch(dctx):
    """The one-shot drop-ins decode small inputs inside a fixed scratch area; what does not fit it (here: more blocks than
    its block array holds, and more sequences than its record area) is found on the device and decoded again as a batch,
    with the same result as the oracle's."""
    import struct
    body = b"".join(struct.pack("<I", (0 << 1) | (1 if i == 1499 else 0))[:3] for i in range(1500))       # 1500 empty raw blocks
    fr = struct.pack("<I", 0xFD2FB528) + bytes([0x20, 0x00]) + body
    want, werr = oracle.decompress(fr, 16)
    assert werr == 0 and want == b""
    assert api.ZSTD_decompressDCtx(dctx, fr, 16) == b""
    e = next(x for x in FRAMES if x["name"] == "manyseq_longcount_l19")
    want, werr = oracle.decompress(read_frame(e), e["capacity"])
    assert werr == 0
    assert api.ZSTD_decompressDCtx(dctx, read_frame(e), e["capacity"]) == want
    # and small inputs afterwards still take the fixed area
    e = next(x for x in FRAMES if x["name"] == "text4096_l3_c1")
    assert api.ZSTD_decompressDCtx(dctx, read_frame(e), e["capacity"]) == regenerate(e["source"])


def test_config1_round_trip(dctx):
    """BASELINE config 1: level-1 frame of a 1 MB text buffer decodes to the original bytes."""
    e = next(x for x in FRAMES if x["name"] == "text1m_l1")
    assert api.ZSTD_decompressDCtx(dctx, read_frame(e), e["capacity"]) == regenerate(e["source"])


def test_ddict_reuse(dctx):
    d = read_dict("dict16k")
    dd = api.DDict(dctx, d)
    for e in FRAMES:
        if e.get("dict") == "dict16k" and "error" not in e:
            want, _ = oracle.decompress(read_frame(e), e["capacity"], d)
            assert api.ZSTD_decompress_usingDDict(dctx, read_frame(e), e["capacity"], dd) == want
    dd.close()


# ------------------------------------------------------------------ batched path on device buffers
def run_batch(dctx, entries, ddict=None, via_descriptors=False):
    frames = [read_frame(e) for e in entries]
    caps = [e["capacity"] for e in entries]
    src, fin, dst_bytes = api.pack_frames(frames, caps)
    d_src = dev(src)
    import torch
    d_dst = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
    b = api.Batch(dctx, src, fin, ddict, via_descriptors=via_descriptors)
    b.decode(d_src.data_ptr(), d_dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = b.status()
    out = d_dst.cpu().numpy()
    return b, d_src, fin, st, out


@pytest.mark.parametrize("via_descriptors", [False, True])
def test_batch_mixed_valid_and_corrupt(dctx, via_descriptors):
    """One launch over every dictionary-less single-frame fixture, valid and corrupt together: each
    frame gets its own verdict, a corrupt neighbour does not disturb a valid frame."""
    entries = [e for e in FRAMES if "dict" not in e and e["capacity"] <= (1 << 20)
               and not e["name"].startswith(("concat", "skippable", "rec_nodict")) and e["csize"] > 0]   # a batch entry is exactly one frame
    b, d_src, fin, st, out = run_batch(dctx, entries, via_descriptors=via_descriptors)
    for i, e in enumerate(entries):
        want, werr = oracle.decompress(read_frame(e), e["capacity"])
        assert int(st[i]["errorCode"]) == werr, e["name"]
        if not werr:
            o = int(fin[i]["dstOff"])
            assert int(st[i]["regenSize"]) == len(want), e["name"]
            assert out[o:o + len(want)].tobytes() == want, e["name"]
    b.close()


def test_sharded_decode_and_gather(dctx):
    """The frame-sharded shape of the multi-GPU configuration on one GPU: every dictionary-less single-frame fixture,
    valid and damaged, partitioned by compressed + regenerated bytes into four shards (sharding.partition_frames);
    every shard decoded as a batch of its own; the per-frame verdicts gathered back into frame order.  Frames are
    independent (zstd_decompressf.pas:1386-1408), so every frame's verdict and bytes equal the oracle's whatever
    shard it lands in, and the shards are disjoint and cover every frame."""
    from zstd_amd import sharding
    entries = [e for e in FRAMES if "dict" not in e and e["capacity"] <= (1 << 20)
               and not e["name"].startswith(("concat", "skippable", "rec_nodict")) and e["csize"] > 0]
    assert len(entries) >= 150
    costs = [e["csize"] + e["capacity"] for e in entries]
    shards = sharding.partition_frames(costs, 4)
    assert sorted(i for s in shards for i in s) == list(range(len(entries)))       # disjoint and covering
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(costs)                                  # largest-first greedy bound
    gathered = [None] * len(entries)         # (errorCode, regenSize, bytes) by frame, as a host-side gather would hold them
    for s in shards:
        b, d_src, fin, st, out = run_batch(dctx, [entries[i] for i in s])
        for k, i in enumerate(s):
            o, n = int(fin[k]["dstOff"]), int(st[k]["regenSize"])
            gathered[i] = (int(st[k]["errorCode"]), n, out[o:o + n].tobytes())
        b.close()
    for i, e in enumerate(entries):
        want, werr = oracle.decompress(read_frame(e), e["capacity"])
        code, n, got = gathered[i]
        assert code == werr, e["name"]
        if not werr:
            assert n == len(want) and got == want, e["name"]


def test_batch_with_dictionary(dctx):
    d = read_dict("dict16k")
    dd = api.DDict(dctx, d)
    entries = [e for e in FRAMES if e.get("dict") == "dict16k"]
    b, d_src, fin, st, out = run_batch(dctx, entries, ddict=dd)
    for i, e in enumerate(entries):
        want, werr = oracle.decompress(read_frame(e), e["capacity"], d)
        assert int(st[i]["errorCode"]) == werr
        o = int(fin[i]["dstOff"])
        assert out[o:o + len(want)].tobytes() == want
    b.close()
    dd.close()


def test_batch_of_dictionary_records_takes_the_shared_table_kernels(dctx):
    """A batch whose every block takes its Huffman and sequence tables from the dictionary (small records) is decoded by
    the kernel variants that keep one copy of the dictionary's tables per workgroup; valid records and records with a
    damaged payload byte (headers intact, so the batch still qualifies), verdicts and bytes as the oracle's."""
    d = read_dict("dict16k")
    dd = api.DDict(dctx, d)
    recs = [e for e in FRAMES if e.get("dict") == "dict16k" and e["name"].startswith("rec") and e["capacity"] == 4096 and "error" not in e]
    assert len(recs) >= 20
    frames, caps = [], []
    rng = np.random.default_rng(9)
    for rep in range(12):                       # 12 x 25 records: more than one workgroup of either kernel
        for e in recs:
            fr = bytearray(read_frame(e))
            if rep % 3 == 2:                    # damage somewhere in the second half: block payload, not the headers
                fr[int(rng.integers(len(fr) // 2, len(fr)))] ^= int(rng.integers(1, 256))
            frames.append(bytes(fr)); caps.append(e["capacity"])
    import torch
    src, fin, dst_bytes = api.pack_frames(frames, caps)
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
    for via in (False, True):
        b = api.Batch(dctx, None, fin, dd, dev_src_ptr=d_src.data_ptr(), dev_src_bytes=src.nbytes) if via else api.Batch(dctx, src, fin, dd)
        assert api.lib().zstdgpu_batch_kernelVariants(b.h) == 3
        d_dst.zero_()
        b.decode(d_src.data_ptr(), d_dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
        st = b.status().copy()
        out = d_dst.cpu().numpy()
        b.close()
        nbad = 0
        for i, fr in enumerate(frames):
            want, werr = oracle.decompress(fr, caps[i], d)
            assert int(st[i]["errorCode"]) == werr, (via, i)
            nbad += werr != 0
            if not werr:
                o = int(fin[i]["dstOff"])
                assert int(st[i]["regenSize"]) == len(want) and out[o:o + len(want)].tobytes() == want, (via, i)
        assert 0 < nbad < len(frames)
    dd.close()


def test_batch_checksum_value(dctx):
    entries = [e for e in FRAMES if e["name"] in ("text128k_l19_c1", "bin512k_l3_c1", "empty_c1", "text33_l3_c1")]
    b, d_src, fin, st, out = run_batch(dctx, entries)
    for i, e in enumerate(entries):
        want, _ = oracle.decompress(read_frame(e), e["capacity"])
        assert int(st[i]["xxh64Low32"]) == oracle.xxh64(want) & 0xFFFFFFFF
    b.close()


# ------------------------------------------------------------------ stage level
@pytest.mark.parametrize("name", ["text128k_l3_c0", "bin128k_l19_c1", "text1m_l1", "manyseq_longcount_l19", "alpha16_l19",
                                  "text300k_l19_c1", "farmatch_l19", "rle300k", "text255_l3_c1", "text4096_l3_c1"])
def test_stage_intermediates_match_oracle_trace(dctx, name):
    """Literals (stage 0) and sequences with resolved repeat offsets (stage 1) of every compressed
    block equal what ZSTD_decodeLiteralsBlock / ZSTD_decodeSequence produce in the oracle."""
    e = next(x for x in FRAMES if x["name"] == name)
    t = oracle.decompress_traced(read_frame(e), e["capacity"])
    assert t["err"] == 0
    b, d_src, fin, st, out = run_batch(dctx, [e])
    assert int(st[0]["errorCode"]) == 0
    fds, bds, nb = api.build_descriptors(*api.pack_frames([read_frame(e)], [e["capacity"]])[:2])
    comp = [i for i in range(nb) if bds[i].blockType == 2]
    assert len(comp) == len(t["blk_lit_size"])
    lpos = spos = 0
    for k, bi in enumerate(comp):
        info = b.block_info(bi)
        assert info["error"] == 0
        assert info["litSize"] == int(t["blk_lit_size"][k])
        assert info["nbSeq"] == int(t["blk_nb_seq"][k])
        assert info["regen"] == int(t["blk_regen"][k])
        lit = b.literals(d_src.data_ptr(), bi)
        assert lit == t["lits"][lpos:lpos + info["litSize"]].tobytes()
        seqs = b.sequences(bi)
        assert np.array_equal(seqs, t["seqs"][spos:spos + info["nbSeq"]])
        lpos += info["litSize"]
        spos += info["nbSeq"]
    b.close()


def test_predefined_tables_equal_the_reference_constants(dctx):
    """The reference-held known answers (tests/golden/ref_default_tables.json): the sequence kernel's predefined tables,
    built by its own description parser and cooperative table builder, equal the reference's literal cells."""
    import json
    import os
    from conftest import GOLD
    ref = json.load(open(os.path.join(GOLD, "ref_default_tables.json")))
    got = api.predefined_tables(dctx)
    assert got[0] == ref["LL_defaultDTable"]["cells"]
    assert got[1] == ref["OF_defaultDTable"]["cells"]
    assert got[2] == ref["ML_defaultDTable"]["cells"]


def test_xxh64_stage(dctx):
    rng = np.random.default_rng(5)
    lens = list(range(0, 70)) + [127, 128, 129, 1000, 4095, 4096, 4097, 131072, 1 << 20]
    # around the multiples of the kernel's read-ahead (8, 16 or 32 stripes of 32 bytes), where its loops hand over
    for stripes in (7, 8, 9, 15, 16, 17, 18, 31, 32, 33, 34, 47, 48, 49, 63, 64, 65, 66, 95, 96, 97, 128, 129, 2048, 2049):
        lens += [32 * stripes - 1, 32 * stripes, 32 * stripes + 1, 32 * stripes + 13]
    blob = rng.integers(0, 256, size=sum(lens) + 8, dtype=np.uint8)
    offs = np.cumsum([0] + lens[:-1])
    d = dev(blob)
    got = api.xxh64_batch(dctx, d.data_ptr(), offs, lens)
    for o, n, g in zip(offs, lens, got):
        assert int(g) == oracle.xxh64(blob[o:o + n].tobytes()), n


# ------------------------------------------------------------------ larger shapes: size-independent properties
def test_many_frames_round_trip(dctx):
    """256 independent 128 KB frames (the benchmark shape at 1/16 scale): compress -> batched decode
    is the identity, checked through XXH64 of every output against XXH64 of the generator's bytes."""
    from zstd_amd import corpus
    if not corpus.SystemZstd.available():
        pytest.skip("no system libzstd to compress with")
    import torch
    z = corpus.SystemZstd()
    n = 256
    plain = [corpus.mixed(1000 + i, 131072) for i in range(n)]
    frames = [z.compress(p, (1, 3, 19)[i % 3] if i % 16 == 0 else 3, checksum=bool(i & 1)) for i, p in enumerate(plain)]
    src, fin, dst_bytes = api.pack_frames(frames, [131072] * n)
    d_src = dev(src)
    d_dst = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
    b = api.Batch(dctx, src, fin)
    for _ in range(2):    # a batch is reusable: second launch gives the same result
        d_dst.zero_()
        b.decode(d_src.data_ptr(), d_dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
        st = b.status()
        assert not st["errorCode"].any()
        assert (st["regenSize"] == 131072).all()
        got = api.xxh64_batch(dctx, d_dst.data_ptr(), fin["dstOff"], st["regenSize"])
        want = [oracle.xxh64(p) for p in plain]
        assert [int(x) for x in got] == want
    out = d_dst.cpu().numpy()
    for i in (0, 1, 100, 255):
        o = int(fin[i]["dstOff"])
        assert out[o:o + 131072].tobytes() == plain[i]
    b.close()


def test_truncated_capacity_and_isolated_errors(dctx):
    """dstCapacity one byte short fails with dstSize_tooSmall and writes nothing past its slot."""
    e = next(x for x in FRAMES if x["name"] == "text128k_l3_c0")
    fr = read_frame(e)
    import torch
    frames, caps = [fr, fr, fr], [131072, 131071, 131072]
    src, fin, dst_bytes = api.pack_frames(frames, caps)
    d_src = dev(src)
    d_dst = torch.full((dst_bytes,), 0xAA, dtype=torch.uint8, device="cuda")
    b = api.Batch(dctx, src, fin)
    b.decode(d_src.data_ptr(), d_dst.data_ptr(), 0)
    st = b.status()
    assert [int(x) for x in st["errorCode"]] == [0, 70, 0]
    out = d_dst.cpu().numpy()
    want = regenerate(e["source"])
    assert out[0:131072].tobytes() == want
    o2 = int(fin[2]["dstOff"])
    assert out[o2:o2 + 131072].tobytes() == want
    o1 = int(fin[1]["dstOff"])
    assert (out[o1 + 131071:o2] == 0xAA).all()
    b.close()


# ------------------------------------------------------------------ object lifetimes
def test_destroy_without_status_then_next_batch(dctx):
    """A batch destroyed right after its (asynchronous) launch hands its big buffers to the next batch: the second
    batch, launched on another stream, must still decode correctly, and so must the first one's output."""
    import torch
    big = [e for e in FRAMES if e["name"] in ("text1m_l1", "bin512k_l3_c1", "text300k_l19_c1")]
    frames = [read_frame(e) for e in big] * 4
    caps = [e["capacity"] for e in big] * 4
    src, fin, dst_bytes = api.pack_frames(frames, caps)
    d_src = dev(src)
    want = [oracle.decompress(f, c)[0] for f, c in zip(frames[:3], caps[:3])] * 4
    outs = []
    for k in range(3):
        s = torch.cuda.Stream()
        d_dst = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        b = api.Batch(dctx, src, fin)
        b.decode(d_src.data_ptr(), d_dst.data_ptr(), s.cuda_stream)
        b.close()                      # no status call, no wait by the caller
        outs.append(d_dst)
    torch.cuda.synchronize()
    for d_dst in outs:
        out = d_dst.cpu().numpy()
        for i, w in enumerate(want):
            o = int(fin[i]["dstOff"])
            assert out[o:o + len(w)].tobytes() == w


def test_context_closed_before_its_children():
    """zstdgpu_destroy with a dictionary and a stream still alive: freeing them afterwards is harmless, and the orphaned
    stream refuses work."""
    import torch
    d = api.DCtx(0)
    dd = api.DDict(d, read_dict("dict16k"))
    zs = api.DStream(d)
    e = next(x for x in FRAMES if x["name"] == "text4096_l3_c1")
    src, fin, dst_bytes = api.pack_frames([read_frame(e)] * 3, [e["capacity"]] * 3)
    b = api.Batch(d, src, fin)
    d.close()
    with pytest.raises(api.ZstdError) as ex:
        zs.decompress_stream(b"\x28\xb5\x2f\xfd", 16)
    assert ex.value.code == 60
    zs.close()
    dd.close()
    # a batch outlives its context too: it still decodes, and frees what it holds itself
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
    b.decode(d_src.data_ptr(), d_dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
    st = b.status()
    assert [int(x) for x in st["errorCode"]] == [0, 0, 0]
    assert d_dst[: e["size"]].cpu().numpy().tobytes() == regenerate(e["source"])
    b.close()


def test_init_stream_cannot_swap_the_dictionary_under_a_busy_stream(dctx):
    e = next(x for x in FRAMES if x["name"] == "text4096_l3_c1")
    fr = read_frame(e)
    dd = api.DDict(dctx, read_dict("dict16k"))
    a, b = api.DStream(dctx), api.DStream(dctx)
    r, used, out = a.decompress_stream(fr[:20], 1 << 16)       # stream a is now inside a frame
    assert r > 0 and used == 20
    assert api.lib().zstdgpu_isError(api.lib().zstdgpu_initDStream_usingDDict(b.h, dd.h))
    assert api.lib().zstdgpu_getErrorCode(api.lib().zstdgpu_initDStream_usingDDict(b.h, dd.h)) == 60
    assert not api.lib().zstdgpu_isError(api.lib().zstdgpu_initDStream(b.h))      # same (no) dictionary: allowed
    r, used, out2 = a.decompress_stream(fr[20:], 1 << 16)
    assert r == 0 and out + out2 == regenerate(e["source"])
    a.close(); b.close(); dd.close()


def test_stream_header_cannot_make_the_stream_allocate_its_content_size(dctx):
    """A 20-byte frame that declares 1 GiB of content in a small window, fed in two pieces (so the one-pass shortcut
    is off): the stream sizes its output by the blocks that arrived, and reports the damaged frame."""
    import struct
    fr = struct.pack("<I", 0xFD2FB528) + bytes([0x80, 0x00]) + struct.pack("<I", 1 << 30) + bytes([0x01, 0x00, 0x00])
    zs = api.DStream(dctx)
    r, used, out = zs.decompress_stream(fr[:8], 1 << 16)
    assert r > 0
    with pytest.raises(api.ZstdError) as ex:
        zs.decompress_stream(fr[8:], 1 << 16)
    assert ex.value.code == 20
    zs.close()


# ------------------------------------------------------------------ descriptors built on the device
def _both_walks(dctx, frames, caps, ddict=None, extra_in=None):
    """The same input through the host walk and the device walk: descriptor digests, verdicts and bytes must agree."""
    import torch
    src, fin, dst_bytes = api.pack_frames(frames, caps)
    if extra_in is not None:
        fin = np.concatenate([fin, np.array(extra_in, dtype=api.FRAME_IN_DTYPE)])
    d_src = dev(src)
    bh = api.Batch(dctx, src, fin, ddict)
    bd = api.Batch(dctx, None, fin, ddict, dev_src_ptr=d_src.data_ptr(), dev_src_bytes=src.nbytes)
    assert bd.n_blocks == bh.n_blocks
    assert bd.src_bytes == bh.src_bytes
    assert bd.descriptor_digest == bh.descriptor_digest
    outs = []
    for b in (bh, bd):
        d_dst = torch.zeros(max(dst_bytes, 1), dtype=torch.uint8, device="cuda")
        b.decode(d_src.data_ptr(), d_dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
        outs.append((b.status().copy(), d_dst.cpu().numpy()))
        b.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    return fin, outs[1]


def test_device_walk_equals_host_walk_on_every_fixture(dctx):
    """Every fixture, valid or corrupt, multi-frame and skippable ones included (as single entries they are errors for
    both walks), plus entries that point outside the source buffer."""
    entries = [e for e in FRAMES if "dict" not in e and e["capacity"] <= (1 << 20)]
    frames = [read_frame(e) for e in entries]
    caps = [e["capacity"] for e in entries]
    total = sum(len(f) for f in frames)
    extra = [(total + 5, 10, 0, 100), (0, total + 1, 0, 100), (total, 0, 0, 0), (3, 4, 0, 16)]
    fin, (st, out) = _both_walks(dctx, frames, caps, extra_in=extra)
    for i, e in enumerate(entries):
        if e["name"].startswith(("concat", "skippable", "rec_nodict")) or e["csize"] == 0:
            continue
        want, werr = oracle.decompress(read_frame(e), e["capacity"])
        assert int(st[i]["errorCode"]) == werr, e["name"]
        if not werr:
            o = int(fin[i]["dstOff"])
            assert out[o:o + len(want)].tobytes() == want, e["name"]
    assert [int(x) for x in st["errorCode"][len(entries):]] == [72, 72, 72, 72]


@pytest.mark.parametrize("kernel", ["one", "eight"])
def test_every_fixture_through_each_execute_kernel(dctx, kernel):
    """The two execute kernels (one wave per frame, eight waves per frame) are chosen by the shape of the batch; here each
    decodes every fixture, valid or damaged, with and without the dictionary: verdicts and bytes as the oracle's."""
    import torch
    dctx.set_execute_kernel(kernel)
    try:
        for dict_name in (None, "dict16k"):
            entries = [e for e in FRAMES if e.get("dict") == dict_name and e["capacity"] <= (1 << 20) and e["csize"] > 0
                       and not e["name"].startswith(("concat", "skippable", "rec_nodict"))]
            if not entries:
                continue
            dictionary = read_dict(dict_name) if dict_name else None
            ddict = api.DDict(dctx, dictionary) if dictionary else None
            frames = [read_frame(e) for e in entries]
            src, fin, dst_bytes = api.pack_frames(frames, [e["capacity"] for e in entries])
            d_src = torch.from_numpy(src).cuda()
            d_dst = torch.zeros(max(dst_bytes, 1), dtype=torch.uint8, device="cuda")
            b = api.Batch(dctx, src, fin, ddict)
            b.decode(d_src.data_ptr(), d_dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
            st = b.status().copy()
            out = d_dst.cpu().numpy()
            b.close()
            for i, e in enumerate(entries):
                want, werr = oracle.decompress(frames[i], e["capacity"], dictionary)
                assert int(st[i]["errorCode"]) == werr, (kernel, e["name"])
                if not werr:
                    o = int(fin[i]["dstOff"])
                    assert int(st[i]["regenSize"]) == len(want), (kernel, e["name"])
                    assert out[o:o + len(want)].tobytes() == want, (kernel, e["name"])
            if ddict:
                ddict.close()
    finally:
        dctx.set_execute_kernel("auto")


def test_batch_launched_on_two_streams_then_destroyed(dctx):
    """A batch's scratch belongs to one launch at a time: a decode on another stream than the last one waits for that one
    (an event of the batch), and destroying the batch right after an asynchronous decode waits for it too -- its buffers
    go back to the context, where the next batch picks them up."""
    import torch
    entries = [e for e in FRAMES if e["name"] in ("text128k_l3_c0", "bin128k_l19_c1", "text1m_l1", "alpha16_l19", "text4096_l3_c1")]
    frames = [read_frame(e) for e in entries]
    src, fin, dst_bytes = api.pack_frames(frames, [e["capacity"] for e in entries])
    d_src = torch.from_numpy(src).cuda()
    want = [oracle.decompress(f, e["capacity"])[0] for f, e in zip(frames, entries)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        b = api.Batch(dctx, src, fin, None)
        d1 = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
        d2 = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        b.decode(d_src.data_ptr(), d1.data_ptr(), s1.cuda_stream)
        b.decode(d_src.data_ptr(), d2.data_ptr(), s2.cuda_stream)       # no wait in between
        b.close()                                                        # no wait before
        torch.cuda.synchronize()
        for d in (d1, d2):
            out = d.cpu().numpy()
            for i, w in enumerate(want):
                o = int(fin[i]["dstOff"])
                assert out[o:o + len(w)].tobytes() == w


def test_execute_kernel_knob_bounds(dctx):
    with pytest.raises(api.ZstdError) as ex:
        api._check(api.lib().zstdgpu_test_setExecuteKernel(dctx.h, 3))
    assert ex.value.code == 42
    dctx.set_execute_kernel("auto")


def test_device_walk_with_dictionary(dctx):
    d = read_dict("dict16k")
    dd = api.DDict(dctx, d)
    entries = [e for e in FRAMES if e.get("dict") == "dict16k"] + [e for e in FRAMES if e["name"] in ("text4096_l3_c1", "text128k_l3_c0")]
    fin, (st, out) = _both_walks(dctx, [read_frame(e) for e in entries], [e["capacity"] for e in entries], ddict=dd)
    for i, e in enumerate(entries):
        want, werr = oracle.decompress(read_frame(e), e["capacity"], d)
        assert int(st[i]["errorCode"]) == werr, e["name"]
        o = int(fin[i]["dstOff"])
        assert out[o:o + len(want)].tobytes() == want
    dd.close()


def test_device_walk_on_damaged_headers(dctx):
    """Frames with bytes overwritten in their first 40 bytes (frame header, first block header, section headers),
    truncated frames and pure noise: whatever the host walk makes of them, the device walk makes the same."""
    rng = np.random.default_rng(77)
    base = [read_frame(e) for e in FRAMES if e["name"] in ("text4096_l3_c1", "text128k_l3_c0", "bin128k_l19_c1", "text1m_l1", "rle300k", "text255_l3_c1")]
    frames = []
    for fr in base:
        for _ in range(60):
            a = bytearray(fr)
            for _ in range(int(rng.integers(1, 4))):
                a[int(rng.integers(0, min(40, len(a))))] = int(rng.integers(0, 256))
            frames.append(bytes(a))
        for cut in (1, 4, 5, 8, 9, 12, 20, len(fr) // 2, len(fr) - 1, len(fr) - 4):
            frames.append(fr[:cut])
    frames += [rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in rng.integers(1, 300, size=40)]
    frames += [b"\x28\xb5\x2f\xfd" + rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes() for n in rng.integers(1, 300, size=60)]
    _both_walks(dctx, frames, [1 << 20] * len(frames))


def test_device_walk_empty_batch(dctx):
    b = api.Batch(dctx, None, np.zeros(0, dtype=api.FRAME_IN_DTYPE), dev_src_ptr=0, dev_src_bytes=0)
    assert b.n_blocks == 0
    b.decode(0, 0, 0)
    assert len(b.status()) == 0
    b.close()


def test_device_walk_many_blocks(dctx):
    """More blocks than one scan tile holds (1024), of mixed kinds, so that offsets and job lists carry across tiles."""
    names = ("text4096_l3_c1", "text255_l3_c1", "rle300k", "text1m_l1", "empty_c1", "bin128k_l19_c1", "cor_b_flip63")
    base = [next(e for e in FRAMES if e["name"] == n) for n in names]
    entries = [base[i % len(base)] for i in range(2600)]
    fin, (st, out) = _both_walks(dctx, [read_frame(e) for e in entries], [e["capacity"] for e in entries])
    want = {e["name"]: oracle.decompress(read_frame(e), e["capacity"]) for e in base}
    for i in (0, 1, 2, 3, 4, 5, 6, 1023, 1024, 1025, 2047, 2048, 2599):
        w, werr = want[entries[i]["name"]]
        assert int(st[i]["errorCode"]) == werr
        if not werr:
            o = int(fin[i]["dstOff"])
            assert out[o:o + len(w)].tobytes() == w


# ------------------------------------------------------------------ streaming over the batched path
def _stream_all(zs, data: bytes, in_chunk: int, out_room: int):
    """Feed `data` in pieces of in_chunk bytes, draining with out_room bytes of room per call.  Returns the output and
    the return value of every call that ended a frame (0) or the error raised."""
    out = bytearray()
    pos, zeros = 0, 0
    guard = 0
    last = None
    while True:
        piece = data[pos:pos + in_chunk]
        r, used, got = zs.decompress_stream(piece, out_room)
        out += got
        pos += used
        last = r
        if r == 0:
            zeros += 1
        guard += 1
        assert guard < 2_000_000
        if pos >= len(data) and (r == 0 or (used == 0 and not got)):
            break
    return bytes(out), zeros, last


@pytest.mark.parametrize("in_chunk,out_room", [(1 << 20, 1 << 21), (4096, 1000), (7, 65536), (1, 1 << 20), (100000, 3)])
def test_stream_matches_oneshot(dctx, in_chunk, out_room):
    names = ["text4096_l3_c1", "text128k_l3_c0", "bin128k_l19_c1", "empty_c1", "text255_l3_c1", "rle300k", "text33_l3_c1"]
    if in_chunk >= 4096:
        names.append("text1m_l1")
    zs = api.DStream(dctx)
    for n in names:
        if out_room == 3 and n in ("rle300k", "text128k_l3_c0", "bin128k_l19_c1"):
            continue      # (hundreds of thousands of calls: nothing new to see)
        e = next(x for x in FRAMES if x["name"] == n)
        want, werr = oracle.decompress(read_frame(e), e["capacity"])
        assert werr == 0
        got, zeros, last = _stream_all(zs, read_frame(e), in_chunk, out_room)
        assert got == want, n
        assert zeros == 1 and last == 0, n
    zs.close()


def test_stream_concatenated_and_skippable_frames(dctx):
    """Frames back to back in one stream, a skippable frame between them: one 0 per frame, output concatenated."""
    es = [next(x for x in FRAMES if x["name"] == n) for n in ("text4096_l3_c1", "text255_l3_c1", "bin128k_l19_c1")]
    skip = b"\x5a\x2a\x4d\x18" + (11).to_bytes(4, "little") + b"hello world"
    data = read_frame(es[0]) + skip + read_frame(es[1]) + read_frame(es[2])
    want = b"".join(oracle.decompress(read_frame(e), e["capacity"])[0] for e in es)
    zs = api.DStream(dctx)
    for in_chunk in (1 << 20, 1000, 13):
        got, zeros, last = _stream_all(zs, data, in_chunk, 50000)
        assert got == want
        assert zeros == 4 and last == 0
    zs.close()


def test_stream_errors_and_reset(dctx):
    zs = api.DStream(dctx)
    e = next(x for x in FRAMES if x["name"] == "text4096_l3_c1")
    fr = read_frame(e)
    # not a zstd frame
    with pytest.raises(api.ZstdError) as ei:
        zs.decompress_stream(b"abcdefgh", 100)
    assert ei.value.code == 10
    with pytest.raises(api.ZstdError):       # the error sticks
        zs.decompress_stream(fr, 100)
    zs.reset()
    # every corrupted fixture gives the one-shot verdict once its last byte is in (or hangs on for more input, when the
    # damage makes the frame look longer than it is)
    for c in [x for x in FRAMES if "error" in x and x["error"] != 70 and "dict" not in x and x["csize"] > 0 and x["capacity"] <= (1 << 20)][:60]:
        if c["name"].startswith(("concat", "skippable")):
            continue      # (a stream ends its first frame before it meets what follows)
        data = read_frame(c)
        _, werr = oracle.decompress(data, c["capacity"])
        zs.reset()
        try:
            r, used, got = zs.decompress_stream(data, c["capacity"] + 64)
            assert r != 0 or werr == 0, c["name"]        # an unfinished frame asks for more
        except api.ZstdError as ex:
            assert werr != 0, c["name"]
            assert ex.code == werr or ex.code in (10, 14, 16, 20, 72), c["name"]
    # a half-fed frame is dropped by a reset, and the stream works again
    zs.reset()
    r, used, got = zs.decompress_stream(fr[:100], 100)
    assert r > 0 and used == 100 and got == b""
    zs.reset()
    got, zeros, last = _stream_all(zs, fr, 1 << 20, 1 << 20)
    assert got == oracle.decompress(fr, e["capacity"])[0] and last == 0
    zs.close()


def test_stream_with_dictionary(dctx):
    d = read_dict("dict16k")
    dd = api.DDict(dctx, d)
    zs = api.DStream(dctx, dd)
    for e in [x for x in FRAMES if x.get("dict") == "dict16k" and "error" not in x][:6]:
        want, _ = oracle.decompress(read_frame(e), e["capacity"], d)
        got, zeros, last = _stream_all(zs, read_frame(e), 500, 4096)
        assert got == want and zeros == 1 and last == 0
    zs.close()
    dd.close()


# ------------------------------------------------------------------ decoder parameters
from conftest import PARAM_FRAMES, PARAM_IDS, read_param_frame  # noqa: E402


def _with_params(dctx, e):
    dctx.reset()
    dctx.set_parameter(api.ZSTD_d_format, e["format"])
    dctx.set_parameter(api.ZSTD_d_forceIgnoreChecksum, e["ignore_checksum"])


@pytest.mark.parametrize("e", [x for x in PARAM_FRAMES if "stream_chunk" not in x], ids=[x["name"] for x in PARAM_FRAMES if "stream_chunk" not in x])
def test_parameters_oneshot(dctx, e):
    """ZSTD_d_format and ZSTD_d_forceIgnoreChecksum on the one-shot entry points: oracle and libzstd's recorded verdict."""
    fr = read_param_frame(e)
    d = read_dict(e.get("dict"))
    want, werr = oracle.decompress(fr, e["capacity"], d, magicless=bool(e["format"]), ignore_checksum=bool(e["ignore_checksum"]))
    try:
        _with_params(dctx, e)
        try:
            got = api.ZSTD_decompress_usingDict(dctx, fr, e["capacity"], d) if d is not None else api.ZSTD_decompressDCtx(dctx, fr, e["capacity"])
            gerr = 0
        except api.ZstdError as ex:
            got, gerr = None, ex.code
    finally:
        dctx.reset()
    assert gerr == werr == e.get("error", 0)
    if not gerr:
        assert got == want
        assert hashlib.sha256(got).hexdigest() == e["sha256"]


@pytest.mark.parametrize("e", [x for x in PARAM_FRAMES if "stream_chunk" in x], ids=[x["name"] for x in PARAM_FRAMES if "stream_chunk" in x])
def test_parameters_stream(dctx, e):
    """The same parameters and ZSTD_d_windowLogMax on a stream fed the way the fixture was made (whole, or in chunks)."""
    fr = read_param_frame(e)
    try:
        _with_params(dctx, e)
        if e["stream_window_log_max"]:
            dctx.set_parameter(api.ZSTD_d_windowLogMax, e["stream_window_log_max"])
        zs = api.DStream(dctx)
        try:
            got, zeros, last = _stream_all(zs, fr, e["stream_chunk"] or len(fr), e["capacity"])
            gerr = 0
        except api.ZstdError as ex:
            got, gerr = None, ex.code
        zs.close()
    finally:
        dctx.reset()
    assert gerr == e.get("error", 0)
    if not gerr:
        assert len(got) == e["size"] and hashlib.sha256(got).hexdigest() == e["sha256"]


def test_parameters_device_walk(dctx):
    """Batches pick the context's format and checksum policy up; the device walk builds the same descriptors."""
    for fmt, ign in ((1, 0), (0, 1), (1, 1)):
        es = [x for x in PARAM_FRAMES if "stream_chunk" not in x and "dict" not in x and x["format"] == fmt and x["ignore_checksum"] == ign
              and "concat" not in x["name"] and "skippable" not in x["name"] and x["name"] != "ml_cut0"]
        assert es
        try:
            dctx.reset()
            dctx.set_parameter(api.ZSTD_d_format, fmt)
            dctx.set_parameter(api.ZSTD_d_forceIgnoreChecksum, ign)
            fin, (st, out) = _both_walks(dctx, [read_param_frame(e) for e in es], [e["capacity"] for e in es])
        finally:
            dctx.reset()
        for i, e in enumerate(es):
            assert int(st[i]["errorCode"]) == e.get("error", 0), e["name"]
            if "error" not in e:
                o = int(fin[i]["dstOff"])
                assert hashlib.sha256(out[o:o + e["size"]].tobytes()).hexdigest() == e["sha256"], e["name"]


def test_parameter_api(dctx):
    """Bounds, defaults, out-of-range values, unknown parameters, stage checks and resets (zstd_decompressf.pas:1680-1826)."""
    dctx.reset()
    assert api.ZSTD_dParam_getBounds(api.ZSTD_d_windowLogMax) == (10, 31)
    assert api.ZSTD_dParam_getBounds(api.ZSTD_d_format) == (0, 1)
    assert api.ZSTD_dParam_getBounds(api.ZSTD_d_stableOutBuffer) == (0, 1)
    assert api.ZSTD_dParam_getBounds(api.ZSTD_d_forceIgnoreChecksum) == (0, 1)
    with pytest.raises(api.ZstdError) as ei:
        api.ZSTD_dParam_getBounds(12345)
    assert ei.value.code == 40
    assert dctx.get_parameter(api.ZSTD_d_windowLogMax) == 27 and dctx.get_parameter(api.ZSTD_d_format) == 0
    for p, v in ((api.ZSTD_d_windowLogMax, 9), (api.ZSTD_d_windowLogMax, 32), (api.ZSTD_d_format, 2), (api.ZSTD_d_forceIgnoreChecksum, -1)):
        with pytest.raises(api.ZstdError) as ei:
            dctx.set_parameter(p, v)
        assert ei.value.code == 42
    with pytest.raises(api.ZstdError) as ei:
        dctx.set_parameter(777, 1)
    assert ei.value.code == 40
    dctx.set_parameter(api.ZSTD_d_windowLogMax, 23)
    assert dctx.get_parameter(api.ZSTD_d_windowLogMax) == 23
    dctx.set_parameter(api.ZSTD_d_windowLogMax, 0)          # 0 selects the default
    assert dctx.get_parameter(api.ZSTD_d_windowLogMax) == 27
    # inside a frame, parameters and the dictionary are frozen
    e = next(x for x in FRAMES if x["name"] == "text4096_l3_c1")
    fr = read_frame(e)
    zs = api.DStream(dctx)
    zs.decompress_stream(fr[:50], 100)
    for call in (lambda: dctx.set_parameter(api.ZSTD_d_format, 1), lambda: dctx.ref_ddict(None), lambda: dctx.reset(api.ZSTD_reset_parameters)):
        with pytest.raises(api.ZstdError) as ei:
            call()
        assert ei.value.code == 60
    dctx.reset(api.ZSTD_reset_session_only)                 # drops the half-read frame
    dctx.set_parameter(api.ZSTD_d_format, 1)
    dctx.reset(api.ZSTD_reset_session_and_parameters)
    assert dctx.get_parameter(api.ZSTD_d_format) == 0
    got, zeros, last = _stream_all(zs, fr, 1000, 1 << 16)
    assert got == oracle.decompress(fr, e["capacity"])[0]
    zs.close()


def test_ref_ddict_is_used_by_decompressDCtx(dctx):
    d = read_dict("dict16k")
    dd = api.DDict(dctx, d)
    e = next(x for x in FRAMES if x.get("dict") == "dict16k" and "error" not in x)
    fr = read_frame(e)
    want, _ = oracle.decompress(fr, e["capacity"], d)
    try:
        dctx.ref_ddict(dd)
        assert api.ZSTD_decompressDCtx(dctx, fr, e["capacity"]) == want
        dctx.ref_ddict(None)
        with pytest.raises(api.ZstdError) as ei:
            api.ZSTD_decompressDCtx(dctx, fr, e["capacity"])
        assert ei.value.code == 32
        dctx.ref_ddict(dd)
        dctx.reset(api.ZSTD_reset_parameters)              # clears the dictionary too
        with pytest.raises(api.ZstdError):
            api.ZSTD_decompressDCtx(dctx, fr, e["capacity"])
    finally:
        dctx.reset()
        dd.close()


def test_stable_out_buffer(dctx):
    """ZSTD_d_stableOutBuffer: same buffer on every call of a frame, large enough for it (zstd_decompressf.pas:1913-1951)."""
    import ctypes
    e = next(x for x in FRAMES if x["name"] == "text128k_l3_c0")
    fr = read_frame(e)
    want, _ = oracle.decompress(fr, e["capacity"])
    try:
        dctx.reset()
        dctx.set_parameter(api.ZSTD_d_stableOutBuffer, 1)
        zs = api.DStream(dctx)
        buf = ctypes.create_string_buffer(len(want) + 100)
        pos, fed, r = 0, 0, 1
        while r != 0:
            r, used, pos = zs.decompress_stream_into(fr[fed:fed + 10000], buf, len(buf), pos)
            fed += used
        assert buf.raw[:pos] == want
        # a different buffer in the middle of a frame
        zs.decompress_stream_into(fr[:10000], buf, len(buf), 0)
        other = ctypes.create_string_buffer(len(want) + 100)
        with pytest.raises(api.ZstdError) as ei:
            zs.decompress_stream_into(fr[10000:20000], other, len(other), 0)
        assert ei.value.code == 104
        zs.reset()
        # too small for the frame
        small = ctypes.create_string_buffer(1000)
        with pytest.raises(api.ZstdError) as ei:
            fed, r = 0, 1
            while r != 0:
                r, used, _ = zs.decompress_stream_into(fr[fed:fed + 50000], small, len(small), 0)
                fed += used
        assert ei.value.code == 70
        zs.close()
    finally:
        dctx.reset()


# ------------------------------------------------------------------ differential fuzz (tools/fuzz.py), a short run
@pytest.mark.parametrize("corrupt,with_dict", [(False, False), (True, False), (False, True), (True, True)],
                         ids=["valid", "damaged", "dictionary", "dictionary-damaged"])
def test_fuzz_short_run(dctx, corrupt, with_dict):
    """Seeded inputs of nine shapes, sizes 1 B .. 300 KB, levels -5 .. 19, random window sizes and flags, compressed with
    the system libzstd: decoded in one batch through both descriptor walks (and, damaged, frame by frame through the
    one-shot entry point) -- bytes by XXH64 against the plain input, verdicts against the oracle."""
    from zstd_amd import corpus
    if not corpus.SystemZstd.available():
        pytest.skip("no system libzstd to compress with")
    import argparse
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("zg_fuzz", os.path.join(ROOT, "tools", "fuzz.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    args = argparse.Namespace(rounds=2, frames=300, seed=4 + int(corrupt) + 2 * int(with_dict), max_kb=300, corrupt=corrupt, dict=with_dict)
    assert fuzz.run(args, dctx) == 0
