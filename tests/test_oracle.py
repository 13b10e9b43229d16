"""The CPU oracle against the golden vectors: every fixture under tests/golden/ carries the verdict
of libzstd 1.4.8 (regenerated size + sha256, or error code); the oracle must reproduce it."""
import hashlib

import pytest

from conftest import DIVERGENT_FRAMES, DIVERGENT_IDS, read_divergent_frame, verdict_of, FRAMES, FRAME_IDS, read_frame, read_dict, regenerate
from oracle import oracle


@pytest.mark.parametrize("e", FRAMES, ids=FRAME_IDS)
def test_oracle_matches_libzstd_verdict(e):
    out, err = oracle.decompress(read_frame(e), e["capacity"], read_dict(e.get("dict")))
    if "error" in e:
        assert err == e["error"]
    else:
        assert err == 0
        assert len(out) == e["size"]
        assert hashlib.sha256(out).hexdigest() == e["sha256"]


@pytest.mark.parametrize("e", [e for e in FRAMES if "source" in e and e["capacity"] <= (1 << 20)],
                         ids=[e["name"] for e in FRAMES if "source" in e and e["capacity"] <= (1 << 20)])
def test_fixture_is_round_trip_of_generator(e):
    """compress -> decode is the identity: the decoded bytes are the seeded generator's bytes."""
    out, err = oracle.decompress(read_frame(e), e["capacity"], read_dict(e.get("dict")))
    assert err == 0 and out == regenerate(e["source"])


def test_xxh64_known_answers():
    assert oracle.xxh64(b"") == 0xEF46DB3751D8E999
    assert oracle.xxh64(b"a") == 0xD24EC4F1A98C6E5B
    assert oracle.xxh64(b"abc") == 0x44BC2CF5AD770999
    assert oracle.xxh64(b"Nobody inspects the spammish repetition") == 0xFBCEA83C8A378BF1


def test_trace_consistency():
    """Traced intermediates add up: per block litSize + sum(matchLength) == regenerated size."""
    e = next(x for x in FRAMES if x["name"] == "text300k_l19_c1")
    t = oracle.decompress_traced(read_frame(e), e["capacity"])
    assert t["err"] == 0
    pos = 0
    for lit, n, regen in zip(t["blk_lit_size"], t["blk_nb_seq"], t["blk_regen"]):
        s = t["seqs"][pos:pos + n]
        assert int(lit) + int(s[:, 1].sum()) == int(regen)
        assert int(s[:, 0].sum()) <= int(lit)
        pos += n
    assert sum(int(x) for x in t["blk_regen"]) == e["size"]


from conftest import PARAM_FRAMES, read_param_frame  # noqa: E402

_ONESHOT_PARAM = [x for x in PARAM_FRAMES if "stream_chunk" not in x]


@pytest.mark.parametrize("e", _ONESHOT_PARAM, ids=[x["name"] for x in _ONESHOT_PARAM])
def test_oracle_matches_libzstd_under_decoder_parameters(e):
    """ZSTD_d_format (magicless frames) and ZSTD_d_forceIgnoreChecksum: libzstd 1.4.8's verdict under the parameter."""
    out, err = oracle.decompress(read_param_frame(e), e["capacity"], read_dict(e.get("dict")),
                                 magicless=bool(e["format"]), ignore_checksum=bool(e["ignore_checksum"]))
    assert err == e.get("error", 0)
    if not err:
        assert len(out) == e["size"] and hashlib.sha256(out).hexdigest() == e["sha256"]


def test_predefined_tables_equal_the_reference_constants():
    """The only known-answer data the reference holds for this path: its three predefined sequence decoding tables and the
    code tables (tests/golden/ref_default_tables.json, extracted by tools/extract_ref_tables.py).  The oracle builds its
    tables from the default distributions; they must come out as the reference's literal cells."""
    import json
    import os
    from conftest import GOLD
    ref = json.load(open(os.path.join(GOLD, "ref_default_tables.json")))
    for which, name, base, bits, norm in ((0, "LL_defaultDTable", "LL_base", "LL_bits", "LL_defaultNorm"),
                                          (1, "OF_defaultDTable", "OF_base", "OF_bits", "OF_defaultNorm"),
                                          (2, "ML_defaultDTable", "ML_base", "ML_bits", "ML_defaultNorm")):
        t = oracle.default_table(which)
        assert t["tableLog"] == ref[name]["tableLog"]
        assert t["cells"] == ref[name]["cells"], name
        assert t["base"] == ref[base] and t["bits"] == ref[bits] and t["norm"] == ref[norm]


@pytest.mark.parametrize("e", DIVERGENT_FRAMES, ids=DIVERGENT_IDS)
def test_divergent_fixtures_every_verdict(e):
    """Damaged frames on which the candidates for "the reference" disagree: the oracle follows zstd 1.4.8 (as 1.4.8
    dispatches, no pre-1.0 formats); with the port's always-prefetching sequence loop, or with the single-symbol Huffman
    decoder forced, it reproduces the other recorded verdicts."""
    fr = read_divergent_frame(e)
    try:
        assert verdict_of(*oracle.decompress(fr, e["capacity"])) == e["build"]
        if e["class"] != "legacy_magic":
            assert e["build"] == e["zstd148"]
        if "as_written" in e:
            oracle.set_body(oracle.BODY_ALWAYS_PREFETCHING)
            assert verdict_of(*oracle.decompress(fr, e["capacity"])) == e["as_written"]
        if "never_prefetching" in e:      # frames that zstd 1.4.8 itself hands to its prefetching loop (window above 2^24)
            oracle.set_body(oracle.BODY_NEVER_PREFETCHING)
            assert verdict_of(*oracle.decompress(fr, e["capacity"])) == e["never_prefetching"]
            assert e["never_prefetching"] != e["build"]
        oracle.set_body(oracle.BODY_AS_ZSTD_148)
        if "x1_only" in e:
            oracle.set_huf(1)
            assert verdict_of(*oracle.decompress(fr, e["capacity"])) == e["x1_only"]
    finally:
        oracle.set_body(oracle.BODY_AS_ZSTD_148)
        oracle.set_huf(0)


from conftest import CONTINUE_ENTRIES, CONTINUE_IDS, CONTINUE_KEYS, continue_input  # noqa: E402
import continue_driver as cd  # noqa: E402


@pytest.mark.parametrize("x", CONTINUE_ENTRIES, ids=CONTINUE_IDS)
def test_oracle_bufferless_interface_matches_libzstd(x):
    """ZSTD_decompressBegin[_usingDict] / ZSTD_nextSrcSizeToDecompress / ZSTD_nextInputType / ZSTD_decompressContinue
    (zstd_decompressf.pas:995-1246, :1386-1421): every call's input type, size and result, the first error and the output
    are those libzstd 1.4.8 recorded in tests/golden/continue_traces.json."""
    data, e, d = continue_input(x)
    be = cd.OracleBackend(bool(e.get("format", 0)), bool(e.get("ignore_checksum", 0)))
    r = cd.drive(be, data, e["capacity"], d, raw_chunk=x["raw_chunk"])
    assert {k: r[k] for k in CONTINUE_KEYS} == {k: x[k] for k in CONTINUE_KEYS}
