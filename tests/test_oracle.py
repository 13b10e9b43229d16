"""The CPU oracle against the golden vectors: every fixture under tests/golden/ carries the verdict
of libzstd 1.4.8 (regenerated size + sha256, or error code); the oracle must reproduce it."""
import hashlib

import pytest

from conftest import FRAMES, FRAME_IDS, read_frame, read_dict, regenerate
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
