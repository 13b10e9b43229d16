import hashlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


MANIFEST = load_manifest()
FRAMES = MANIFEST["frames"]
FRAME_IDS = [e["name"] for e in FRAMES]


def read_frame(e) -> bytes:
    with open(os.path.join(GOLD, e["file"]), "rb") as f:
        return f.read()


_dict_cache = {}


def read_dict(name):
    if name is None:
        return None
    if name not in _dict_cache:
        meta = MANIFEST["dicts"][name]
        with open(os.path.join(GOLD, meta["file"]), "rb") as f:
            d = f.read()
        assert hashlib.sha256(d).hexdigest() == meta["sha256"]
        _dict_cache[name] = d
    return _dict_cache[name]


def regenerate(source) -> bytes:
    """Plain bytes of a fixture from its recorded generator and seed."""
    from zstd_amd import corpus
    import random
    g = source["gen"]
    if g == "text":
        return corpus.text_like(source["seed"], source["n"])
    if g == "bin":
        return corpus.binary_records(source["seed"], source["n"])
    if g == "periodic":
        rnd = random.Random(source["seed"])
        unit = bytes(rnd.randrange(256) for _ in range(source["period"]))
        n = source["n"]
        return (unit * (n // source["period"] + 1))[:n]
    raise KeyError(g)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the CPU-side helpers (oracle, corpus generator) if missing; the GPU library is prebuilt in-tree."""
    from zstd_amd import build
    build.build_corpus()
    build.build_oracle()
    yield


# fixtures whose verdict depends on a decoder parameter (tools/gen_param_fixtures.py)
def load_param_manifest():
    with open(os.path.join(GOLD, "params", "manifest.json")) as f:
        return json.load(f)


PARAM_FRAMES = load_param_manifest()["frames"]
PARAM_IDS = [e["name"] for e in PARAM_FRAMES]


def read_param_frame(e) -> bytes:
    with open(os.path.join(GOLD, "params", e["file"]), "rb") as f:
        return f.read()


# damaged frames on which 1.4.8, the port as written and a single-symbol-only decoder disagree (tools/gen_divergent_fixtures.py)
def load_divergent_manifest():
    with open(os.path.join(GOLD, "divergent", "manifest.json")) as f:
        return json.load(f)


DIVERGENT_FRAMES = load_divergent_manifest()["frames"]
DIVERGENT_IDS = [e["name"] for e in DIVERGENT_FRAMES]


def read_divergent_frame(e) -> bytes:
    with open(os.path.join(GOLD, "divergent", e["file"]), "rb") as f:
        return f.read()


def verdict_of(out, err):
    """(error code, size, sha256) in the shape of the manifests' verdict entries."""
    if err:
        return {"error": err}
    return {"size": len(out), "sha256": hashlib.sha256(out).hexdigest()}


# what libzstd 1.4.8 does through its bufferless interface on every frame above (tools/gen_continue_fixtures.py)
def load_continue_entries():
    with open(os.path.join(GOLD, "continue_traces.json")) as f:
        return json.load(f)["entries"]


CONTINUE_ENTRIES = load_continue_entries()
CONTINUE_IDS = ["%s%s-%s" % (x["set"] + "/" if x["set"] else "", x["name"], x["variant"]) for x in CONTINUE_ENTRIES]
CONTINUE_KEYS = ("steps", "trace_sha256", "err", "consumed", "out_size", "out_sha256", "frames")
_BY_SET = {"": {e["name"]: e for e in FRAMES}, "params": {e["name"]: e for e in PARAM_FRAMES},
           "divergent": {e["name"]: e for e in DIVERGENT_FRAMES}}


def continue_input(x):
    """(frame bytes, manifest entry, dictionary or None) of an entry of continue_traces.json."""
    e = _BY_SET[x["set"]][x["name"]]
    with open(os.path.join(GOLD, x["set"], e["file"]), "rb") as f:
        data = f.read()
    return data, e, read_dict(e.get("dict"))
