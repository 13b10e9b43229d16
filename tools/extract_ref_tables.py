#!/usr/bin/env python3
"""Extract the only known-answer data the reference tree holds for the decode path into tests/golden/ref_default_tables.json
(numbers only): the three predefined sequence decoding tables (decompress/zstd_decompress_block.pas:29-125, cells
{nextState, nbAdditionalBits, nbBits, baseValue}, the first entry of each being the header {fastMode, tableLog}) and the
code tables LL_base / OF_base / OF_bits / ML_base (decompress/zstd_ddict.pas:23-49) with LL_bits / ML_bits and the
default distributions (common/zstd_internal.pas:57-99).

    python tools/extract_ref_tables.py [/root/reference]

The tests compare the oracle's and the GPU's predefined tables with this file.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def numbers(text):
    return [int(t[1:], 16) if t.startswith("$") else int(t) for t in re.findall(r"\$[0-9A-Fa-f]+|-?\d+", text)]


def array_body(src, name):
    m = re.search(re.escape(name) + r"\s*:\s*array[^=]*=\s*\((.*?)\);", src, re.S)
    assert m, name
    return m.group(1)


def cells(src, name):
    body = array_body(src, name)
    out = []
    for m in re.finditer(r"nextState:\s*(\d+);\s*nbAdditionalBits:\s*(\d+);\s*nbBits:\s*(\d+);\s*baseValue:\s*(\d+)", body):
        out.append([int(x) for x in m.groups()])
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    blk = open(os.path.join(ref, "decompress", "zstd_decompress_block.pas"), encoding="latin-1").read()
    dd = open(os.path.join(ref, "decompress", "zstd_ddict.pas"), encoding="latin-1").read()
    zi = open(os.path.join(ref, "common", "zstd_internal.pas"), encoding="latin-1").read()
    out = {"source": "terrylao/zstd: decompress/zstd_decompress_block.pas:29-125, decompress/zstd_ddict.pas:23-49, common/zstd_internal.pas:57-99",
           "cell_fields": ["nextState", "nbAdditionalBits", "nbBits", "baseValue"]}
    for name, n in (("LL_defaultDTable", 65), ("OF_defaultDTable", 33), ("ML_defaultDTable", 65)):
        c = cells(blk, name)
        assert len(c) == n, (name, len(c))
        out[name] = {"tableLog": c[0][3], "cells": c[1:]}      # header cell overlays {fastMode, tableLog}: baseValue is the log
    for name, src_, n in (("LL_base", dd, 36), ("OF_base", dd, 32), ("OF_bits", dd, 32), ("ML_base", dd, 53),
                          ("LL_bits", zi, 36), ("ML_bits", zi, 53), ("LL_defaultNorm", zi, 36), ("ML_defaultNorm", zi, 53), ("OF_defaultNorm", zi, 29)):
        v = numbers(array_body(src_, name))
        assert len(v) == n, (name, len(v))
        out[name] = v
    path = os.path.join(ROOT, "tests", "golden", "ref_default_tables.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=None, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, {k: (len(v["cells"]) if isinstance(v, dict) else len(v)) for k, v in out.items() if k not in ("source", "cell_fields")})


if __name__ == "__main__":
    main()
