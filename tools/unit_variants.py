#!/usr/bin/env python3
"""Tools builds of librqae_hip.so that differ in ONE translation unit's -D flags: timing-only ablations of a kernel.

    python tools/unit_variants.py proj3.hip p3 base= nomfma=-DP3_ABL=1 nobarrier=-DP3_ABL=2 ...

builds tools/_build/<tag>/<name>/librqae_hip.so; run one with RQAE_HIP_LIBRARY=<that path>.  The other units' objects are
taken from rqae_amd/csrc (build the product first)."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

unit, tag = sys.argv[1], sys.argv[2]
for spec in sys.argv[3:]:
    name, _, flags = spec.partition("=")
    d = os.path.join(ROOT, "tools", "_build", tag, name)
    os.makedirs(d, exist_ok=True)
    for src in ge.TRANSLATION_UNITS:
        if src != unit:
            shutil.copy(os.path.join(ge.CSRC, src.replace(".hip", ".o")), d)
    ge.build_library(out_so=os.path.join(d, "librqae_hip.so"), extra_flags=[f for f in flags.split(",") if f], objdir=d, units=[unit])
    print("built", name, flags)
