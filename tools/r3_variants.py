#!/usr/bin/env python3
"""Tools builds of librqae_hip.so that differ only in recon3.hip (the fp16 two-piece reconstruction): timing-only ablations.

    python tools/r3_variants.py            # builds tools/_build/r3/{base,nomfma,nostore,nomfma_nostore,nodma}/librqae_hip.so

Run a variant with RQAE_HIP_LIBRARY=tools/_build/r3/<name>/librqae_hip.so (and R3_FORCE=nt4 / nopref in the `base` build).
The other translation units' objects are taken from rqae_amd/csrc (build the product first)."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

VARIANTS = {"base": [], "nomfma": ["-DR3_ABL=1"], "nostore": ["-DR3_ABL=2"], "nomfma_nostore": ["-DR3_ABL=3"], "nodma": ["-DR3_ABL=4"]}
for name, flags in VARIANTS.items():
    d = os.path.join(ROOT, "tools", "_build", "r3", name)
    os.makedirs(d, exist_ok=True)
    for src in ge.TRANSLATION_UNITS:
        if src != "recon3.hip":
            shutil.copy(os.path.join(ge.CSRC, src.replace(".hip", ".o")), d)
    ge.build_library(out_so=os.path.join(d, "librqae_hip.so"), extra_flags=["-DR3_TOOLS", *flags], objdir=d, units=["recon3.hip"])
    print("built", name)
