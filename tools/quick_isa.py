#!/usr/bin/env python3
"""Compile ONE kernel instantiation of gram.hip to ISA in seconds (the whole translation unit takes two minutes) and print its
resource usage and loop / instruction-class counts.

    python tools/quick_isa.py 'gram_levels2_kernel<5, true, 16>' 'Lev2Params' [-DMACRO ...]

Writes tools/_build/quick.s.  The kernel is instantiated explicitly in a wrapper translation unit that includes gram.hip with
RQAE_GRAM_NO_ABI (kernels and helpers only).  Same flags as the product build (__graft_entry__.HIPCC_FLAGS + GRAM_TU_FLAGS).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    kern, params = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    out = os.path.join(ROOT, "tools", "_build")
    os.makedirs(out, exist_ok=True)
    src = os.path.join(out, "quick.hip")
    with open(src, "w") as fh:
        fh.write('#define RQAE_GRAM_NO_ABI\n#include "../../rqae_amd/csrc/gram.hip"\n'
                 f"namespace {{ template __global__ void {kern}({params}); }}\n")
    asm = os.path.join(out, "quick.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *ge.HIPCC_FLAGS, *ge.GRAM_TU_FLAGS, *extra, "-S", "--cuda-device-only", src, "-o", asm],
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(asm).read()
    for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize", "SGPRBlocks", "NumSgprs"):
        m = re.findall(rf"; {key}: (\d+)", txt)
        if m:
            print(f"{key}: {m[-1]}", end="  ")
    print()
    name = kern.split("<")[0]
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_loops.py"), asm, name])
    body = txt[txt.index(name):]
    body = body[:body.index("s_endpgm")]
    def cnt(pat):
        return len(re.findall(pat, body, flags=re.M))

    pats = {"scratch_load": r"^\s+scratch_load", "scratch_store": r"^\s+scratch_store", "global_store": r"^\s+global_store",
            "vmcnt waits": r"s_waitcnt.*vmcnt", "v_readlane": r"^\s+v_readlane", "v_mov": r"^\s+v_mov_b32"}
    print("  ".join(f"{k} {cnt(v)}" for k, v in pats.items()))


if __name__ == "__main__":
    main()
