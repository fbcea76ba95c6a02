#!/usr/bin/env python3
"""Build-time guard: no compiler-generated packed fp32 arithmetic in ANY kernel of the matrix-core arm's units or the search kernels.

    python tools/isa_guard.py rqae_amd/csrc/gram.o rqae_amd/csrc/search.o

Round 2 found that the SLP vectoriser's `v_pk_fma_f32` (op_sel forms) in the exact rescoring of the level chain gave wrong scores
in lanes 48-63 when several waves share a SIMD (cause not established).  The translation units are built
with -fno-slp-vectorize (the one shared build definition: __graft_entry__.GRAM_TU_FLAGS); this check makes that deterministic --
whatever pass, compiler version or build line re-introduces packed fp32 there fails the build instead of a stochastic test.
Exit status 1 and a listing when `v_pk_fma_f32` / `v_pk_mul_f32` / `v_pk_add_f32` appears in a guarded kernel.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
GUARDED = re.compile(r".")      # every kernel of the objects named on the command line (the matrix-core arm's units and the search)
PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")


def device_disassembly(obj):
    """Disassembly of the gfx950 code object bundled in a hipcc host object."""
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [os.path.join(td, f) for f in os.listdir(td) if "amdgcn" in f]
        if not cos:
            raise RuntimeError(f"{obj}: no amdgcn bundle found")
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", cos[0]], check=True, capture_output=True, text=True).stdout


def check(obj):
    bad = {}
    sym = None
    for line in device_disassembly(obj).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            sym = m.group(1)
            continue
        if sym and GUARDED.search(sym) and PACKED.search(line):
            bad.setdefault(sym, []).append(line.strip())
    return bad


def main(objs):
    failed = False
    for obj in objs:
        bad = check(obj)
        for sym, lines in bad.items():
            failed = True
            print(f"isa_guard: {obj}: {len(lines)} packed fp32 instruction(s) in {sym}, e.g. {lines[0]}", file=sys.stderr)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
