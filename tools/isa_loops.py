#!/usr/bin/env python3
"""List the loops of one kernel in a hipcc -S listing with instruction counts per class.

    python tools/isa_loops.py file.s <substring of the mangled kernel name>
"""
import re, sys

def main():
    path, key = sys.argv[1], sys.argv[2]
    L = open(path).read().split("\n")
    start = next(i for i, l in enumerate(L) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
    L = L[start:end + 1]
    V, S, D, M = r"^\tv_", r"^\ts_", r"^\tds_", r"^\t(global|buffer|scratch|flat)_"
    def count(seg, pat): return sum(1 for x in seg if re.search(pat, x))
    print(f"kernel: {len(L)} lines, valu={count(L, V)} salu={count(L, S)} lds={count(L, D)} vmem={count(L, M)}")
    for i, l in enumerate(L):
        if "Loop Header" not in l: continue
        name = l.split(":")[0]
        be = [j for j in range(i, len(L)) if re.search(r"s_c?branch\w* " + re.escape(name) + r"\b", L[j])]
        e = be[-1] if be else i
        seg = L[i:e + 1]
        depth = l.split("Depth=")[1].strip() if "Depth=" in l else "?"
        print(f"{name:12s} lines {i:6d}-{e:6d} depth={depth} valu={count(seg, V):4d} pk_fma={count(seg, 'v_pk_fma'):3d} "
              f"salu={count(seg, S):4d} lds={count(seg, D):3d} vmem={count(seg, M):3d} sleep={count(seg, 's_sleep')} "
              f"dma={count(seg, 'global_load_lds')} readfirstlane={count(seg, 'v_readfirstlane')}")

if __name__ == "__main__":
    main()
