#!/bin/bash
# proj3 timing-only ablations (tools/unit_variants.py): which part of a stage the projection's time is
R=gpurun_out/r04k
mkdir -p $R
for v in ${VARIANTS:-base nomfma nobarrier nowread nowdma nomfma_nowread pf1 pf5}; do
  for spec in "C4 65536" "C5 262144"; do
    set -- $spec
    echo "== $v $1" >> $R/abl.txt
    RQAE_HIP_LIBRARY=tools/_build/p3/$v/librqae_hip.so timeout -k 10 120 python tools/aux_bench/config_bench.py $1 $2 --arms gram --reps 10 2>> $R/abl.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['gram']['kernel_ms'])" >> $R/abl.txt || { tail -5 $R/abl.err; exit 1; }
  done
done
cat $R/abl.txt
