#!/bin/bash
# gram_recon3_kernel variants at C4 and C5 (tools/unit_variants.py recon3.hip r3v ...)
R=gpurun_out/r04n
mkdir -p $R
for v in ${VARIANTS:-base nomfma nostore nodma nomfma_nostore}; do
  for spec in ${SPECS:-C4:65536 C5:262144}; do
    echo "== $v $spec" >> $R/abl.txt
    RQAE_HIP_LIBRARY=tools/_build/r3v/$v/librqae_hip.so timeout -k 10 120 python tools/aux_bench/config_bench.py ${spec%%:*} ${spec##*:} --arms gram --reps 10 2>> $R/abl.err | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['gram']['kernel_ms'])" >> $R/abl.txt || { tail -5 $R/abl.err; exit 1; }
  done
done
cat $R/abl.txt
