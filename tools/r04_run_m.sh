#!/bin/bash
# batched plane loads in the generic piece build of gram_recon3_kernel: parity, then the configs' per-kernel times
R=gpurun_out/r04m
mkdir -p $R
( timeout -k 10 900 python -m pytest tests/test_gram_arm.py tests/test_hip_parity.py -x -q -m gpu > $R/pytest.txt 2>&1; rc=$?; tail -4 $R/pytest.txt; [ $rc -eq 0 ] ) || exit 1
for spec in "C4 65536 bf16" "C4 65536 f32" "C5 262144 bf16" "P 262144 bf16 --levels 16" "C2 262144 bf16"; do
  set -- $spec
  python tools/aux_bench/config_bench.py $1 $2 --dtype $3 $4 $5 --arms gram --reps 10 >> $R/configs.txt 2>> $R/configs.err || { tail -5 $R/configs.err; exit 1; }
done
cat $R/configs.txt
