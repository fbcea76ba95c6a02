#!/bin/bash
# the sign chain (gram_sign3.hpp): parity of the matrix-core arm, then C4's per-kernel times
R=gpurun_out/r04l
mkdir -p $R
( timeout -k 10 900 python -m pytest tests/test_gram_arm.py tests/test_hip_parity.py -x -q -m gpu > $R/pytest.txt 2>&1; rc=$?; tail -15 $R/pytest.txt; [ $rc -eq 0 ] ) || exit 1
for spec in "C4 65536 bf16" "C4 65536 f32" "C4 262144 bf16"; do
  set -- $spec
  python tools/aux_bench/config_bench.py $1 $2 --dtype $3 --arms gram --reps 10 >> $R/configs.txt 2>> $R/configs.err || { tail -5 $R/configs.err; exit 1; }
done
cat $R/configs.txt
