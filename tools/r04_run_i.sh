#!/bin/bash
R=gpurun_out/${OUT:-r04i}
mkdir -p $R
( timeout -k 10 600 python -m pytest tests/test_gram_arm.py -x -q -m gpu -k "deep or released or fixtures" > $R/pytest.txt 2>&1; rc=$?; tail -3 $R/pytest.txt; [ $rc -eq 0 ] ) || exit 1
python tools/aux_bench/config_bench.py P 262144 --levels 1024 --reps 3 --arms gram > $R/p1024.txt 2> $R/p1024.err || { tail -5 $R/p1024.err; exit 1; }
python tools/aux_bench/config_bench.py P 524288 --levels 1024 --reps 2 --arms gram >> $R/p1024.txt 2>> $R/p1024.err || { tail -5 $R/p1024.err; exit 1; }
cat $R/p1024.txt
