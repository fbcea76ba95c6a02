#!/bin/bash
# round 4, first GPU pass: parity of the new reconstruction kernel, the bench line with its configs block, before / after per config
R=gpurun_out/r04a
mkdir -p $R
( timeout -k 10 420 python -m pytest tests/test_gram_arm.py -x -q -m gpu > $R/pytest_gram.txt 2>&1; rc=$?; tail -4 $R/pytest_gram.txt; [ $rc -le 1 ] ) &&
python bench.py --steps 20 --warmup 5 > $R/bench.json 2> $R/bench.err && tail -c 300 $R/bench.json && echo &&
for spec in "C4 65536 --dtype bf16" "C4 65536 --dtype f32" "C5 262144" "P 262144" "C2 262144"; do
  python tools/aux_bench/config_bench.py $spec --arms gram --recon 2 >> $R/configs_old_recon.jsonl 2>> $R/cfg.err &&
  python tools/aux_bench/config_bench.py $spec --arms gram >> $R/configs_new_recon.jsonl 2>> $R/cfg.err || exit 1
done
cat $R/configs_old_recon.jsonl $R/configs_new_recon.jsonl
