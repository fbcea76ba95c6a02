#!/bin/bash
# round 4: recon3 ablations (timing only) at C2 and C4
R=gpurun_out/r04b
mkdir -p $R
( timeout -k 10 420 python -m pytest tests/test_gram_arm.py -x -q -m gpu > $R/pytest_gram.txt 2>&1; rc=$?; tail -4 $R/pytest_gram.txt; [ $rc -le 1 ] ) || exit 1
run() {  # label, env..., -- args
  label=$1; shift
  echo "== $label" >> $R/abl.txt
  env "$@" python tools/aux_bench/config_bench.py $SPEC --arms gram --reps 20 >> $R/abl.txt 2>> $R/abl.err || exit 1
}
for SPEC in "C2 262144" "C4 65536" "P 262144"; do
  echo "#### $SPEC" >> $R/abl.txt
  run product X=1
  run base_nopref RQAE_HIP_LIBRARY=tools/_build/r3/base/librqae_hip.so R3_FORCE=nopref
  run base_nt4_nopref RQAE_HIP_LIBRARY=tools/_build/r3/base/librqae_hip.so R3_FORCE=nt4,nopref
  run nomfma RQAE_HIP_LIBRARY=tools/_build/r3/nomfma/librqae_hip.so
  run nostore RQAE_HIP_LIBRARY=tools/_build/r3/nostore/librqae_hip.so
  run nomfma_nostore RQAE_HIP_LIBRARY=tools/_build/r3/nomfma_nostore/librqae_hip.so
  run nodma RQAE_HIP_LIBRARY=tools/_build/r3/nodma/librqae_hip.so
  echo "== old recon (options recon=2)" >> $R/abl.txt
  python tools/aux_bench/config_bench.py $SPEC --arms gram --reps 20 --recon 2 >> $R/abl.txt 2>> $R/abl.err || exit 1
done
grep -o '== .*\|#### .*\|"kernel_ms": {[^}]*}' $R/abl.txt
