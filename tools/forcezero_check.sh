#!/bin/bash
# Timing-dependent errors: rebuild the library with every compiler-inserted wait forced to a full wait
# (-mllvm -amdgpu-waitcnt-forcezero; hand-written s_waitcnt inside inline asm is untouched) and run the stress and parity tests on
# it.  A correct kernel gives the same bits, slower; an error that depends on timing usually becomes far more frequent (the
# packed-FMA error of round 2: 100 x).  Every GPU test of the hand-written kernels runs, the similarity search's ranking
# included (round 2 left it out: rocPRIM's segmented sort did not survive the flag; round 3 replaced it by search_sort_kernel).
# Run on a GPU box from the repo root; the normal library is put back at the end.  Same flags per translation unit as the
# product build: both go through __graft_entry__.build_library().
set -u
cd "$(dirname "$0")/.."
CS=rqae_amd/csrc
T=$(mktemp -d)
cp $CS/librqae_hip.so $T/normal.so || exit 1
python3 -c "import __graft_entry__ as g; g.build_library(extra_flags=['-mllvm', '-amdgpu-waitcnt-forcezero'], objdir='$T')" 2>/dev/null \
    || { cp $T/normal.so $CS/librqae_hip.so; exit 1; }
python -m pytest tests/test_hip_parity.py tests/test_gram_arm.py tests/test_search.py -m gpu -q -x
rc=$?
cp $T/normal.so $CS/librqae_hip.so
rm -rf $T
exit $rc
