#!/bin/bash
mkdir -p gpurun_out/r04h
RQAE_HIP_LIBRARY=tools/_build/slpnop/librqae_hip.so timeout -k 10 240 python tools/slp_experiment.py > gpurun_out/r04h/slp_on_nop_padded.json 2> gpurun_out/r04h/err2.txt || { tail -5 gpurun_out/r04h/err2.txt; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/r04h/slp_on_nop_padded.json"))
for k,v in j.items(): print(k, json.dumps(v)[:300])
PY
