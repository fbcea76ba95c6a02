cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0,'.')
sys.argv=['x']
exec(open('scratch/time_variants.py').read().split("for ab in")[0])
for ab in [0,1,2,4]:
    os.environ["RQAE_HIP_ABLATE"]=str(ab)
    tr=t(m.forward); tc=t(m.encode)
    print(f"ablate={ab}: forward {tr:.3f} ms ({n/tr/1e3:.1f} Mtok/s)  codes-only {tc:.3f} ms ({n/tc/1e3:.1f} Mtok/s)", flush=True)
PY
timeout -k 10 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
