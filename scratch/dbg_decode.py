import sys, numpy as np, torch
sys.path.insert(0,'.')
import oracle
from rqae_amd import RQAE, synth
cfg=dict(synth.CONFIGS["C4"]); cfg["num_quantizers"]=4
w=synth.weights(1234,cfg["dim"],cfg["codebook_dim"],cfg["num_quantizers"],cfg["codebook_size"])
m=RQAE(**cfg).eval(); m.load_state_dict(synth.state_dict_from_weights(w)); m=m.to("cuda")
rng=np.random.Generator(np.random.PCG64(1))
n=64
idx=rng.integers(0,8192,size=(n,4))
ref=oracle.decode(idx,w)
for lv in ([0],[1],[0,1],[0,1,2,3]):
    with torch.inference_mode():
        d=m.decode(torch.from_numpy(idx).cuda().view(1,n,4),layers=lv).cpu().numpy()[0]
    r=oracle.decode(idx,w,layers=lv)
    bad=np.argwhere(d!=r)
    print("layers",lv,"mismatch",len(bad),"of",d.size, "maxabs",np.abs(d-r).max())
    if len(bad):
        print("  tokens",np.unique(bad[:,0])[:20],"elems mod 64:",np.unique(bad[:,1]%64)[:10],"elem//64 uniq",np.unique(bad[:,1]//64)[:20])
import os
os.environ["RQAE_HIP_VARIANT"]="GEN"
with torch.inference_mode():
    d=m.decode(torch.from_numpy(idx).cuda().view(1,n,4)).cpu().numpy()[0]
print("GEN variant mismatch", (d!=ref).sum())
