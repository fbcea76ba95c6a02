import os, sys, time, torch, numpy as np
sys.path.insert(0,'.')
from rqae_amd import RQAE, synth
import bench
cfg=bench.CFG
w=synth.weights(1234,cfg["dim"],cfg["codebook_dim"],cfg["num_quantizers"],cfg["codebook_size"])
m=RQAE(**cfg).eval(); m.load_state_dict(synth.state_dict_from_weights(w)); m=m.to("cuda")
n=int(sys.argv[1]) if len(sys.argv)>1 else 262144
x=bench.synth_batch(n,torch.device("cuda"),1000).view(1,n,-1)
def t(fn,reps=10):
    with torch.inference_mode():
        fn(x); fn(x); torch.cuda.synchronize()
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn(x)
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps
for ab in [0,1,2,4,3,5,6,7]:
    os.environ["RQAE_HIP_ABLATE"]=str(ab)
    tr=t(m.forward); tc=t(m.encode)
    print(f"ablate={ab} (1=noDMA 2=noQuant 4=noPass): forward {tr:.3f} ms ({n/tr/1e3:.1f} Mtok/s)  codes-only {tc:.3f} ms ({n/tc/1e3:.1f} Mtok/s)", flush=True)
os.environ["RQAE_HIP_ABLATE"]="0"
