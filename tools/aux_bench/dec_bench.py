import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rqae_amd import RQAE, synth
import bench
def t(fn,reps=10):
    with torch.inference_mode():
        fn(); fn(); torch.cuda.synchronize()
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps
cfg=bench.CFG
w=synth.weights(1234,cfg["dim"],cfg["codebook_dim"],cfg["num_quantizers"],cfg["codebook_size"])
m=RQAE(**cfg).eval(); m.load_state_dict(synth.state_dict_from_weights(w)); m=m.to("cuda")
n=262144
x=bench.synth_batch(n,torch.device("cuda"),1000).view(1,n,-1)
with torch.inference_mode(): q,idx=m(x)
td=t(lambda: m.decode(idx)); print(f"NBUF={os.environ.get('RQAE_HIP_DECODE_NBUF','2')} decode C2 {n} tok: {td:.3f} ms  {n/td/1e3:.1f} Mtok/s  out {n*9216/td/1e6:.0f} GB/s")
i16=idx.to(torch.uint16)
td=t(lambda: m.decode(i16)); print(f"   uint16 idx: {td:.3f} ms")
