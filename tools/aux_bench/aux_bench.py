import os, sys, torch, time
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
td=t(lambda: m.decode(idx)); print(f"decode C2 {n} tok: {td:.3f} ms  {n/td/1e3:.1f} Mtok/s  out {n*9216/td/1e6:.0f} GB/s")
i32=idx.to(torch.int32)
td=t(lambda: m.decode(i32)); print(f"decode C2 int32 idx: {td:.3f} ms")
p=torch.randn(1,n,5,device="cuda")
tq=t(lambda: m.quantize(p,0)); print(f"quantize C2 {n} rows: {tq:.3f} ms  {n/tq/1e3:.1f} Mrows/s")
xf=x.float()
tf=t(lambda: m(xf)); print(f"forward fp32 input: {tf:.3f} ms {n/tf/1e3:.1f} Mtok/s")
# C1 and pretrained-shape and C4 (generic) quick numbers
for name,nt,lv in [("C1",262144,None),("P",65536,16),("C4",16384,4),("C5",32768,8)]:
    c=dict(synth.CONFIGS[name]); 
    if lv: c["num_quantizers"]=lv
    ww=synth.weights(1,c["dim"],c["codebook_dim"],c["num_quantizers"],c["codebook_size"])
    mm=RQAE(**c).eval(); mm.load_state_dict(synth.state_dict_from_weights(ww)); mm=mm.to("cuda")
    xx=torch.randn(1,nt,c["dim"],device="cuda")
    tt=t(lambda: mm(xx),reps=3)
    from rqae_amd import _native
    print(f"{name} {c} N={nt}: forward {tt:.3f} ms  {nt/tt/1e3:.2f} Mtok/s   [{_native.kernel_info(c['dim'],c['codebook_dim'],mm.codebook.shape[1],c['num_quantizers'])}]")
