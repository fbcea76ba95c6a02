"""The cached tables on one MI355X: rqae_hip_subfeature_sims / _subfeatures / _codebook_sims at the
pretrained shape (L 1024, K 625, c 4, d 2304), the C2 shape and the large-codebook shape, next to the reference's
torch ops on the same GPU where they fit.  Output kept in profiles/r01_tables_bench.txt."""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rqae_amd import _native
dev = 'cuda:0'
def bench(name, fn, nbytes, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): fn()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    print(f"{name}: {ms:.3f} ms  {nbytes/ms/1e9:.2f} TB/s written", flush=True)
for (L, K, c, d, tag) in [(1024, 625, 4, 2304, 'pretrained'), (8, 1024, 5, 2304, 'C2'), (16, 8192, 13, 3584, 'C4')]:
    cb = torch.randn(L, K, c, device=dev); cb /= cb.norm(dim=-1, keepdim=True)
    wo = (torch.rand(L, d, c, device=dev) - 0.5); bo = (torch.rand(L, d, device=dev) - 0.5)
    print(tag, L, K, c, d)
    bench(' subfeature_sims+layer_norms', lambda: _native.subfeature_sims(cb, wo, bo), L*K*K*2)
    bench(' layer_norms only', lambda: _native.subfeature_sims(cb, wo, bo, want_sims=False), L*K*8)
    if L*K*d*4 < 8e9:
        bench(' subfeatures', lambda: _native.subfeatures(cb, wo, bo), L*K*d*4, reps=3)
    bench(' codebook_sims', lambda: _native.codebook_sims(cb[0]), K*K*2)
    if tag in ('C2', 'pretrained'):
        def torch_way():
            sub = torch.einsum('lkc,ldc->lkd', cb, wo) + bo[:, None, :]
            u = torch.nn.functional.normalize(sub, dim=-1)
            return (u @ u.transpose(-1, -2)).half()
        bench(' torch subfeature_sims (reference ops on GPU)', torch_way, L*K*K*2)
