"""RQAEFeature.intensity: GB/s of the code matrix at the pretrained shape (K = 625, 1024 levels)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rqae_amd import _native as nat
L, K = 1024, 625
rng = np.random.Generator(np.random.PCG64(5))
unit = rng.standard_normal((K, 4)).astype(np.float32); unit /= np.linalg.norm(unit, axis=1, keepdims=True)
sims = torch.from_numpy((unit @ unit.T).astype(np.float16)).cuda()
center = torch.from_numpy(rng.integers(0, K, size=L).astype(np.int32))
lw = torch.from_numpy((0.5 + rng.random(L)).astype(np.float16))
out = {}
for n in (131072, 524288):
    tok32 = torch.randint(0, K, (n, L), dtype=torch.int32, device="cuda")
    for name, tok in (("int32", tok32), ("uint16", tok32.to(torch.uint16))):
        for layers in ([1023], [63, 255, 511, 1023], [63]):
            M = max(layers) + 1
            table, w, wcum = nat.intensity_prepare(sims, center, lw, M)
            ld = torch.tensor(layers, dtype=torch.int32, device="cuda")
            for _ in range(2): nat.intensity(table, w, wcum, M, K, tok, ld)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): nat.intensity(table, w, wcum, M, K, tok, ld)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            # algorithmic bytes: the M columns read (whole 128-byte lines are fetched: count the row segment), + out
            es = tok.element_size()
            byt = n * (M * es + len(layers) * 2)
            out[f"n={n} {name} layers={layers}"] = {"ms": round(ms, 3), "GB/s": round(byt / ms / 1e6, 1), "Gcodes/s": round(n * M / ms / 1e6, 2)}
    del tok32
print(json.dumps(out, indent=1))
