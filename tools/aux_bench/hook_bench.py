"""RQAE.hook(llm=gemma) per call: the matrix-core arm with the norm folded into its projection and the way back folded into its
reconstruction's stores (rqae_hip_encode_gram_gemma_patch: the layer output is patched in place), the
exact arm's fused hook (rqae_hip_encode_gemma), and the generic hook (torch ops around forward) -- ms per hook call, fp16 hidden
states, no store.  Run ON the GPU box:  python tools/aux_bench/hook_bench.py [--quick]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from rqae_amd import RQAE, synth  # noqa: E402
from test_hip_parity import FakeGemma2LLM  # noqa: E402

cfg = synth.CONFIGS["C2"]
if "--released" in sys.argv:      # the released model's shape at its full depth (rqae/model.py:18-24), the production call 4 x 128 (scripts/1:152)
    cfg = dict(dim=2304, codebook_dim=4, codebook_size=5, num_quantizers=1024)
w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
models = {}
for arm in ("gram", "exact"):
    m = RQAE(**cfg, arm=arm).eval()
    m.load_state_dict(synth.state_dict_from_weights(w))
    models[arm] = m.cuda()
nw = torch.randn(cfg["dim"], device="cuda") * 0.1
llm = FakeGemma2LLM(nw)
out = {}
shapes = [(64, 128), (2048, 128)] if "--quick" in sys.argv else [(4, 128), (64, 128), (256, 128), (2048, 128)]
if "--released" in sys.argv:
    shapes = [(4, 128), (32, 128), (256, 128)]
for B, S in shapes:
    hs0 = (torch.randn(B, S, cfg["dim"], device="cuda") * 3).half()
    for name, arm, fused in (("gram arm, norm and way back folded", "gram", True), ("exact arm, fused", "exact", True), ("generic (torch ops + exact forward)", "exact", False)):
        fn = models[arm].hook(llm=llm, fused=fused)
        hs = hs0.clone()
        reps = (30 if B * S >= 100000 else 50) if cfg["num_quantizers"] <= 64 else 5
        with torch.inference_mode():
            for _ in range(5):
                fn(None, None, (hs,))
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn(None, None, (hs,))
            b.record()
            torch.cuda.synchronize()
        out[f"{B}x{S} tokens: {name}"] = round(a.elapsed_time(b) / reps, 4)
print(json.dumps(out, indent=1))
