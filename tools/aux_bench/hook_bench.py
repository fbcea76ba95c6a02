"""Fused (HIP) vs generic (torch ops) hook around forward(): ms per hook call."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, numpy as np
from rqae_amd import RQAE, synth
from test_hip_parity import FakeGemma2LLM
cfg = synth.CONFIGS["C2"]
w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
m = RQAE(dim=cfg["dim"], codebook_dim=cfg["codebook_dim"], codebook_size=4, num_quantizers=cfg["num_quantizers"]).eval()
m.load_state_dict(synth.state_dict_from_weights(w)); m = m.cuda()
nw = torch.randn(cfg["dim"], device="cuda") * 0.1
llm = FakeGemma2LLM(nw)
out = {}
for B, S in [(64, 128), (256, 128), (2048, 128)]:
    hs0 = (torch.randn(B, S, cfg["dim"], device="cuda") * 3).half()
    for fused in (True, False):
        fn = m.hook(llm=llm, fused=fused)
        hs = hs0.clone()
        with torch.inference_mode():
            for _ in range(3): fn(None, None, (hs,))
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): fn(None, None, (hs,))
            b.record(); torch.cuda.synchronize()
        out[f"{B}x{S} fused={fused}"] = a.elapsed_time(b) / 10
        # forward alone
    x = hs0.float()
    with torch.inference_mode():
        for _ in range(3): m(x)
        torch.cuda.synchronize(); a.record()
        for _ in range(10): m(x)
        b.record(); torch.cuda.synchronize()
    out[f"{B}x{S} forward(fp32 in) alone"] = a.elapsed_time(b) / 10
print(json.dumps(out, indent=1))
