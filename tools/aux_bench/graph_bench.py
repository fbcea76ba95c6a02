"""Small-batch forward: eager call vs replay of a captured HIP graph (launch-bound regime)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rqae_amd import RQAE, synth
import bench
cfg = bench.CFG
w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
m = RQAE(**cfg).eval(); m.load_state_dict(synth.state_dict_from_weights(w)); m = m.cuda()
out = {}
for B, S in [(1, 128), (8, 128), (64, 128)]:
    x = bench.synth_batch(B * S, torch.device("cuda"), 3).view(B, S, -1)
    with torch.inference_mode():
        for _ in range(5): m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): m(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 200 * 1e6
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            q, idx = m(x)
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 200 * 1e6
    out[f"{B}x{S} tokens"] = {"eager_us_per_forward": round(eager, 1), "graph_replay_us": round(graph, 1),
                             "Mtok/s eager": round(B * S / eager, 2), "Mtok/s graph": round(B * S / graph, 2)}
print(json.dumps(out, indent=1))
