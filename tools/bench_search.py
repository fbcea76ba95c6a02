"""Time the similarity search (rqae_amd.search.IntensityEngine.find_examples) on the demo server's geometry.

    python tools/bench_search.py [--shards 1] [--levels 1023] [--reps 3] [--cpu-sample 4]

One shard = 1024 sequences x 127 positions; K = 625; the reference's default layer list.  Prints one JSON line: whole
find_examples time (all layer ranges, ranking and example gathers included), the accumulate kernel's time alone with the rate of
table-row gathers it sustains (one 256-byte row per dataset token and level), and the numpy oracle timed on `--cpu-sample`
sequences for scale.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=1)
    ap.add_argument("--levels", type=int, default=1023)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=4)
    a = ap.parse_args()
    from rqae_amd import _native
    from rqae_amd.search import DEFAULT_LAYERS, IntensityEngine

    N, S, L, K = 1024 * a.shards, 127, a.levels, 625
    layers = [v for v in DEFAULT_LAYERS if v < L] + [L]
    g = torch.Generator(device="cuda").manual_seed(1)
    codes = torch.randint(0, K, (N, S, L), generator=g, dtype=torch.int32, device="cuda")
    sims = ((torch.rand((L, K, K), generator=g, device="cuda") * 2 - 1) * (4.0 * 0.995 ** torch.arange(L, device="cuda"))[:, None, None]).half()
    eng = IntensityEngine(sims, codes, device="cuda")
    del codes
    for _ in eng.find_examples(idx=3, layers=layers):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        for _ in eng.find_examples(idx=3, layers=layers):
            pass
    torch.cuda.synchronize()
    whole_ms = (time.perf_counter() - t0) / a.reps * 1e3
    # the accumulate kernel alone over the full depth, one range per layer entry as find_examples issues them
    query = eng._query(3, None)[:, :L].contiguous()
    tables = []
    bounds = [0] + layers
    for l0, l1 in zip(bounds[:-1], bounds[1:]):
        tables.append(_native.search_query_table(eng.sims, query, l0, l1).clone())
    acc_buf = torch.empty((N, S, 128), dtype=torch.float16, device="cuda")
    _native.set_code_check("lazy")      # no synchronisation between the timed calls
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(a.reps):
        for i, (l0, l1) in enumerate(zip(bounds[:-1], bounds[1:])):
            _native.search_accumulate(eng.activations, tables[i], K, S, l0, l1, acc_buf, first=(i == 0))
    ev1.record()
    torch.cuda.synchronize()
    acc_ms = ev0.elapsed_time(ev1) / a.reps
    gathers = N * S * L                                   # (token, level) pairs: one 256-byte table row each
    out = {"workload": f"find_examples: {a.shards} shard(s) of 1024 x 127 codes, depth {L}, K 625, layers {layers}",
           "find_examples_ms": round(whole_ms, 3), "accumulate_ms": round(acc_ms, 3),
           "row_gathers_per_s": gathers / (acc_ms * 1e-3), "lds_gather_TBps": gathers * 256 / (acc_ms * 1e-3) / 1e12,
           "code_bytes_read_GBps": N * S * L * eng.activations.element_size() * 4 / (acc_ms * 1e-3) / 1e9,
           "sequences_per_s": N / (whole_ms * 1e-3)}
    if a.cpu_sample > 0:
        from oracle import search_oracle as so

        sub = eng.activations[: a.cpu_sample].cpu().view(torch.int16).to(torch.int32).numpy() & 0xFFFF
        q = query.cpu().numpy()
        s16 = eng.sims.cpu().numpy()
        t0 = time.perf_counter()
        for _ in so.find_examples(s16, sub, q, 1, 0, 1, layers):
            pass
        dt = time.perf_counter() - t0
        out["cpu_oracle"] = {"sequences": a.cpu_sample, "seconds": round(dt, 3), "sequences_per_s": a.cpu_sample / dt, "kind": "port (numpy, 1 thread)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
