#!/usr/bin/env python3
"""Modal-free replay of the reference's feature builder on one MI355X.

What scripts/3_make_rqae_features.py:98-168,174-215 does per feature, with the pieces that need the network or
the Modal volume replaced by stand-ins:

  * the dataset's code shards (``activations/<model>/NNNNNN.pt``, int32 ``(1024, 128, NQ)``)  -> random codes of
    that shape, kept on the device;
  * the RQAE weights -> seeded synthetic ones (``from_pretrained`` needs the HF hub).

The product does the rest: ``RQAEFeature.from_quantizer`` reads ``RQAE.codebook_sims`` / ``layer_norms`` (the
tables kernels), ``RQAEFeature.intensity`` runs the intensity kernel over the dataset in
batches of 1024 sequences as the reference does (scripts/3:104-110), and the top / middle / bottom-k selection per
layer is the reference's argsort (scripts/3:115-127), here on the device.

    python examples/make_features.py --sequences 4096 --features 8
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rqae_amd import RQAE, RQAEFeature, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=4096, help="the reference's dataset: ~36k sequences")
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--num-quantizers", type=int, default=1024)
    ap.add_argument("--features", type=int, default=8, help="the reference builds 1024")
    ap.add_argument("--top-k", type=int, default=100)
    ap.add_argument("--codes", default="", help="a code shard written by create_activations.py (int32 or uint16); default: random codes")
    args = ap.parse_args()

    dev = torch.device("cuda")
    nq = args.num_quantizers
    cfg = dict(dim=2304, codebook_dim=4, codebook_size=5, num_quantizers=nq)
    w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], nq, cfg["codebook_size"])
    rqae = RQAE(**cfg, name=f"rqae-synth-round_fsq-cbd4-cbs5-nq{nq}").eval()
    rqae.load_state_dict(synth.state_dict_from_weights(w), strict=True)
    rqae = rqae.to(dev)
    K = rqae.codebook.shape[1]

    g = torch.Generator(device=dev); g.manual_seed(0)
    if args.codes:
        from rqae_amd import shards
        indices = shards.load_codes(args.codes).to(dev)          # consumed in its on-disk dtype (uint16 shards are not widened)
        args.sequences, args.seq_len = indices.shape[0], indices.shape[1]
        assert indices.shape[2] == nq, f"shard depth {indices.shape[2]} != --num-quantizers {nq}"
    else:
        indices = torch.randint(0, K, (args.sequences, args.seq_len, nq), device=dev, generator=g, dtype=torch.int32)
    layers = [l for l in [2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128, 256, 512, 1023] if l < nq]      # scripts/3:196
    picks = torch.randint(0, args.sequences * args.seq_len, (args.features,), generator=torch.Generator().manual_seed(1))

    torch.cuda.synchronize(); t0 = time.perf_counter()
    _ = rqae.codebook_sims, rqae.layer_norms                     # the tables from_quantizer reads (first use)
    torch.cuda.synchronize(); t_tables = time.perf_counter() - t0

    t_int = t_sel = 0.0
    for f in range(args.features):
        k = int(picks[f])
        center = indices[k // args.seq_len, k % args.seq_len].cpu().numpy()                      # scripts/3:186,199
        feature = RQAEFeature.from_quantizer(rqae, center=center, layers=layers)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.inference_mode():
            parts = [feature.intensity(indices[b:b + 1024], layers=layers)                       # scripts/3:104-110
                     for b in range(0, args.sequences, 1024)]
            all_int = torch.cat(parts, dim=0).flatten(0, 1)                                        # tokens x layers
        torch.cuda.synchronize(); t1 = time.perf_counter()
        chosen = {}
        for l_idx, l in enumerate(layers):                                                        # scripts/3:115-127
            order = torch.argsort(all_int[:, l_idx], descending=True)
            mid = len(order) // 2
            chosen[l] = torch.cat([order[:args.top_k], order[mid - args.top_k // 2: mid + args.top_k // 2],
                                   order[-args.top_k:]])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        t_int += t1 - t0
        t_sel += t2 - t1
        if f == 0:
            top = all_int[chosen[layers[-1]][:3], -1].float().tolist()
            print(f"feature 0: centre token {k}, its own intensity at the last layer "
                  f"{float(all_int[k, -1]):.3f} (a token is most similar to itself), top-3 {top}")
    toks = args.sequences * args.seq_len
    print(f"{toks} tokens x {nq} levels, {args.features} features, {len(layers)} layers each: "
          f"tables {t_tables * 1e3:.1f} ms once; per feature intensity {t_int / args.features * 1e3:.2f} ms "
          f"({toks * nq * args.features / t_int / 1e12:.2f} T codes/s), selection {t_sel / args.features * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
