#!/usr/bin/env python3
"""Modal-free replay of the reference's activation/codes producer on one MI355X.

What scripts/1_create_activations.py:143-186,225-261 does per 1024-sequence shard, with the pieces
that need the network replaced by stand-ins:

  * the Gemma-2 decoder layer  -> a module that returns ``(hidden_states_fp16,)`` (rqae/llm.py:19-22 hooks
    exactly such a layer with ``register_forward_hook``);
  * ``llm.norm`` / ``llm.denorm`` -> the Gemma2RMSNorm math of rqae/llm.py:65-73 with a random weight;
  * the RQAE weights           -> seeded synthetic ones (``from_pretrained`` needs the HF hub).

Everything on the hot path is the product: ``rqae_amd.RQAE.hook`` -> ``forward`` -> the gfx950 kernels.
Output, as the reference writes it: fp16 original activations and int32 codes, one ``.pt`` per shard.

    python examples/create_activations.py --sequences 1024 --batch-size 4 --out /tmp/acts
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rqae_amd import RQAE, shards, synth  # noqa: E402


class FakeDecoderLayer(torch.nn.Module):
    def forward(self, hidden_states):
        return (hidden_states,)


class FakeGemma2:
    """norm / denorm as rqae/llm.py:65-73 computes them."""

    def __init__(self, weight):
        self.weight = weight

    def norm(self, hs):
        x = hs.float()
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
        return x * (1.0 + self.weight.float())

    def denorm(self, hidden_states, original_hidden_states):
        hidden_states = hidden_states / (1.0 + self.weight.float())
        return hidden_states.float() / torch.rsqrt(
            original_hidden_states.float().pow(2).mean(-1, keepdim=True) + 1e-6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=256)
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--batch-size", type=int, default=4, help="the reference uses 4 (T4 memory)")
    ap.add_argument("--num-quantizers", type=int, default=64, help="pretrained checkpoint: 1024")
    ap.add_argument("--out", default=None)
    ap.add_argument("--codes-dtype", choices=["int32", "uint16"], default="int32",
                    help="on-disk dtype of the code shard: int32 as the reference writes, uint16 = half the bytes (K <= 65536)")
    args = ap.parse_args()

    dev = torch.device("cuda")
    cfg = dict(dim=2304, codebook_dim=4, codebook_size=5, num_quantizers=args.num_quantizers)
    w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    rqae = RQAE(**cfg, name=f"rqae-synth-round_fsq-cbd4-cbs5-nq{args.num_quantizers}").eval()
    rqae.load_state_dict(synth.state_dict_from_weights(w), strict=True)
    rqae = rqae.to(dev)
    g = torch.Generator(device=dev); g.manual_seed(0)
    llm = FakeGemma2(0.1 * torch.randn(cfg["dim"], device=dev, generator=g))
    layer = FakeDecoderLayer()

    originals, codes = [], []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in range(0, args.sequences, args.batch_size):
        n = min(args.batch_size, args.sequences - b)
        hs = (3.0 * torch.randn(n, args.seq_len, cfg["dim"], device=dev, generator=g)).to(torch.float16)
        vals = {}
        handle = layer.register_forward_hook(rqae.hook(llm=llm, store=lambda k, v: vals.update({k: v})))
        with torch.inference_mode():
            layer(hs)                              # hook patches hs in place (rqae/model.py:289)
        handle.remove()
        originals.append(vals["original"].detach().cpu().to(torch.float16))     # scripts/1:181-183
        codes.append(vals["indices"].detach().cpu().to(torch.int32))            # scripts/1:184-186
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    originals, codes = torch.cat(originals), torch.cat(codes)
    toks = args.sequences * args.seq_len
    print(f"{toks} tokens, depth {args.num_quantizers}: {dt:.2f} s, {toks / dt / 1e3:.1f} k tok/s end to end "
          f"(hook + clones + D2H per batch of {args.batch_size}); codes {tuple(codes.shape)} {codes.dtype}")
    if args.out:
        os.makedirs(os.path.join(args.out, rqae.name), exist_ok=True)
        torch.save(originals, os.path.join(args.out, "000000_original.pt"))
        shards.save_codes(os.path.join(args.out, rqae.name, "000000.pt"), codes, dtype=args.codes_dtype,
                          codebook_rows=rqae.codebook.shape[1])                  # scripts/1:258-261 (int32), or half the bytes


if __name__ == "__main__":
    main()
