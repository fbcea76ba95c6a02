"""Capture golden vectors from the reference implementation.

Runs ONLY in the build container (it imports /root/reference/rqae/model.py, which
does not exist on the GPU box).  Output: small .npz fixtures next to this file.
Inputs are regenerated from seeds by rqae_amd.synth (sha256 of the regenerated
bytes is stored as a guard); outputs are what the reference computed on CPU,
fp32, under torch.inference_mode().

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from rqae.model import RQAE as RefRQAE  # noqa: E402  (the reference itself)
from rqae_amd import synth  # noqa: E402

SEED_W = 1234


def ref_model(cfg, w, method="round_fsq"):
    m = RefRQAE(**cfg, quantization_method=method).eval()
    m.load_state_dict(synth.state_dict_from_weights(w), strict=True)
    return m


def sample_rows(n):
    """Rows kept from [n, d] outputs: head, tail and a strided middle."""
    idx = sorted(set(list(range(min(8, n))) + list(range(max(0, n - 8), n)) + list(range(0, n, max(1, n // 16)))))
    return np.array(idx, dtype=np.int64)


def forward_case(name, cfgname, n, seed_x, bf16=False, levels=None, max_layers=None):
    cfg = dict(synth.CONFIGS[cfgname])
    if levels:
        cfg["num_quantizers"] = levels
    w = synth.weights(SEED_W, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    m = ref_model(cfg, w)
    assert np.array_equal(m.codebook.detach().numpy(), w["codebook"])
    x = synth.activations(seed_x, n, cfg["dim"])
    if bf16:
        x = synth.round_to_bf16(x)
    B = 4 if n % 4 == 0 else 1
    xt = torch.from_numpy(x).view(B, n // B, -1)
    with torch.inference_mode():
        kw = {} if max_layers is None else dict(max_layers=max_layers)
        q, idx = m(xt, **kw)
        dec = m.decode(idx) if max_layers is None else None
        dec_sub = m.decode(idx, layers=[0, 2]) if (max_layers is None and cfg["num_quantizers"] > 2) else None
    q = q.reshape(n, -1).numpy()
    idx = idx.reshape(n, -1).numpy()
    rows = sample_rows(n)
    out = dict(
        meta=json.dumps(dict(name=name, cfg=cfg, cfgname=cfgname, n=n, seed_x=seed_x, seed_w=SEED_W, bf16=bf16,
                             max_layers=max_layers, torch=torch.__version__, numpy=np.__version__)),
        x_sha256=synth.sha256_of(x),
        codebook_sha256=synth.sha256_of(w["codebook"]),
        codes=idx.astype(np.uint16),
        rows=rows,
        q_rows=q[rows],
        q_rownorm=np.sqrt((q.astype(np.float64) ** 2).sum(-1)).astype(np.float32),
    )
    if dec is not None:
        out["dec_rows"] = dec.reshape(n, -1).numpy()[rows]
    if dec_sub is not None:
        out["dec02_rows"] = dec_sub.reshape(n, -1).numpy()[rows]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "codes", idx.shape, "max", idx.max(), "file KB",
          os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024)


def free_codebook_case(name, n, seed_x):
    """quantization_method outside {fsq, round_fsq}: per-level random unit codebooks
    (rqae/model.py:46-54,126-131); decode still reads codebook[0] (model.py:234)."""
    cfg = dict(dim=512, codebook_dim=6, codebook_size=300, num_quantizers=5)
    w = synth.weights(SEED_W + 1, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], free_codebook_rows=300)
    m = ref_model(cfg, w, method="vq")
    # the ctor + every quantize() call renormalise the codebook (model.py:61,126-131,196)
    x = synth.activations(seed_x, n, cfg["dim"])
    used = []
    inner = m.quantize_gumbel

    def spy(xq, layer_idx, temperature=0.0):  # record the codebook each level actually saw
        used.append(m.codebook[layer_idx].detach().numpy().copy())
        return inner(xq, layer_idx, temperature)

    m.quantize_gumbel = spy
    with torch.inference_mode():
        q, idx = m(torch.from_numpy(x).view(1, n, -1))
        cb_after = m.codebook.detach().numpy().copy()
        dec = m.decode(idx)
    rows = sample_rows(n)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        meta=json.dumps(dict(name=name, cfg=cfg, n=n, seed_x=seed_x, seed_w=SEED_W + 1, method="vq")),
        x_sha256=synth.sha256_of(x), codebook_after=cb_after, codebook_used=np.stack(used),
        codes=idx.reshape(n, -1).numpy().astype(np.uint16), rows=rows,
        q_rows=q.reshape(n, -1).numpy()[rows], dec_rows=dec.reshape(n, -1).numpy()[rows])
    print(name, "cb renorm changed bits:", int((cb_after != w["codebook"]).sum()))


def kat_case():
    """Known answers on the default (pretrained-shape) codebook: SURVEY 8(c) i-ii."""
    m = RefRQAE(dim=64, codebook_dim=4, codebook_size=5, num_quantizers=2).eval()
    p = torch.tensor([[[1.0, 0, 0, 0], [0, 0, 0, 0], [0.3, -0.2, 0.9, 0.1], [-1, -1, -1, -1],
                       [1e-30, 0, 0, 0], [float("nan"), 1, 0, 0], [float("inf"), 1, 0, 0]]])
    with torch.inference_mode():
        cv, idx = m.quantize(p, 0)
    rng = np.random.Generator(np.random.PCG64(77))
    pr = rng.standard_normal((1, 512, 4)).astype(np.float32)
    with torch.inference_mode():
        cvr, idxr = m.quantize(torch.from_numpy(pr), 1)
    np.savez_compressed(os.path.join(HERE, "kat_quantize.npz"),
                        p=p.numpy(), cv=cv.numpy(), idx=idx.numpy(),
                        pr=pr, cvr=cvr.numpy(), idxr=idxr.numpy(),
                        codebook=m.codebook.detach().numpy()[0])
    print("kat idx", idx.numpy().tolist())


class FakeGemma:
    """Stand-in for rqae/llm.py:Gemma2 exposing the two callables RQAE.hook pulls
    (rqae/model.py:254-260): norm = Gemma2RMSNorm math, denorm = llm.py:68-73."""

    def __init__(self, weight):
        self.weight = weight

    def norm(self, hs):
        x = hs.float()
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
        return x * (1.0 + self.weight.float())

    def denorm(self, hidden_states, original_hidden_states):
        hidden_states = hidden_states / (1.0 + self.weight.float())
        hidden_states = hidden_states.float() / torch.rsqrt(
            original_hidden_states.float().pow(2).mean(-1, keepdim=True) + 1e-6)
        return hidden_states


def hook_case():
    cfg = dict(synth.CONFIGS["C2"])
    w = synth.weights(SEED_W, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    m = ref_model(cfg, w)
    B, S, d = 2, 12, cfg["dim"]
    rng = np.random.Generator(np.random.PCG64(4242))
    hs = (synth.activations(555, B * S, d) * 3.0).astype(np.float16).reshape(B, S, d)
    nw = (0.1 * rng.standard_normal(d)).astype(np.float32)
    llm = FakeGemma(torch.from_numpy(nw))
    store = {}
    out0 = torch.from_numpy(hs.copy())
    fn = m.hook(llm=llm, store=lambda k, v: store.update({k: v}))
    with torch.inference_mode():
        fn(None, None, (out0,))
    np.savez_compressed(
        os.path.join(HERE, "hook_c2.npz"),
        meta=json.dumps(dict(cfgname="C2", B=B, S=S, seed_w=SEED_W)),
        hs=hs, norm_weight=nw,
        normed=store["normed"].numpy(), quantized=store["quantized"].numpy(),
        indices=store["indices"].numpy().astype(np.uint16), new=store["new"].numpy(),
        patched=out0.numpy())
    print("hook_c2 written; patched dtype", out0.dtype)


def feature_case():
    """RQAEFeature.intensity (rqae/feature.py:102-129) on the pretrained shape's codebook (K = 625, c = 4)."""
    from rqae.feature import RQAEFeature as RefFeature
    cfg = dict(dim=256, codebook_dim=4, codebook_size=5, num_quantizers=48)
    w = synth.weights(SEED_W, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    m = ref_model(cfg, w)
    rng = np.random.Generator(np.random.PCG64(777))
    L, K = cfg["num_quantizers"], m.codebook.shape[1]
    center = rng.integers(0, K, size=L).astype(np.int32)
    tok = rng.integers(0, K, size=(4, 32, L)).astype(np.int32)
    layers = [0, 5, 17, L - 1]
    feat = RefFeature.from_quantizer(m, center=center, layers=layers)
    with torch.inference_mode():
        out = feat.intensity(torch.from_numpy(tok), layers=layers)
        out_default = RefFeature.from_quantizer(m, center=center).intensity(torch.from_numpy(tok))
        out_sub = feat.intensity(torch.from_numpy(tok), layers=[3])
    assert out.dtype == torch.float16
    np.savez_compressed(
        os.path.join(HERE, "feat_p.npz"),
        meta=json.dumps(dict(cfg=cfg, seed_w=SEED_W, layers=layers)),
        codebook_sims=m.codebook_sims.numpy(), layer_weights=feat.layer_weights.numpy(), center=center,
        token_indices=tok, intensity=out.numpy(), intensity_default=out_default.numpy(), intensity_l3=out_sub.numpy())
    print("feat_p written", out.shape, out.dtype, os.path.getsize(os.path.join(HERE, "feat_p.npz")))


def tables_case():
    """The cached tables (rqae/model.py:133-178): codebook_sims, subfeatures, subfeature_sims, layer_norms."""
    out = {}
    cases = dict(c1=("C1", None), c2=("C2", 2), p=("P", 3), c5=("C5", 2))
    for tag, (cfgname, levels) in cases.items():
        cfg = dict(synth.CONFIGS[cfgname])
        if levels:
            cfg["num_quantizers"] = levels
        w = synth.weights(SEED_W, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
        m = ref_model(cfg, w)
        with torch.inference_mode():
            sub = m.subfeatures.numpy()
            sims = m.subfeature_sims.numpy()
            cbs = m.codebook_sims.numpy()
            ln = m.layer_norms.numpy()
        K = sub.shape[1]
        rows = sample_rows(K)
        assert sims.dtype == np.float16 and cbs.dtype == np.float16 and ln.dtype == np.float32
        out[tag + "_cfg"] = json.dumps(cfg)
        out[tag + "_rows"] = rows
        out[tag + "_subfeatures_rows"] = sub[:, rows]          # (L, rows, d) fp32
        out[tag + "_subfeature_sims_rows"] = sims[:, rows]     # (L, rows, K) fp16
        out[tag + "_codebook_sims_rows"] = cbs[rows]           # (rows, K) fp16
        out[tag + "_layer_norms"] = ln
        print(tag, sub.shape, sims.shape, cbs.shape, ln)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), meta=json.dumps(dict(seed_w=SEED_W)), **out)
    print("tables written", os.path.getsize(os.path.join(HERE, "tables.npz")))


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "tables":
        tables_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "feature":
        feature_case()
        sys.exit(0)
    forward_case("fwd_c1", "C1", 4096, 1000)
    forward_case("fwd_c2", "C2", 4096, 1000)
    forward_case("fwd_c2_bf16", "C2", 2048, 1001, bf16=True)
    forward_case("fwd_c2_max3", "C2", 512, 1002, max_layers=3)
    forward_case("fwd_c4", "C4", 512, 1003, levels=4)
    forward_case("fwd_c5", "C5", 1024, 1004, levels=8)
    forward_case("fwd_p", "P", 2048, 1005)
    forward_case("fwd_ragged", "C1", 37, 1006)
    free_codebook_case("fwd_vq", 1024, 1007)
    kat_case()
    hook_case()
    feature_case()
    tables_case()
