"""CPU: the oracle (oracle/rqae_oracle.c) against golden vectors captured from the
reference (tests/golden/make_golden.py).  This is what pins the oracle.

Bar: codes identical on every fixture token; reconstruction within 1e-5 relative
(BASELINE.json north_star), in practice ~2e-7.
"""
import numpy as np
import pytest

import oracle
from conftest import FORWARD_CASES, case_inputs, load_golden, rel_err
from rqae_amd import synth

RTOL = 1e-5  # north_star: "reconstructed activations match within 1e-5 relative fp32"


@pytest.mark.parametrize("name", FORWARD_CASES)
def test_forward_matches_reference(name):
    g = load_golden(name)
    cfg, w, x = case_inputs(g)
    q, idx, mg = oracle.forward(x, w, max_layers=g["meta"]["max_layers"], want_margins=True)
    assert idx.shape == g["codes"].shape
    assert np.array_equal(idx, g["codes"].astype(np.int64)), (
        f"{name}: {(idx != g['codes']).any(-1).sum()} tokens differ from the reference; "
        f"min margin {mg.min():.3g}")
    rows = g["rows"]
    assert rel_err(q[rows], g["q_rows"]) < RTOL
    rn = np.sqrt((q.astype(np.float64) ** 2).sum(-1))
    assert np.allclose(rn, g["q_rownorm"], rtol=RTOL, atol=0)


@pytest.mark.parametrize("name", ["fwd_c1", "fwd_c2", "fwd_p", "fwd_c4"])
def test_decode_matches_reference(name):
    g = load_golden(name)
    cfg, w, x = case_inputs(g)
    codes = g["codes"].astype(np.int64)
    dec = oracle.decode(codes, w)
    # decode has no order-dependent reduction: bit-equal to ATen on the sampled rows
    assert np.array_equal(dec[g["rows"]], g["dec_rows"])
    dec02 = oracle.decode(codes, w, layers=[0, 2])
    assert np.array_equal(dec02[g["rows"]], g["dec02_rows"])
    assert oracle.decode(codes, w, layers=[]) is None


def test_max_layers_prefix_property():
    g = load_golden("fwd_c2_max3")
    cfg, w, x = case_inputs(g)
    _, full = oracle.forward(x, w)
    _, cut = oracle.forward(x, w, max_layers=3)
    assert np.array_equal(full[:, :3], cut)


def test_decode_close_to_forward_recon():
    # decode(forward(x).idx) differs from q_out only by the STE rounding
    g = load_golden("fwd_c2")
    cfg, w, x = case_inputs(g)
    q, idx = oracle.forward(x[:256], w)
    dec = oracle.decode(idx, w)
    assert rel_err(dec, q) < 1e-6


def test_quantize_known_answers():
    k = load_golden("kat_quantize")
    cv, idx = oracle.quantize(k["p"], k["codebook"])
    assert idx.tolist() == k["idx"].tolist() == [[437, 0, 447, 0, 0, 0, 0]]  # tie 437/562 -> lowest; 0-vector/NaN/inf -> 0
    assert np.array_equal(cv, k["cv"])
    cvr, idxr = oracle.quantize(k["pr"], k["codebook"])
    assert np.array_equal(idxr, k["idxr"])
    assert np.array_equal(cvr, k["cvr"])


def test_duplicate_rows_lowest_index():
    cb = synth.fsq_codebook(4, 5)
    # 625 rows, only 545 distinct directions; the zero row is 312
    assert len({tuple(r) for r in cb.tolist()}) == 545
    assert not cb[312].any()
    rng = np.random.Generator(np.random.PCG64(3))
    p = rng.standard_normal((4096, 4)).astype(np.float32)
    _, idx = oracle.quantize(p, cb)
    first = {}
    for i, r in enumerate(cb.tolist()):
        first.setdefault(tuple(r), i)
    assert all(first[tuple(cb[i].tolist())] == i for i in idx.tolist())


def test_free_codebook_case():
    g = load_golden("fwd_vq")
    meta = g["meta"]; cfg = meta["cfg"]
    w = synth.weights(meta["seed_w"], cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"],
                      free_codebook_rows=cfg["codebook_size"])
    x = synth.activations(meta["seed_x"], meta["n"], cfg["dim"])
    assert synth.sha256_of(x) == str(g["x_sha256"])
    w_used = dict(w, codebook=g["codebook_used"])  # what each level saw after the in-place renormalisations
    q, idx = oracle.forward(x, w_used)
    assert np.array_equal(idx, g["codes"].astype(np.int64))
    assert rel_err(q[g["rows"]], g["q_rows"]) < RTOL
    # decode reads codebook[0] of the *final* parameter state for every level (model.py:234)
    w_after = dict(w, codebook=g["codebook_after"])
    dec = oracle.decode(g["codes"].astype(np.int64), w_after)
    assert np.array_equal(dec[g["rows"]], g["dec_rows"])


def test_hook_fixture_forward():
    h = load_golden("hook_c2")
    cfg = synth.CONFIGS["C2"]
    w = synth.weights(h["meta"]["seed_w"], cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    q, idx = oracle.forward(h["normed"], w)
    assert np.array_equal(idx, h["indices"].astype(np.int64))
    assert rel_err(q, h["quantized"]) < RTOL
    new = oracle.gemma_denorm(q, h["hs"].astype(np.float32), h["norm_weight"])
    new[:, 0] = h["hs"][:, 0].astype(np.float32)
    assert rel_err(new, h["new"]) < RTOL
    assert np.array_equal(h["patched"], h["new"].astype(np.float16))  # output[0].data.copy_(q_out), fp16


def test_hook_fixture_at_the_production_call_shape():
    """the reference's own hook capture at B=4, S=128 (scripts/1_create_activations.py:152) -- every code of
    the 512 tokens, and normed / quantized / new / patched on sampled rows (BOS rows included)."""
    from conftest import hook_4x128_inputs

    h, hs = hook_4x128_inputs()
    cfg = synth.CONFIGS["C2"]
    w = synth.weights(h["meta"]["seed_w"], cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    B, S, d = hs.shape
    rows = h["rows"]
    normed = oracle.gemma_norm(hs.astype(np.float32), h["norm_weight"]).reshape(B * S, d)
    assert rel_err(normed[rows], h["normed_rows"]) < 1e-6
    q, idx = oracle.forward(normed, w)
    assert np.array_equal(idx.reshape(B, S, -1), h["indices"].astype(np.int64))
    assert rel_err(q[rows], h["quantized_rows"]) < RTOL
    new = oracle.gemma_denorm(q.reshape(B, S, d), hs.astype(np.float32), h["norm_weight"])
    new[:, 0] = hs[:, 0].astype(np.float32)
    new = new.reshape(B * S, d)
    assert rel_err(new[rows], h["new_rows"]) < RTOL
    assert np.array_equal(h["patched_rows"], h["new_rows"].astype(np.float16))
    assert np.array_equal(h["patched_rows"][np.isin(rows, np.arange(0, B * S, S))], hs.reshape(B * S, d)[np.arange(0, B * S, S)])   # BOS rows kept


def test_empty_and_single_token():
    cfg = synth.CONFIGS["C1"]
    w = synth.weights(1, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    q, idx = oracle.forward(np.zeros((0, cfg["dim"]), np.float32), w)
    assert q.shape == (0, cfg["dim"]) and idx.shape == (0, 4)
    q, idx = oracle.forward(np.zeros((1, cfg["dim"]), np.float32), w)
    assert np.isfinite(q).all()


def test_canonical_gemma_norm_against_reference_capture():
    """The hook pre/post restatement (oracle.gemma_*_canonical) against the reference's hook capture."""
    h = load_golden("hook_c2")
    cfg = synth.CONFIGS["C2"]
    w = synth.weights(h["meta"]["seed_w"], cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    normed, rs = oracle.gemma_norm_canonical(h["hs"], h["norm_weight"])
    assert rel_err(normed, h["normed"]) < 1e-6
    q, idx = oracle.forward(normed, w)
    assert np.array_equal(idx, h["indices"].astype(np.int64))
    new = oracle.gemma_denorm_canonical(q, rs, h["norm_weight"])
    new[:, 0] = h["hs"][:, 0].astype(np.float32)
    assert rel_err(new, h["new"]) < 1e-5
    assert (new.astype(np.float16) != h["patched"]).mean() < 1e-3      # fp16 write-back: at most a rare last-bit flip


def test_intensity_against_reference_capture():
    """the oracle's RQAEFeature.intensity restatement equals the reference's fp16 output bit for bit."""
    g = load_golden("feat_p")
    layers = g["meta"]["layers"]
    L = g["center"].shape[0]
    out = oracle.intensity(g["codebook_sims"], g["center"], g["layer_weights"], g["token_indices"], layers)
    assert out.dtype == np.float16 and np.array_equal(out.view(np.uint16), g["intensity"].view(np.uint16))
    out = oracle.intensity(g["codebook_sims"], g["center"], g["layer_weights"], g["token_indices"], [L - 1])
    assert np.array_equal(out.view(np.uint16), g["intensity_default"].view(np.uint16))
    out = oracle.intensity(g["codebook_sims"], g["center"], g["layer_weights"], g["token_indices"], [3])
    assert np.array_equal(out.view(np.uint16), g["intensity_l3"].view(np.uint16))


TABLE_CASES = ["c1", "c2", "p", "c5"]


def _tables_case(g, tag):
    import json
    cfg = json.loads(str(g[tag + "_cfg"]))
    w = synth.weights(g["meta"]["seed_w"], cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
    return cfg, w, g[tag + "_rows"]


@pytest.mark.parametrize("tag", TABLE_CASES)
def test_tables_against_reference_capture(tag):
    """codebook_sims and subfeatures restated with the reference's bits; subfeature_sims / layer_norms
    (BLAS-ordered in the reference) restated in double, within the reference's own fp32 rounding of the capture."""
    g = load_golden("tables")
    cfg, w, rows = _tables_case(g, tag)
    cbs = oracle.codebook_sims(w["codebook"][0])
    assert np.array_equal(cbs[rows].view(np.uint16), g[tag + "_codebook_sims_rows"].view(np.uint16))
    sub = oracle.subfeatures(w)
    assert np.array_equal(sub[:, rows], g[tag + "_subfeatures_rows"])
    ref = g[tag + "_subfeature_sims_rows"]
    mine = oracle.subfeature_sims(w)[:, rows]
    ulp = np.spacing(np.abs(ref)).astype(np.float64)
    assert (np.abs(mine - ref.astype(np.float64)) <= 0.5 * ulp + 2e-6).all()        # fp16 rounding + fp32 product
    assert (mine.astype(np.float16).view(np.uint16) == ref.view(np.uint16)).mean() > 0.999
    assert np.abs(oracle.layer_norms(w) / g[tag + "_layer_norms"] - 1).max() < 1e-6


@pytest.mark.parametrize("name", ["fwd_c1", "fwd_c2", "fwd_c2_max3", "fwd_c5", "fwd_p", "fwd_ragged", "fwd_p128"])
def test_torch_ops_restatement_matches_reference(name):
    """oracle/torch_ops.py (bench.py's cpu_baseline of kind "reference-ops": the reference's operator sequence) against
    the reference captures: the same ATen operators on the same values, so codes AND reconstruction rows are the capture's bits."""
    from oracle import torch_ops

    g = load_golden(name)
    cfg, w, x = case_inputs(g)
    q, idx = torch_ops.forward(x, w, max_layers=g["meta"]["max_layers"])
    assert np.array_equal(idx, g["codes"].astype(np.int64))
    assert np.array_equal(q[g["rows"]], g["q_rows"])
