"""CPU: host-side mirror of the reference interface, the C-ABI symbol table, and the
no-fallback rule.  No kernel is launched here."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from rqae_amd import RQAE, _native, synth


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rqae_hip.h")).read()
    declared = set(re.findall(r"\b(rqae_hip_\w+)\s*\(", hdr))
    assert declared == set(_native.SYMBOLS)
    lib = ctypes.CDLL(_native.SO_PATH)
    for s in declared:
        assert hasattr(lib, s), s
    assert _native.lib().rqae_hip_abi_version() == _native.ABI_VERSION


def test_packed_bytes_and_shape_limits():
    assert _native.packed_bytes(2304, 5, 1024, 8) > 8 * (2 * 5 * 2304 + 2304 + 5 + 1024 * 5) * 4
    assert _native.lib().rqae_hip_packed_bytes(9000, 5, 10, 1) == 0     # d > 8192
    assert _native.lib().rqae_hip_packed_bytes(256, 17, 10, 1) == 0     # c > 16
    assert b"unsupported" in _native.lib().rqae_hip_last_error()
    assert "P=9,C=5" in _native.kernel_info(2304, 5, 1024, 8)
    assert "generic" in _native.kernel_info(512, 6, 300, 5)


def test_state_dict_keys_and_ctor_defaults():
    m = RQAE(dim=32, codebook_dim=2, codebook_size=3, num_quantizers=3)
    keys = set(m.state_dict().keys())
    want = {"codebook", "codebook_counts"} | {f"layers.{l}.{i}.{p}" for l in range(3) for i in (0, 1) for p in ("weight", "bias")}
    assert keys == want
    assert m.codebook.shape == (3, 9, 2) and not m.codebook.requires_grad
    assert m.codebook_counts.shape == (3, 9)
    assert m.layers[0][0].weight.shape == (2, 32) and m.layers[0][1].weight.shape == (32, 2)
    assert (m.num_quantizers, m.codebook_dim, m.codebook_size, m.quantization_method, m.name) == (3, 2, 3, "round_fsq", "")
    d = RQAE.__init__.__defaults__
    assert d[:6] == (2304, 4, 5, 1024, "round_fsq", "")
    assert RQAE.PRETRAINED["google/gemma-2-2b"] == "harish-kamath/rqae/gemma-2-2b"


def test_fsq_codebook_matches_reference_fixture():
    k = load_golden("kat_quantize")
    m = RQAE(dim=8, codebook_dim=4, codebook_size=5, num_quantizers=2)
    assert np.array_equal(m.codebook[0].numpy(), k["codebook"])
    assert np.array_equal(m.codebook[1].numpy(), k["codebook"])
    assert np.array_equal(synth.fsq_codebook(4, 5), k["codebook"])
    plain = RQAE(dim=8, codebook_dim=2, codebook_size=3, num_quantizers=1, quantization_method="fsq")
    assert plain.codebook[0].abs().max() == 1.0
    free = RQAE(dim=8, codebook_dim=3, codebook_size=7, num_quantizers=2, quantization_method="vq")
    assert free.codebook.shape == (2, 7, 3) and free.codebook.requires_grad
    assert torch.allclose(free.codebook.norm(dim=-1), torch.ones(2, 7), atol=1e-6)


def test_metadata_properties_and_pickle():
    m = RQAE(dim=16, codebook_dim=2, codebook_size=3, num_quantizers=2).eval()
    assert m.codebook_sims.shape == (9, 9) and m.codebook_sims.dtype == torch.float16
    assert m.subfeatures.shape == (2, 9, 16)
    assert m.subfeature_sims.shape == (2, 9, 9) and m.subfeature_sims.dtype == torch.float16
    assert m.layer_norms.shape == (2,)
    m2 = pickle.loads(pickle.dumps(m))
    assert torch.equal(m2.codebook, m.codebook) and m2._packed is None
    with pytest.raises(ValueError):
        RQAE(dim=16, codebook_dim=2, codebook_size=3, num_quantizers=1, quantization_method="fsq").codebook_sims
    assert m.update_codebook_counts(torch.zeros(1, 1, 2)) is None


def test_no_cpu_fallback():
    m = RQAE(dim=256, codebook_dim=4, codebook_size=4, num_quantizers=2).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 256))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.decode(torch.zeros(1, 3, 2, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.quantize(torch.zeros(1, 3, 4), 0)


def test_training_branch_runs_on_torch_ops():
    """Training with gradients / temperature >= 1e-7 (rqae/model.py:183-185, 218-224) is served by torch ops: the
    straight-through estimator trains lin_in and lin_out, the gumbel branch samples; eval calls never take it."""
    torch.manual_seed(0)
    m = RQAE(dim=32, codebook_dim=3, codebook_size=3, num_quantizers=3).train()
    x = torch.randn(2, 5, 32)
    q, idx = m(x, temperature=0.7)
    assert q.shape == (2, 5, 32) and idx.shape == (2, 5, 3) and idx.dtype == torch.int64 and q.requires_grad
    q.square().mean().backward()
    assert m.layers[0][0].weight.grad is not None and m.layers[2][1].weight.grad.abs().sum() > 0
    # temperature 0 while training: deterministic argmax, still differentiable (straight-through)
    q0, idx0 = m(x)
    q1, idx1 = m(x)
    assert torch.equal(idx0, idx1) and q0.requires_grad
    # the same values as a hand-written reference level loop
    r, tot = x, 0
    for l in range(3):
        p = m.layers[l][0](r)
        u = p / p.norm(dim=-1, keepdim=True)
        k = (u @ m.codebook[l].T).argmax(-1)
        assert torch.equal(k, idx0[..., l])
        y = m.layers[l][1](p + (m.codebook[l][k] - p))
        r, tot = r - y, tot + y
    assert torch.allclose(tot, q0, atol=1e-6)
    cv, k = m.quantize(torch.randn(4, 3, requires_grad=True), 1, temperature=0.5)
    assert cv.shape == (4, 3) and k.shape == (4,)
    with torch.no_grad():            # no gradients wanted and deterministic -> the HIP path (raises on CPU)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m(x)


def test_weights_key_sees_every_parameter():
    m = RQAE(dim=16, codebook_dim=2, codebook_size=3, num_quantizers=5).eval()
    k0 = m._weights_key()
    assert m._weights_key() == k0
    with torch.no_grad():
        m.layers[3][0].weight.add_(1.0)          # a middle level, in place
    assert m._weights_key() != k0
    k1 = m._weights_key()
    m.layers[2][1].bias.data = m.layers[2][1].bias.data.clone()   # new storage
    assert m._weights_key() != k1
    # a module with more than 512 parameter tensors (the released 4097-tensor shape): by default EVERY version counter is read at
    # EVERY call (the reference reads weights live, rqae/model.py:208-221) -- an in-place edit of any level
    # changes the very next key, exactly once; only the storage POINTERS are re-read a sixteenth per call
    big = RQAE(dim=8, codebook_dim=2, codebook_size=2, num_quantizers=200).eval()
    k0 = big._weights_key()
    assert big._weights_key() == k0
    with torch.no_grad():
        big.layers[137][1].weight.add_(1.0)
    keys = [big._weights_key() for _ in range(20)]
    assert keys[0] != k0 and len(set(keys)) <= 2 and keys[-1] == keys[-2]      # seen at once; settles (no flapping)
    # re-binding a parameter's storage: seen within 16 calls by the pointer slices
    kk = big._weights_key()
    big.layers[90][0].bias.data = big.layers[90][0].bias.data.clone()
    assert any(big._weights_key() != kk for _ in range(16))
    # load_state_dict restarts the snapshots: ONE new key, not sixteen
    big.load_state_dict(big.state_dict())
    keys = [big._weights_key() for _ in range(18)]
    assert len(set(keys)) == 1
    # the explicit opt-in: versions too are sampled -- an edit is seen within 16 calls, at once after invalidate_packed()
    big.key_check = "sliced"
    k0 = big._weights_key()
    with torch.no_grad():
        big.layers[137][1].weight.add_(1.0)
    keys = [big._weights_key() for _ in range(16)]
    assert keys[-1] != k0 and len(set(keys)) == 2           # one change of key, no flapping
    with torch.no_grad():
        big.layers[5][0].bias.add_(1.0)
    big.invalidate_packed()
    assert big._packed is None and big._gram is None
    big.train()
    k1 = big._weights_key()
    with torch.no_grad():
        big.layers[77][0].weight.add_(1.0)
    assert big._weights_key() != k1
    with pytest.raises(ValueError):
        RQAE(dim=16, codebook_dim=2, codebook_size=3, num_quantizers=1, arm="fast")
    assert RQAE(dim=16, codebook_dim=2, codebook_size=3, num_quantizers=1, arm="gram").arm == "gram"


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "rqae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f
                assert "librqae_oracle" not in src, f


def test_hook_argument_protocol():
    m = RQAE(dim=16, codebook_dim=2, codebook_size=3, num_quantizers=1)
    with pytest.raises(ValueError, match="norm"):
        m.hook(denorm=lambda a, b: a)
    with pytest.raises(ValueError, match="denorm"):
        m.hook(norm=lambda a: a)
    with pytest.raises(AssertionError):
        m.hook(llm=object())
    assert callable(m.hook(norm=lambda a: a, denorm=lambda a, b: a))


def test_feature_mirror_host_side():
    """rqae_amd.RQAEFeature keeps the reference's constructor defaults, attributes and error behaviour
    (rqae/feature.py:42-137); the kernel itself is covered by the gpu tests."""
    import pickle

    import numpy as np
    import torch

    from rqae_amd import RQAE, Feature, RQAEFeature

    f = RQAEFeature(num_quantizers=6, dim=4)
    assert f.layers == [5] and f.center.dtype == torch.int32 and f.center.shape == (6,)
    assert f.layer_weights.shape == (6,) and f.explanations == [""] and f.scores == [{}] and f.activations == {5: []}
    with pytest.raises(ValueError):
        f.intensity(torch.zeros(2, 6, dtype=torch.int32))          # "Model not loaded"
    m = RQAE(dim=64, codebook_dim=2, codebook_size=3, num_quantizers=6).eval()
    g = RQAEFeature.from_quantizer(m, center=np.arange(6) % 9, layers=[1, 5])
    assert g.num_quantizers == 6 and g.dim == 2 and g.layer_weights.dtype == torch.float16
    want = torch.tensor([l[1].weight.data.norm(dim=0).mean().item() for l in m.layers]).to(torch.float16)
    assert torch.equal(g.layer_weights, want)
    with pytest.raises(RuntimeError):
        g.intensity(torch.zeros(2, 6, dtype=torch.int32))          # CPU-resident model: no CPU path
    h = pickle.loads(pickle.dumps(g))
    assert h.layers == [1, 5] and torch.equal(h.center, g.center) and h._prepared is None
    t = g.to_feature(1)
    assert isinstance(t, Feature) and t.activations == []


def test_feature_save_load_round_trip(tmp_path):
    """RQAEFeature.save / load (rqae/feature.py:139-153): what scripts 4-7 read back."""
    import numpy as np

    from rqae_amd import RQAEFeature

    f = RQAEFeature(id="tok-17", explanations=["first", "second"], scores=[{"a": 1.0}, {"b": 2.0}], model="rqae-x",
                    num_quantizers=6, dim=4, layers=[1, 5], layer_weights=np.linspace(0.5, 1.0, 6), center=np.arange(6))
    path = str(tmp_path / "feat.npz")
    f.save(path)
    g = RQAEFeature.load(path)
    assert g.id == "tok-17" and g.model == "rqae-x" and g.num_quantizers == 6 and g.dim == 4
    assert g.explanations == ["first", "second"] and all(isinstance(e, str) for e in g.explanations)
    assert list(g.layers) == [1, 5] and torch.equal(g.center, f.center) and torch.equal(g.layer_weights, f.layer_weights)
    assert list(g.scores) == [{"a": 1.0}, {"b": 2.0}]
    one = RQAEFeature(id=3, explanations=["only"], num_quantizers=2, dim=4)     # one-element arrays unwrap, as ndarray.item() does
    one.save(path)
    h = RQAEFeature.load(path)
    assert h.explanations == "only" and h.num_quantizers == 2


def test_from_pretrained_reads_the_hub_layout(tmp_path, monkeypatch):
    """RQAE.from_pretrained (rqae/model.py:75-99) against a local copy of the hub layout: <folder>/config.json +
    <folder>/model.safetensors, strict state-dict load, the name built from the config."""
    import json

    import huggingface_hub
    from safetensors.torch import save_file

    cfg = dict(dim=32, codebook_dim=3, codebook_size=3, num_quantizers=4, quantization_method="round_fsq")
    src = RQAE(**cfg)
    with torch.no_grad():
        for p in src.layers.parameters():
            p.uniform_(-0.3, 0.3)
    folder = tmp_path / "gemma-2-2b"
    folder.mkdir()
    save_file({k: v.contiguous() for k, v in src.state_dict().items()}, str(folder / "model.safetensors"))
    (folder / "config.json").write_text(json.dumps(cfg))
    asked = []

    def fake_download(repo_id, filename, **kw):
        asked.append((repo_id, filename))
        return str(tmp_path / filename)

    monkeypatch.setattr(huggingface_hub, "hf_hub_download", fake_download)
    m = RQAE.from_pretrained("google/gemma-2-2b")                       # alias of harish-kamath/rqae/gemma-2-2b
    assert asked == [("harish-kamath/rqae", "gemma-2-2b/model.safetensors"), ("harish-kamath/rqae", "gemma-2-2b/config.json")]
    assert m.name == "rqae-rqae-round_fsq-cbd3-cbs3-nq4" and m.num_quantizers == 4 and m.codebook.shape == (4, 27, 3)
    for k, v in src.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    bad = dict(cfg, num_quantizers=5)
    (folder / "config.json").write_text(json.dumps(bad))
    with pytest.raises(RuntimeError):                                    # strict load: a checkpoint of another depth is refused
        RQAE.from_pretrained("harish-kamath/rqae/gemma-2-2b")


def test_operators_are_registered():
    """torch.ops.rqae_hip.{encode, decode, quantize} exist with fake implementations (shape/dtype inference on meta)."""
    import rqae_amd  # noqa: F401

    x = torch.empty(5, 16, device="meta")
    w_in, b_in = torch.empty(3, 2, 16, device="meta"), torch.empty(3, 2, device="meta")
    cb, w_out, b_out = torch.empty(3, 9, 2, device="meta"), torch.empty(3, 16, 2, device="meta"), torch.empty(3, 16, device="meta")
    q, idx = torch.ops.rqae_hip.encode(x, w_in, b_in, cb, w_out, b_out, 2, True)
    assert q.shape == (5, 16) and q.dtype == torch.float32 and idx.shape == (5, 2) and idx.dtype == torch.int64
    q, idx = torch.ops.rqae_hip.encode(x, w_in, b_in, cb, w_out, b_out, 99, False)
    assert q.numel() == 0 and idx.shape == (5, 3)
    out = torch.ops.rqae_hip.decode(torch.empty(5, 3, dtype=torch.int64, device="meta"), cb[0], w_out, b_out, torch.empty(0, dtype=torch.bool, device="meta"))
    assert out.shape == (5, 16)
    cv, k = torch.ops.rqae_hip.quantize(torch.empty(7, 2, device="meta"), cb[0])
    assert cv.shape == (7, 2) and k.shape == (7,) and k.dtype == torch.int64


def test_operator_pack_cache_is_keyed_on_the_tensors_themselves():
    """callers of torch.ops.rqae_hip.encode stack their per-level weights per call, so the keyed tensors are
    temporaries; after an in-place update the next stack can land at the same address with version 0.  The cache must miss
    then: a hit needs the very same tensor objects at the same versions (the entry keeps them alive)."""
    from rqae_amd import ops

    ops._PACK_CACHE.clear()
    built = []

    def build():
        built.append(1)
        return len(built)

    srcs = [torch.randn(3, 4) for _ in range(4)]
    a = torch.stack(srcs)
    assert ops._cached_block("t", (a,), build) == 1
    assert ops._cached_block("t", (a,), build) == 1            # same object, same version: hit
    ptr = a.data_ptr()
    del a
    for s in srcs:
        s.add_(1.0)
    b = torch.stack(srcs)                                      # a new temporary (may or may not reuse the address)
    assert ops._cached_block("t", (b,), build) == 2            # never the stale block
    assert b.data_ptr() != ptr or True                         # the entry held `a` alive, so the address cannot repeat while cached
    b.mul_(2.0)
    assert ops._cached_block("t", (b,), build) == 3            # in-place edit of the keyed tensor: version moved
    assert ops._cached_block("u", (b,), build) == 4            # slots do not share entries
    with torch.inference_mode():
        c = torch.ones(2)
    assert ops._cached_block("t", (c,), build) == 5 and ops._cached_block("t", (c,), build) == 6   # no version counter: no caching
    ops._PACK_CACHE.clear()


def test_code_shards_round_trip(tmp_path):
    """The reference's int32 shard and the uint16 shard (half the bytes) hold the same codes; both load without pickle."""
    from rqae_amd import shards

    codes = torch.randint(0, 625, (6, 16, 40), dtype=torch.int64)
    p32, p16 = str(tmp_path / "a.pt"), str(tmp_path / "b.pt")
    shards.save_codes(p32, codes, dtype="int32")
    shards.save_codes(p16, codes, dtype="uint16", codebook_rows=625)
    a, b = shards.load_codes(p32), shards.load_codes(p16)
    assert a.dtype == torch.int32 and b.dtype == torch.uint16 and a.shape == b.shape == codes.shape
    assert torch.equal(a.long(), codes) and torch.equal(b.to(torch.int32).long(), codes)
    assert os.path.getsize(p16) < 0.6 * os.path.getsize(p32)
    with pytest.raises(ValueError):
        shards.save_codes(p16, codes + 70000, dtype="uint16")
    with pytest.raises(ValueError):
        shards.save_codes(p16, codes, dtype="int8")
