"""Similarity search: oracle vs the reference's own output, HIP path vs both.

tests/golden/search.npz was written by tests/golden/make_golden_search.py, which executes the reference's
``get_intensities`` and ``IntensityEngine.find_examples`` (demo/server/server.py:41-68, :160-325) on CPU tensors.

Parity bar: intensities bit-exact (fp16) against the oracle everywhere and against the reference capture on the fixtures;
ranks must hold the same maxima as the reference's (its ``argsort`` leaves the order of equal maxima open, ours is ascending
sequence number), and agree index for index wherever the maximum is unique.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import search_oracle as so  # noqa: E402  (the checker)

GOLD = np.load(os.path.join(HERE, "golden", "search.npz"))


def synth_case(seed, n_shards, per_shard, S, L, K):
    """Same construction as tests/golden/make_golden_search.py::synth_case."""
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, K, size=(n_shards * per_shard, S, L), dtype=np.int32)
    sims = rng.uniform(-1.0, 1.0, size=(L, K, K)).astype(np.float32)
    norms = (8.0 * 0.97 ** np.arange(L)).astype(np.float32)
    sims16 = (sims.astype(np.float16).astype(np.float32) * norms[:, None, None]).astype(np.float16)
    return codes, sims16


def fixture(name):
    seed, ns, ps, S, L, K, idx, top, mid, bot = [int(v) for v in GOLD[f"{name}_cfg"]]
    layers = [int(v) for v in GOLD[f"{name}_layers"]]
    codes, sims16 = synth_case(seed, ns, ps, S, L, K)
    query = codes[idx] if idx >= 0 else GOLD[f"{name}_query"]
    return dict(codes=codes, sims16=sims16, query=query, idx=idx, top=top, mid=mid, bot=bot, layers=layers, n_shards=ns)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


def check_against_capture(name, layer, out, acc, max_values):
    """`out` from our side, `acc` [N,S,Q] / `max_values` [N,Q] the accumulation it came from."""
    for part in ("top", "middle", "bottom"):
        ri, rv = GOLD[f"{name}_{layer}_{part}_indices"], GOLD[f"{name}_{layer}_{part}_intensities"]
        oi, ov = np.asarray(out[part]["indices"]), np.asarray(out[part]["intensities"])
        assert oi.shape == ri.shape and ov.shape == rv.shape
        Q = ri.shape[0]
        # the reference's examples carry exactly our intensities
        mine_at_ref = np.stack([acc[ri[q], :, q] for q in range(Q)])
        assert np.array_equal(bits(mine_at_ref), bits(rv)), (name, layer, part)
        # our examples' rows are the accumulation's rows
        assert np.array_equal(bits(ov), bits(np.stack([acc[oi[q], :, q] for q in range(Q)]))), (name, layer, part)
        for q in range(Q):
            # rank for rank the same maximum; the same sequence wherever that maximum is unique in the dataset
            assert np.array_equal(max_values[ri[q], q], max_values[oi[q], q]), (name, layer, part, q)
            col = max_values[:, q]
            unique = np.array([np.count_nonzero(col == v) == 1 for v in col[ri[q]]])
            assert np.array_equal(ri[q][unique], oi[q][unique]), (name, layer, part, q)


# ------------------------------------------------------------------------------------------------ CPU: oracle vs capture
def test_oracle_get_intensities_matches_reference():
    assert np.array_equal(bits(so.get_intensities(GOLD["gi_codes"], GOLD["gi_sims"])), bits(GOLD["gi_out"]))


@pytest.mark.parametrize("name", ["small", "deep"])
def test_oracle_find_examples_matches_reference(name):
    f = fixture(name)
    n = 0
    for out, layer, acc, mv in so.find_examples(f["sims16"], f["codes"], f["query"], f["top"], f["mid"], f["bot"], f["layers"]):
        check_against_capture(name, layer, out, acc, mv)
        n += 1
    assert n == len(f["layers"])


def test_get_intensities_mirror_matches_reference():
    """rqae_amd.search.get_intensities is index arithmetic in torch (no kernel): checked on the CPU."""
    from rqae_amd.search import get_intensities

    out = get_intensities(torch.from_numpy(GOLD["gi_codes"]), torch.from_numpy(GOLD["gi_sims"]))
    assert np.array_equal(bits(out.numpy()), bits(GOLD["gi_out"]))


# ------------------------------------------------------------------------------------------------ GPU: the HIP path
def run_engine(f, code_dtype=None, as_shards=True):
    from rqae_amd.search import IntensityEngine

    codes = torch.from_numpy(f["codes"])
    shards = list(torch.chunk(codes, f["n_shards"])) if as_shards else codes
    eng = IntensityEngine(torch.from_numpy(f["sims16"]), shards, device="cuda:0")
    if code_dtype is not None and eng.activations.dtype != code_dtype:
        eng.activations = torch.from_numpy(f["codes"]).to("cuda:0").to(code_dtype).contiguous()
    kw = dict(top_examples=f["top"], middle_examples=f["mid"], bottom_examples=f["bot"], layers=f["layers"])
    if f["idx"] >= 0:
        kw["idx"] = f["idx"]
    else:
        kw["activation"] = torch.from_numpy(f["query"])
    Q = f["query"].shape[0]
    for out, layer in eng.find_examples(**kw):
        yield out, layer, eng.accumulated(Q).cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small", "deep"])
@pytest.mark.parametrize("code_dtype", [torch.uint16, torch.int32])
def test_find_examples_fixtures(name, code_dtype):
    f = fixture(name)
    oracle_steps = so.find_examples(f["sims16"], f["codes"], f["query"], f["top"], f["mid"], f["bot"], f["layers"])
    n = 0
    for (out, layer, acc), (o_out, o_layer, o_acc, o_mv) in zip(run_engine(f, code_dtype), oracle_steps):
        assert layer == o_layer
        assert np.array_equal(bits(acc), bits(o_acc)), (name, layer)                 # the whole accumulation, bit for bit
        check_against_capture(name, layer, out, acc, acc.max(axis=1))                # the reference's own output
        for part in ("top", "middle", "bottom"):                                     # and the oracle's ranking, tie rule included
            assert np.array_equal(np.asarray(out[part]["indices"]), o_out[part]["indices"]), (name, layer, part)
            assert np.array_equal(bits(np.asarray(out[part]["intensities"])), bits(o_out[part]["intensities"]))
        n += 1
    assert n == len(f["layers"])


@pytest.mark.gpu
@pytest.mark.parametrize("code_dtype", [torch.uint16, torch.int32])
def test_find_examples_server_shape(code_dtype):
    """The server's own geometry (127 positions, K = 625) at a depth that exercises a chunked range with a ragged tail, with
    fp16 subnormals and zeros in the table (deep levels of the real table underflow like this)."""
    rng = np.random.default_rng(77)
    N, S, L, K = 40, 127, 150, 625
    codes = rng.integers(0, K, size=(N, S, L), dtype=np.int32)
    norms = (2.0 * 0.90 ** np.arange(L)).astype(np.float32)           # 2 * 0.9^149 = 3e-7: subnormal fp16 and below
    sims16 = (rng.uniform(-1, 1, size=(L, K, K)).astype(np.float32) * norms[:, None, None]).astype(np.float16)
    assert (np.abs(sims16[-1].astype(np.float32)) < 6.2e-5).all() and (sims16[-1] != 0).any()
    layers = [4, 6, 8, 12, 16, 24, 32, 48, 64, 149]
    f = dict(codes=codes, sims16=sims16, query=codes[3], idx=3, top=8, mid=6, bot=5, layers=layers, n_shards=2)
    oracle_steps = so.find_examples(sims16, codes, codes[3], 8, 6, 5, layers)
    for (out, layer, acc), (o_out, _, o_acc, _) in zip(run_engine(f, code_dtype), oracle_steps):
        assert np.array_equal(bits(acc), bits(o_acc)), layer
        for part in ("top", "middle", "bottom"):
            assert np.array_equal(np.asarray(out[part]["indices"]), o_out[part]["indices"]), (layer, part)
            assert np.array_equal(bits(np.asarray(out[part]["intensities"])), bits(o_out[part]["intensities"]))


@pytest.mark.gpu
def test_find_examples_ragged_and_small():
    """Fewer query positions than dataset positions, a dataset smaller than a workgroup's tile, an odd depth: the edges of the
    kernel's tiling (1024 tokens x 32 query positions per workgroup, 8-level code blocks)."""
    rng = np.random.default_rng(5)
    N, S, L, K, Q = 3, 5, 13, 9, 33
    codes = rng.integers(0, K, size=(N, S, L), dtype=np.int32)
    sims16 = rng.standard_normal((L, K, K)).astype(np.float16)
    query = rng.integers(0, K, size=(Q, L), dtype=np.int32)
    f = dict(codes=codes, sims16=sims16, query=query, idx=-1, top=2, mid=2, bot=1, layers=[1, 2, 9, 13], n_shards=1)
    oracle_steps = so.find_examples(sims16, codes, query, 2, 2, 1, f["layers"])
    for (out, layer, acc), (o_out, _, o_acc, _) in zip(run_engine(f, None, as_shards=False), oracle_steps):
        assert acc.shape == (N, S, Q)
        assert np.array_equal(bits(acc), bits(o_acc)), layer
        for part in ("top", "middle", "bottom"):
            assert np.array_equal(np.asarray(out[part]["indices"]), o_out[part]["indices"]), (layer, part)


@pytest.mark.gpu
def test_find_examples_full_shard_properties():
    """One full shard of the server's dataset (1024 x 127 positions) to depth 256: size-independent properties -- every
    ranking is a permutation in descending order of the accumulation's row maxima, and a strided sample of the
    accumulation matches the oracle bit for bit."""
    from rqae_amd.search import IntensityEngine

    g = torch.Generator(device="cpu").manual_seed(9)
    N, S, L, K = 1024, 127, 256, 625
    codes = torch.randint(0, K, (N, S, L), generator=g, dtype=torch.int32)
    sims16 = ((torch.rand((L, K, K), generator=g) * 2 - 1) * (4.0 * 0.985 ** torch.arange(L))[:, None, None]).to(torch.float16)
    eng = IntensityEngine(sims16, codes, device="cuda:0")
    layers = [4, 6, 8, 12, 16, 24, 32, 48, 64, 128, 256]
    sample = np.arange(0, N, 97)
    o_steps = so.find_examples(sims16.numpy(), codes.numpy()[sample], codes.numpy()[5], 1, 0, 1, layers)
    for (out, layer), (_, _, o_acc, _) in zip(eng.find_examples(idx=5, layers=layers), o_steps):
        acc = eng.accumulated(S)
        assert np.array_equal(bits(acc[sample].cpu().numpy()), bits(o_acc)), layer
        mx, order, _ = __import__("rqae_amd")._native.search_rank(eng._acc, S)      # the LDS bitonic rank on its own
        ref_max = acc.float().amax(dim=1).T                                           # [Q, N]
        assert torch.equal(torch.sort(order, dim=1).values, torch.arange(N, device=order.device, dtype=torch.int32).expand(S, N))
        assert torch.equal(mx.float(), torch.gather(ref_max, 1, order.long()))
        assert (mx[:, 1:].float() <= mx[:, :-1].float()).all()
        assert torch.equal(out["top"]["indices"], order[:, :30].cpu())


@pytest.mark.gpu
def test_find_examples_two_slot_ring():
    """K = 1024: the level's table slice (64 KB) leaves room for two ring slots, not three -- the other schedule of the kernel."""
    rng = np.random.default_rng(31)
    N, S, L, K = 20, 33, 70, 1024
    codes = rng.integers(0, K, size=(N, S, L), dtype=np.int32)
    sims16 = rng.standard_normal((L, K, K)).astype(np.float16)
    layers = [3, 8, 70]
    f = dict(codes=codes, sims16=sims16, query=codes[7], idx=7, top=4, mid=2, bot=3, layers=layers, n_shards=1)
    oracle_steps = so.find_examples(sims16, codes, codes[7], 4, 2, 3, layers)
    for (out, layer, acc), (o_out, _, o_acc, _) in zip(run_engine(f, torch.uint16, as_shards=False), oracle_steps):
        assert np.array_equal(bits(acc), bits(o_acc)), layer
        for part in ("top", "middle", "bottom"):
            assert np.array_equal(np.asarray(out[part]["indices"]), o_out[part]["indices"]), (layer, part)


@pytest.mark.gpu
def test_find_examples_is_repeatable():
    """Many passes over the same query leave the same bits (a timing-dependent error would differ from pass to pass:
    DESIGN.md section 10 has the one this round found in another kernel)."""
    from rqae_amd.search import IntensityEngine

    g = torch.Generator(device="cpu").manual_seed(4)
    N, S, L, K = 256, 127, 96, 625
    codes = torch.randint(0, K, (N, S, L), generator=g, dtype=torch.int32)
    sims16 = (torch.rand((L, K, K), generator=g) * 2 - 1).to(torch.float16)
    eng = IntensityEngine(sims16, codes, device="cuda:0")
    layers = [4, 16, 96]
    ref = None
    for rep in range(12):
        accs = []
        for out, layer in eng.find_examples(idx=9, layers=layers):
            accs.append((eng.accumulated(S).clone(), out["top"]["indices"].clone()))
        if ref is None:
            ref = accs
        for (a, i), (ra, ri) in zip(accs, ref):
            assert torch.equal(a.view(torch.int16), ra.view(torch.int16)) and torch.equal(i, ri), rep


@pytest.mark.gpu
def test_find_examples_errors():
    from rqae_amd.search import IntensityEngine

    codes = torch.zeros((4, 3, 8), dtype=torch.int32)
    sims = torch.zeros((8, 5, 5), dtype=torch.float16)
    eng = IntensityEngine(sims, codes, device="cuda:0")
    with pytest.raises(ValueError):
        next(eng.find_examples(idx=1, activation=codes[0], layers=[4]))
    with pytest.raises(ValueError):
        next(eng.find_examples(layers=[4]))
    with pytest.raises(ValueError):
        next(eng.find_examples(idx=1, layers=[4, 9]))
    with pytest.raises(ValueError):
        next(eng.find_examples(idx=1, layers=[4, 4]))
    with pytest.raises(IndexError):
        next(eng.find_examples(idx=7, layers=[4]))


@pytest.mark.gpu
@pytest.mark.parametrize("N,S,Q", [(1, 3, 5), (2, 1, 1), (47, 5, 127), (1024, 4, 127), (8192, 2, 9), (8193, 2, 3), (20000, 1, 4), (70001, 1, 2)])
def test_rank_is_the_stable_descending_order(N, S, Q):
    """rqae_hip_search_rank (the hand-written bitonic sort: one chunk in LDS up to 8192 sequences, chunks merged through global
    memory beyond) against numpy's stable argsort of the row maxima: descending, equal maxima in ascending sequence order, NaN
    on top, -0 ranked with +0; few distinct values so that ties are everywhere."""
    from rqae_amd import _native

    rng = np.random.default_rng(N * 131 + Q)
    vals = rng.integers(-6, 7, size=(N, S, 128)).astype(np.float16) * np.float16(0.5)
    vals[rng.random((N, S, 128)) < 0.02] = np.float16(-0.0)
    if N > 3:
        vals[N // 3, :, 0] = np.float16("nan")
        vals[N // 2, 0, 1 % Q] = np.float16("inf")
        vals[N - 1, :, 0] = np.float16("-inf")
    acc = torch.from_numpy(vals).to("cuda:0")
    mx, order, ws = _native.search_rank(acc, Q)
    torch.cuda.synchronize()
    m = np.fmax.reduce(vals.astype(np.float32), axis=1)[:, :Q]                  # [N, Q]; NaN takes the maximum, like torch.max
    nan_row = np.isnan(vals.astype(np.float32)).any(axis=1)[:, :Q]
    m[nan_row] = np.nan
    key = np.where(np.isnan(m), np.inf, m)
    tie_break = np.where(np.isnan(m), 1, 0)                                      # NaN ranks above +inf
    want = np.stack([np.lexsort((np.arange(N), -key[:, q], -tie_break[:, q])) for q in range(Q)])   # NaN, then descending, then n
    got = order.cpu().numpy()
    assert np.array_equal(got, want.astype(np.int32))
    gm = mx.cpu().numpy().astype(np.float32)
    wm = np.take_along_axis(m.T, want, axis=1)
    assert np.array_equal(np.isnan(gm), np.isnan(wm)) and np.array_equal(gm[~np.isnan(gm)], wm[~np.isnan(wm)])
    mx2, order2, _ = _native.search_rank(acc, Q, ws)                                 # workspace reuse: same answer
    assert torch.equal(order2, order)


@pytest.mark.gpu
def test_two_live_generators_do_not_share_state():
    """the server streams find_examples through remote_gen, so two generators of one engine can be live at once;
    each keeps its own accumulation (as the reference does, server.py:206).  Interleaved = run one after the other."""
    from rqae_amd.search import IntensityEngine

    g = torch.Generator(device="cpu").manual_seed(12)
    N, S, L, K = 96, 31, 40, 625
    codes = torch.randint(0, K, (N, S, L), generator=g, dtype=torch.int32)
    sims16 = (torch.rand((L, K, K), generator=g) * 2 - 1).to(torch.float16)
    eng = IntensityEngine(sims16, codes, device="cuda:0")
    layers = [4, 9, 40]
    alone_a = [(o["top"]["indices"].clone(), o["bottom"]["intensities"].clone()) for o, _ in eng.find_examples(idx=3, layers=layers)]
    alone_b = [(o["top"]["indices"].clone(), o["bottom"]["intensities"].clone()) for o, _ in eng.find_examples(idx=77, layers=layers)]
    ga, gb = eng.find_examples(idx=3, layers=layers), eng.find_examples(idx=77, layers=layers)
    for step in range(len(layers)):
        oa, _ = next(ga)
        ob, _ = next(gb)
        assert torch.equal(oa["top"]["indices"], alone_a[step][0]) and torch.equal(oa["bottom"]["intensities"], alone_a[step][1])
        assert torch.equal(ob["top"]["indices"], alone_b[step][0]) and torch.equal(ob["bottom"]["intensities"], alone_b[step][1])


@pytest.mark.gpu
def test_example_slices_follow_the_reference():
    """server.py:268-280 slices the ranking with [:top], [mid0:mid1] and [-bottom:]; `[-0:]` is the whole ranking."""
    from rqae_amd.search import IntensityEngine

    g = torch.Generator(device="cpu").manual_seed(13)
    N, S, L, K = 12, 7, 8, 50
    codes = torch.randint(0, K, (N, S, L), generator=g, dtype=torch.int32)
    sims16 = (torch.rand((L, K, K), generator=g) * 2 - 1).to(torch.float16)
    eng = IntensityEngine(sims16, codes, device="cuda:0")
    full = np.arange(N)
    for top, mid, bot in ((3, 2, 0), (0, 0, 2), (20, 30, 40), (1, 3, 1)):
        out, _ = next(eng.find_examples(idx=2, top_examples=top, middle_examples=mid, bottom_examples=bot, layers=[8]))
        assert out["top"]["indices"].shape[1] == len(full[:top])
        assert out["middle"]["indices"].shape[1] == len(full[N // 2 - mid // 2: N // 2 + mid // 2])
        assert out["bottom"]["indices"].shape[1] == len(full[-bot:])
        assert out["bottom"]["intensities"].shape == (S, len(full[-bot:]), S)


@pytest.mark.gpu
def test_bad_dataset_codes_and_examples_raise_index_error():
    """A dataset code outside the table: the reference's gather raises IndexError (server.py:55); here the accumulate kernel
    flags the call (include/rqae_hip.h "Status word") and the wrapper raises IndexError; the process carries on.  Same for an
    example index outside the dataset in the gather."""
    from rqae_amd import _native
    from rqae_amd.search import IntensityEngine

    g = torch.Generator(device="cpu").manual_seed(14)
    N, S, L, K = 40, 9, 16, 625
    codes = torch.randint(0, K, (N, S, L), generator=g, dtype=torch.int32)
    sims16 = (torch.rand((L, K, K), generator=g) * 2 - 1).to(torch.float16)
    good = [o["top"]["indices"].clone() for o, _ in IntensityEngine(sims16, codes, device="cuda:0").find_examples(idx=1, layers=[4, 16])]
    for bad_value, dt in ((K, torch.int32), (K + 3, torch.int32), (70000, torch.int32), (-1, torch.int32)):
        bad = codes.clone()
        bad[17, 4, 9] = bad_value
        if bad_value < 0 or bad_value > 65535:
            with pytest.raises(IndexError):
                IntensityEngine(sims16, bad, device="cuda:0")                   # caught at load (the uint16 fold would hide it)
            continue
        eng = IntensityEngine(sims16, bad, device="cuda:0")
        gen = eng.find_examples(idx=1, layers=[4, 16])
        next(gen)                                                                # levels 0..3: clean
        with pytest.raises(IndexError, match="out of range"):
            next(gen)                                                            # level 9 lies in the second range
    eng = IntensityEngine(sims16, codes, device="cuda:0")
    assert [o["top"]["indices"] for o, _ in eng.find_examples(idx=1, layers=[4, 16])][1].equal(good[1])
    acc = torch.zeros((N, S, 128), dtype=torch.float16, device="cuda:0")
    ex = torch.tensor([[0, N - 1, N]] * 5, dtype=torch.int32, device="cuda:0")
    with pytest.raises(IndexError, match="outside the dataset"):
        _native.search_gather(acc, 5, ex)
    with pytest.raises(IndexError):
        next(eng.find_examples(activation=torch.full((S, L), K, dtype=torch.int32), layers=[4]))    # the query's own codes
