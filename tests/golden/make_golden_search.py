"""Capture golden vectors for the similarity search from the reference's own functions.

Runs ONLY in the build container.  /root/reference/demo/server/server.py cannot be imported whole (its first import is the
`modal` client, which this image does not have), so this script parses the file and executes exactly two function
definitions from it -- `get_intensities` (:41-68) and `IntensityEngine.find_examples` (:160-325, taken without its
decorator) -- on CPU tensors.  Nothing of the file is kept: the fixture holds seeds, shapes and the outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_search.py
"""
import ast
import os
import sys
from typing import List  # noqa: F401  (the reference's signature names it)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
SERVER = "/root/reference/demo/server/server.py"


def reference_functions():
    tree = ast.parse(open(SERVER).read(), SERVER)
    picked = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "get_intensities":
            picked.append(node)
        if isinstance(node, ast.ClassDef) and node.name == "IntensityEngine":
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == "find_examples":
                    sub.decorator_list = []
                    picked.append(sub)
    assert [n.name for n in picked] == ["get_intensities", "find_examples"]
    mod = ast.Module(body=picked, type_ignores=[])
    ns = {"torch": torch, "List": List}
    exec(compile(mod, SERVER, "exec"), ns)
    return ns["get_intensities"], ns["find_examples"]


def synth_case(seed, n_shards, per_shard, S, L, K):
    """Dataset codes, a scaled fp16 table and nothing else; regenerated from the seed by the tests."""
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, K, size=(n_shards * per_shard, S, L), dtype=np.int32)
    # cosine-like values times a per-level norm that decays with depth, as `sims *= layer_norms` leaves them
    sims = rng.uniform(-1.0, 1.0, size=(L, K, K)).astype(np.float32)
    norms = (8.0 * 0.97 ** np.arange(L)).astype(np.float32)
    sims16 = (sims.astype(np.float16).astype(np.float32) * norms[:, None, None]).astype(np.float16)
    return codes, sims16


class Engine:
    """What find_examples reads from `self` (server.py:99-141)."""

    def __init__(self, sims16, shards):
        self.sims = torch.from_numpy(sims16)
        self.activations = [torch.from_numpy(s) for s in shards]


def main():
    get_intensities, find_examples = reference_functions()
    torch.Tensor.cuda = lambda self, *a, **k: self          # the reference moves slices to its GPU; here they stay on the CPU
    out = {}

    # ---- get_intensities alone
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 9, size=(3, 5, 4), dtype=np.int32)
    sims = rng.standard_normal((6, 4, 9)).astype(np.float16)
    gi = get_intensities(torch.from_numpy(codes), torch.from_numpy(sims)).numpy()
    out["gi_codes"], out["gi_sims"], out["gi_out"] = codes, sims, gi

    # ---- find_examples: small (every range <= 64 levels) and deep (chunked ranges), dataset index and explicit activation
    cases = {
        "small": dict(seed=11, n_shards=3, per_shard=16, S=7, L=40, K=25, layers=[4, 6, 8, 12, 16, 24, 32, 40], idx=9,
                      top=5, mid=4, bot=3),
        "deep": dict(seed=12, n_shards=2, per_shard=24, S=9, L=300, K=25, layers=[4, 6, 8, 12, 16, 24, 32, 48, 64, 128, 299], idx=None,
                     top=6, mid=4, bot=4),
    }
    for name, c in cases.items():
        codes, sims16 = synth_case(c["seed"], c["n_shards"], c["per_shard"], c["S"], c["L"], c["K"])
        shards = np.split(codes, c["n_shards"])
        eng = Engine(sims16, shards)
        kw = dict(top_examples=c["top"], middle_examples=c["mid"], bottom_examples=c["bot"], layers=list(c["layers"]))
        if c["idx"] is not None:
            kw["idx"] = c["idx"]
        else:
            q = np.random.default_rng(c["seed"] + 100).integers(0, c["K"], size=(c["S"], c["L"]), dtype=np.int32)
            kw["activation"] = torch.from_numpy(q)
            out[f"{name}_query"] = q
        with torch.inference_mode():
            for res, layer in find_examples(eng, **kw):
                for part in ("top", "middle", "bottom"):
                    out[f"{name}_{layer}_{part}_indices"] = res[part]["indices"].numpy()
                    out[f"{name}_{layer}_{part}_intensities"] = res[part]["intensities"].numpy()
        out[f"{name}_cfg"] = np.array([c["seed"], c["n_shards"], c["per_shard"], c["S"], c["L"], c["K"], -1 if c["idx"] is None else c["idx"],
                                       c["top"], c["mid"], c["bot"]], dtype=np.int64)
        out[f"{name}_layers"] = np.array(c["layers"], dtype=np.int64)
    path = os.path.join(HERE, "search.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
