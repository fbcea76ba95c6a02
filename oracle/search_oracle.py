"""CPU restatement of the demo server's similarity search -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product (rqae_amd/) never does.

Follows /root/reference/demo/server/server.py:
  * get_intensities (:41-68): out[b, s, q, l] = sims[q, l, codes[b, s, l]] -- a gather, no arithmetic;
  * IntensityEngine.find_examples (:160-325): the query's table rows (:186-200), per layer range the sum over the range's
    levels (:214-252: one fp16 sum when the range has <= 64 levels, otherwise fp16 sums of 64-level chunks added in fp16),
    the running fp16 accumulation over ranges (:257-260), the maximum over dataset positions and the descending ordering of
    the sequences per query position (:263-266), the top / middle / bottom slices (:269-290) and the gathered
    intensities (:293-311).

Arithmetic pinned here (the reference leaves two things to the backend, see tests/test_search.py):
  * a sum of fp16 values over levels accumulates in fp32, levels in ascending order, and rounds once to fp16 (torch's
    half-precision sum accumulates in fp32 too; its order of additions is backend-specific);
  * equal maxima are ordered by ascending sequence number (torch.argsort without stable=True promises no order).
Pinned by tests/golden/search_*.npz, captured from the reference's own functions (tests/golden/make_golden_search.py).
"""
import numpy as np

CHUNK = 64   # server.py:216-218


def get_intensities(codes, sims):
    """server.py:41-68.  codes [B, S, D] int, sims [S2, D, K] -> [B, S, S2, D] (dtype of sims)."""
    B, S, D = codes.shape
    S2, D2, _ = sims.shape
    assert D == D2
    out = np.empty((B, S, S2, D), dtype=sims.dtype)
    lv = np.arange(D)
    for q in range(S2):
        out[:, :, q, :] = sims[q][lv[None, None, :], codes]
    return out


def query_rows(sims16, query_codes, max_level):
    """server.py:186-200: rows of the (scaled, fp16) table the query's codes select.  -> [max_level, Q, K] fp16."""
    qa = np.asarray(query_codes)[:, :max_level].T          # (levels, Q)
    return np.stack([sims16[l][qa[l]] for l in range(max_level)])


def _sum_levels_fp16(qrows, codes, l0, l1):
    """fp16( fp32 sum over l in [l0, l1), ascending, of qrows[l, q, codes[n, s, l]] ) -> [N, S, Q] fp16."""
    N, S, _ = codes.shape
    Q = qrows.shape[1]
    acc = np.zeros((N, S, Q), dtype=np.float32)
    for l in range(l0, l1):
        # qrows[l] is [Q, K]; gather column codes[n, s, l] for every q
        acc += qrows[l].T[codes[:, :, l]].astype(np.float32)      # [N, S, Q]
    return acc.astype(np.float16)


def range_intensities(qrows, codes, l0, l1):
    """server.py:214-252 for one layer range: chunked when the range has more than 64 levels."""
    if l1 - l0 > CHUNK:
        out = None
        for c0 in range(l0, l1, CHUNK):
            part = _sum_levels_fp16(qrows, codes, c0, min(c0 + CHUNK, l1))
            out = part if out is None else (out.astype(np.float32) + part.astype(np.float32)).astype(np.float16)
        return out
    return _sum_levels_fp16(qrows, codes, l0, l1)


def order_desc(max_values):
    """[N, Q] -> [N, Q] sequence numbers, per column by descending value, ties by ascending sequence number."""
    return np.argsort(-max_values.astype(np.float32), axis=0, kind="stable")


def find_examples(sims16, codes, query_codes, top_examples=30, middle_examples=10, bottom_examples=10,
                  layers=(4, 6, 8, 12, 16, 24, 32, 48, 64, 128, 256, 512, 1023)):
    """Generator like server.py:160-325.  sims16 [L, K, K] fp16 (already scaled by layer_norms), codes [N, S, L] int (all
    shards concatenated), query_codes [Q, >= max(layers)] int.  Yields (dict, layer, accumulated [N, S, Q] fp16)."""
    layers = list(layers)
    qrows = query_rows(sims16, query_codes, max(layers))
    bounds = [0] + layers
    acc = None
    for layer, (l0, l1) in zip(layers, zip(bounds[:-1], bounds[1:])):
        part = range_intensities(qrows, codes, l0, l1)
        acc = part if acc is None else (acc.astype(np.float32) + part.astype(np.float32)).astype(np.float16)
        max_values = acc.max(axis=1)                     # [N, Q]
        order = order_desc(max_values)
        n = order.shape[0]
        mid0, mid1 = n // 2 - middle_examples // 2, n // 2 + middle_examples // 2
        sl = {"top": order[:top_examples], "middle": order[mid0:mid1], "bottom": order[n - bottom_examples:]}
        out = {}
        for name, rows in sl.items():
            idx = rows.T                                  # [Q, E]
            inten = np.stack([acc[idx[q], :, q] for q in range(acc.shape[2])])     # [Q, E, S]
            out[name] = {"indices": idx.astype(np.int32), "intensities": inten.astype(np.float16)}
        yield out, layer, acc, max_values
