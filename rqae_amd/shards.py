"""On-disk format of the code shards (first half).

The reference writes one ``.pt`` per 1024 sequences: an int32 tensor ``(sequences, seq_len, num_quantizers)`` -- 512 MiB
per shard at depth 1024 (scripts/1_create_activations.py:184-186, 258-261) -- and reads it back with ``torch.load``
(scripts/3_make_rqae_features.py:44-51).  Codes are codebook row numbers, so for K <= 65536 rows they fit 16 bits:
``save_codes(..., dtype="uint16")`` writes the same tensor at half the bytes, and ``decode`` / ``RQAEFeature.intensity``
take uint16, int32 and int64 index tensors alike (include/rqae_hip.h), so the narrow shard is consumed as it is, without
widening on the host.  ``dtype="int32"`` writes exactly what the reference writes.
"""
import torch


def save_codes(path: str, codes: torch.Tensor, dtype: str = "int32", codebook_rows: int = None) -> None:
    """codes: integer tensor (..., num_quantizers) of codebook row numbers."""
    codes = codes.detach().cpu()
    if dtype == "int32":
        out = codes.to(torch.int32)
    elif dtype == "uint16":
        hi = int(codes.max()) if codes.numel() else 0
        lo = int(codes.min()) if codes.numel() else 0
        if lo < 0 or hi > 65535 or (codebook_rows is not None and codebook_rows > 65536):
            raise ValueError(f"codes in {lo}..{hi} do not fit uint16")
        out = codes.to(torch.int32).to(torch.uint16)
    else:
        raise ValueError(f"unknown shard dtype {dtype!r} (int32 | uint16)")
    torch.save(out.contiguous(), path)


def load_codes(path: str) -> torch.Tensor:
    """The shard as stored (int32 as the reference's, or uint16); no code is executed from the file."""
    t = torch.load(path, weights_only=True)
    if not isinstance(t, torch.Tensor) or t.dtype not in (torch.int32, torch.uint16, torch.int64):
        raise ValueError(f"{path}: not a code shard")
    return t
