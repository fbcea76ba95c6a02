"""Per-phase timeline of sample workgroups of the encode kernel.

    python tools/timeline.py [codes]        (on a GPU box; prints what profiles/r01_timeline_fwd.txt holds)

Builds rqae_amd/csrc/rqae_hip.hip with -DRQAE_TIMELINE (one translation unit, only the d = 2304 / c = 5 kernels)
into tools/_build/librqae_hip_tl.so: every wave of four sample workgroups stamps s_memtime at its phase boundaries.
"""
import os, sys, ctypes, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rqae_amd._native as nat
import subprocess
TL_SO = os.path.join(ROOT, "tools", "_build", "librqae_hip_tl.so")
SRC = os.path.join(ROOT, "rqae_amd", "csrc", "rqae_hip.hip")
if not os.path.exists(TL_SO) or any(os.path.getmtime(TL_SO) < os.path.getmtime(os.path.join(os.path.dirname(SRC), f))
                                    for f in ("rqae_hip.hip", "encode_lds.hpp", "rqae_common.hpp", "decode.hpp", "hook.hpp", "intensity.hpp", "tables.hpp")):
    os.makedirs(os.path.dirname(TL_SO), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
                    "-fhip-fp32-correctly-rounded-divide-sqrt", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                    "-DRQAE_SINGLE_TU", "-DRQAE_TIMELINE", "-DRQAE_ABLATE_BUILD", "-DRQAE_EXPERIMENT_ONLY_95", SRC, "-o", TL_SO],     # one translation unit
                   check=True, cwd=os.path.dirname(SRC))
nat.SO_PATH = TL_SO
from rqae_amd import RQAE, synth
import bench
cfg = bench.CFG
w = synth.weights(1234, cfg["dim"], cfg["codebook_dim"], cfg["num_quantizers"], cfg["codebook_size"])
m = RQAE(**cfg).eval(); m.load_state_dict(synth.state_dict_from_weights(w)); m = m.to("cuda")
n = 262144
x = bench.synth_batch(n, torch.device("cuda"), 1000).view(1, n, -1)
codes_only = len(sys.argv) > 1 and sys.argv[1] == "codes"
fn = m.encode if codes_only else m.forward
with torch.inference_mode():
    for _ in range(3): fn(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(x); b.record(); torch.cuda.synchronize()
print("kernel ms", a.elapsed_time(b))
L = nat._load() if hasattr(nat, "_load") else nat.lib()
buf = np.zeros((4, 8, 64), dtype=np.int64)
rc = L.rqae_hip_debug_timeline(ctypes.c_void_p(buf.ctypes.data))
assert rc == 0
names = ["entry", "x+rec0 landed", "slot_end", "pass0", "barA", "q:unit vectors", "q:scan", "q:winners", "q:end", "slot_end"]
for l in range(1, 8):
    names += [f"pass{l}", "barA", "q:unit vectors", "q:scan", "q:winners", "q:end", "slot_end"]
names += ["last pass", "stores drained"]
for blk in range(4):
    t = buf[blk]
    t0 = t[:, 0].min()
    print(f"--- sample workgroup {blk}: wave rows, cycles since the first wave's entry (delta to previous stamp)")
    for k, nm in enumerate(names):
        if t[0, k] == 0: break
        row = " ".join(f"{int(t[wv, k] - t0):7d}" for wv in range(8))
        d = " ".join(f"{int(t[wv, k] - t[wv, k - 1]) if k else 0:6d}" for wv in (0, 4))
        print(f"{k:2d} {nm:16s} {row}   d(w0,w4)= {d}")
# aggregate: mean over waves and blocks of each delta
d = np.diff(buf[:, :, :len(names)].astype(np.float64), axis=2)
agg = {}
for k, nm in enumerate(names[1:]):
    key = nm if not nm.startswith("pass") else ("pass(mid)" if nm not in ("pass0",) else "pass0")
    agg.setdefault(key, []).append(d[:, :, k].mean())
tot = (buf[:, :, len(names) - 1] - buf[:, :, 0]).mean()
print("mean cycles per phase (sum over levels), share of the workgroup's", int(tot), "cycles:")
for k, v in agg.items():
    print(f"  {k:16s} {sum(v):9.0f}  {100 * sum(v) / tot:5.1f} %   ({len(v)} x {np.mean(v):.0f})")
