"""Times the relative-position causal self-attention prefill kernel (attn_prefill_kernel<64,true>) and the cross-attention one
at config 2's shapes through the operator entry points (HIP events on the launch stream)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import _lib

B, H, L, hd, S = 32, 8, int(os.environ.get("L", 1024)), 64, 300
d = H * hd
q = torch.randn(B, L, d, device="cuda") * 0.1
k = torch.randn(B, L, d, device="cuda")
v = torch.randn(B, L, d, device="cuda")
Er = torch.rand(L, hd, device="cuda")
o = torch.empty(B, L, d, device="cuda")
kx, vx = torch.randn(B, S, d, device="cuda"), torch.randn(B, S, d, device="cuda")
st = _lib.stream_ptr()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


rpr = timeit(lambda: _lib.call("amt_rpr_attn_fwd", _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(Er), _lib.ptr(o), B, H, L, hd, L, st))
cross = timeit(lambda: _lib.call("amt_cross_attn_fwd", _lib.ptr(q), _lib.ptr(kx), _lib.ptr(vx), _lib.ptr(o), B, H, L, S, hd, 0, st))
causal = timeit(lambda: _lib.call("amt_cross_attn_fwd", _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), B, H, L, L, hd, 1, st))
useful = 4 * B * L * L * d / 2            # QK^T + PV on the causal half
with_er = 6 * B * L * L * d / 2           # + Q.Er^T (the reference's einsum over every distance, model/rpr.py:392-393)
print(json.dumps({"opt": os.environ.get("AMT_PREFILL_OPT", "default"),
                  "rpr_causal_us": round(rpr, 1), "rpr_TFLOPs_QK_PV": round(useful / rpr / 1e6, 1), "rpr_TFLOPs_incl_QEr": round(with_er / rpr / 1e6, 1),
                  "plain_causal_us": round(causal, 1), "plain_causal_TFLOPs": round(useful / causal / 1e6, 1),
                  "cross_us": round(cross, 1), "cross_TFLOPs": round(4 * B * L * S * d / cross / 1e6, 1)}))
