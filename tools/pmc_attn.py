"""Drives the decode-attention kernel alone (same launch shape as the decode step of config 2, six
layer-sized K/V caches cycled so that nothing is re-read from the 256 MiB Infinity Cache) for a
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass.  Prints the algorithmic bytes per launch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import _lib

B, H, hd, cap, S, nl = 32, 8, 64, 1024, 300, 6
q = torch.randn(B, H * hd, device="cuda")
o = torch.empty_like(q)
Er = torch.rand(cap, hd, device="cuda")
kc = [torch.randn(B, H, cap, hd, device="cuda") for _ in range(nl)]
vc = [torch.randn(B, H, cap, hd, device="cuda") for _ in range(nl)]
kx = [torch.randn(B, H, S, hd, device="cuda") for _ in range(nl)]
vx = [torch.randn(B, H, S, hd, device="cuda") for _ in range(nl)]
st = _lib.stream_ptr()
fold = os.environ.get("PMC_PLAIN", "0") != "1"       # default: the shipped kernels (LayerNorm-folded prologue)
d = H * hd
raw3, u = torch.randn(B, 3 * d, device="cuda"), torch.randn(B, d, device="cuda")
g3, c3 = torch.randn(3 * d, device="cuda"), torch.randn(3 * d, device="cuda")
lnw, lnb, xn = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda"), torch.empty(B, d, device="cuda")
pos = torch.zeros(1, dtype=torch.int32, device="cuda")
alg_self, n_self = 0, 0
for t in range(7, cap, 8):
    pos.fill_(t)
    for l in range(nl):
        if fold:
            _lib.call("amt_attn_decode_fold_fwd", _lib.ptr(raw3), 3 * d, _lib.ptr(kc[l]), _lib.ptr(vc[l]), _lib.ptr(Er), _lib.ptr(u),
                      _lib.ptr(g3), _lib.ptr(c3), _lib.ptr(lnw), _lib.ptr(lnb), _lib.ptr(xn), _lib.ptr(o), B, H, hd, cap, _lib.ptr(pos), 0,
                      cap, 1, 1e-5, 0.125, st)
            _lib.call("amt_attn_decode_fold_fwd", _lib.ptr(raw3), 3 * d, _lib.ptr(kx[l]), _lib.ptr(vx[l]), None, _lib.ptr(u),
                      _lib.ptr(g3), _lib.ptr(c3), _lib.ptr(lnw), _lib.ptr(lnb), _lib.ptr(xn), _lib.ptr(o), B, H, hd, S, None, S,
                      0, 0, 1e-5, 0.125, st)
        else:
            _lib.call("amt_attn_decode_fwd", _lib.ptr(q), _lib.ptr(kc[l]), _lib.ptr(vc[l]), _lib.ptr(Er), _lib.ptr(o), B, H, hd, cap, t, cap, st)
            _lib.call("amt_attn_decode_fwd", _lib.ptr(q), _lib.ptr(kx[l]), _lib.ptr(vx[l]), None, _lib.ptr(o), B, H, hd, S, S - 1, 0, st)
        alg_self += B * (t + 1) * H * hd * 8
        n_self += 1
torch.cuda.synchronize()
print(json.dumps({"variant": "folded" if fold else "plain", "self_launches": n_self, "self_algorithmic_bytes_per_launch": alg_self / n_self,
                  "cross_launches": n_self, "cross_algorithmic_bytes_per_launch": B * S * H * hd * 8}))
