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
alg_self, n_self = 0, 0
for t in range(7, cap, 8):
    for l in range(nl):
        _lib.call("amt_attn_decode_fwd", _lib.ptr(q), _lib.ptr(kc[l]), _lib.ptr(vc[l]), _lib.ptr(Er), _lib.ptr(o), B, H, hd, cap, t, cap, st)
        alg_self += B * (t + 1) * H * hd * 8
        n_self += 1
        _lib.call("amt_attn_decode_fwd", _lib.ptr(q), _lib.ptr(kx[l]), _lib.ptr(vx[l]), None, _lib.ptr(o), B, H, hd, S, S - 1, 0, st)
torch.cuda.synchronize()
print(json.dumps({"self_launches": n_self, "self_algorithmic_bytes_per_launch": alg_self / n_self,
                  "cross_launches": n_self, "cross_algorithmic_bytes_per_launch": B * S * H * hd * 8}))
