"""Micro-benchmark of the wide skinny product (the MoE gate|up stack of the V1/V2 lockstep step) through the C ABI: the kernel
alone on pre-packed weights (amt_decode_linear_fwd packs on every call, so pack time is measured separately and subtracted)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import _lib

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

B = 32
st = _lib.stream_ptr()
for (N, K, ln) in ((14336, 512, True), (14336, 512, False), (4096, 1024, True), (7168, 1024, False)):
    x = torch.randn(B, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    lw, lb = torch.randn(K, device="cuda"), torch.randn(K, device="cuda")
    y = torch.empty(B, N, device="cuda"); xn = torch.empty(B, K, device="cuda")
    scratch = torch.empty(N * K, device="cuda")
    full = lambda: _lib.call("amt_decode_linear_fwd", _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(lw) if ln else None, _lib.ptr(lb) if ln else None,
                             None, _lib.ptr(y), _lib.ptr(xn), _lib.ptr(scratch), B, N, K, 0, 1e-5, st)
    pack = lambda: _lib.call("amt_pack_weight_fwd", _lib.ptr(w), _lib.ptr(scratch), N, K, st)
    tf, tp = timeit(full), timeit(pack)
    print(f"N={N} K={K} ln={ln}: pack+gemm {tf:.2f} us, pack {tp:.2f} us, gemm ~{tf - tp:.2f} us ({N * K * 4 / (tf - tp) / 1e6:.2f} TB/s of weights)", flush=True)
