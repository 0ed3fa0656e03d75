"""Micro-benchmark of the decode-step kernels through the C ABI (not a test; run on the GPU box)."""
import os, sys, time
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
for (N, K, ln) in ((1536, 512, True), (512, 512, False), (512, 512, True), (1024, 512, True), (512, 1024, False)):
    x = torch.randn(B, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    lw, lb = torch.randn(K, device="cuda"), torch.randn(K, device="cuda")
    r = torch.randn(B, N, device="cuda"); y = torch.empty(B, N, device="cuda"); xn = torch.empty(B, K, device="cuda")
    scratch = torch.empty(N * K, device="cuda")
    st = _lib.stream_ptr()
    lib = _lib.load()
    # pack once, then launch the kernel only: call decode_linear once and reuse scratch via a raw launcher is not exposed,
    # so time pack+gemm and pack alone
    def full():
        _lib.call("amt_decode_linear_fwd", _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(lw) if ln else None, _lib.ptr(lb) if ln else None,
                  _lib.ptr(r), _lib.ptr(y), _lib.ptr(xn), _lib.ptr(scratch), B, N, K, 0, 1e-5, st)
    res = {}
    os.environ["AMT_DBG"] = "0"
    full()
    for dbg in (0,):
        os.environ["AMT_DBG"] = str(dbg | 16)
        res[dbg] = timeit(full)
    print(f"N={N} K={K} ln={ln}: " + "  ".join(f"dbg{d}={t:.2f}us" for d, t in res.items()), flush=True)
