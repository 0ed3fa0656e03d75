import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import _lib
def run(M, N, K, n=20):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda")
    st = _lib.stream_ptr()
    f = lambda: _lib.call("amt_linear_fwd", _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(y), M, N, K, 0, st)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K}: {dt*1e6:.1f} us  {2*M*N*K/dt/1e12:.1f} TFLOP/s", flush=True)
for s in ((4096, 4096, 4096), (8192, 8192, 1024), (32768, 512, 512), (32768, 1536, 512), (32768, 1024, 512), (32768, 512, 1024), (9600, 512, 1312), (9600, 1536, 512), (32768, 159, 512)):
    run(*s)
