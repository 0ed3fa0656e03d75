"""Crossover between the 128x128-tile GEMM and the skinny GEMM as a function of the row count (run on the GPU box;
AMT_GEMM_SMALL_M selects: 0 = always the tiled kernel, a large value = the skinny one up to 4096 rows)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd import ops
res = {}
for N, K in ((512, 512), (1536, 512), (512, 1024)):
    w, b = torch.randn(N, K, device="cuda") * K ** -0.5, torch.randn(N, device="cuda")
    for M in (64, 128, 256, 384, 512, 768, 1024, 2048, 4096):
        x = torch.randn(M, K, device="cuda")
        for _ in range(5): ops.linear(x, w, b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): ops.linear(x, w, b)
        torch.cuda.synchronize()
        res[f"N{N}_K{K}_M{M}"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
print(json.dumps(res))
