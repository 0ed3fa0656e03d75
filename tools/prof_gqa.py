"""rocprofv3 --kernel-trace target: MultiheadGQA(512, 8, 2), L = 2048, causal, B = $NB (config 4 of BASELINE.json)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video2music_amd.model.grouped_query_attention import MultiheadGQA
B = int(os.environ.get("NB", "1"))
g = MultiheadGQA(512, 8, 2).cuda().eval()
x = torch.randn(2048, B, 512, device="cuda")
for _ in range(3): g(x, x, x, is_causal=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g(x, x, x, is_causal=True)
torch.cuda.synchronize(); print("ms per call", (time.perf_counter() - t0) / 20 * 1e3)
