"""Golden fixture of MultiheadGQA(RoPE=...) (model/grouped_query_attention.py:216,316-322) from the REFERENCE classes on CPU:
the rotary embedding applied through the raw (heads, len, B, head_dim) view of the projection buffers, with a cache built for
dim = head_dim (one slab, broadcast) and for dim = embed_dim (the view folds the extra frequencies into the leading axis and
truncates it to the tensor's head count, rotate_operation.py:148-149), B in {1, 2, 3}, causal and not.

TEST INFRASTRUCTURE; runs only in the build container:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_gqa_rope.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG                                   # noqa: E402

CASES = [("hd", 32), ("full", 256)]                        # name -> dim the RotaryPositionalEmbeddings is built with


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    ref = MG.import_reference()
    t = MG.t
    out = {}
    rs = np.random.RandomState(77)
    for name, dim in CASES:
        rope = ref.ro.RotaryPositionalEmbeddings(dim=dim, max_seq_len=80)
        m = ref.gqa.MultiheadGQA(256, 8, 2, RoPE=rope).eval()          # head_dim 32, 4 query heads per kv head
        MG.load_synthetic(m, seed=3)
        for L, B in ((6, 1), (6, 2), (64, 1), (64, 3)):
            x = rs.standard_normal((L, B, 256)).astype(np.float32)
            out[f"{name}_x_L{L}_B{B}"] = x
            for causal in (False, True):
                y, _ = m(t(x), t(x), t(x), is_causal=causal)
                out[f"{name}_y_L{L}_B{B}_c{int(causal)}"] = y.numpy()
        # cross form: keys / values of another length than the queries (two rope views of different seq)
        xq = rs.standard_normal((10, 2, 256)).astype(np.float32)
        xk = rs.standard_normal((37, 2, 256)).astype(np.float32)
        y, _ = m(t(xq), t(xk), t(xk), is_causal=False)
        out[f"{name}_xq"], out[f"{name}_xk"], out[f"{name}_y_cross"] = xq, xk, y.numpy()
    np.savez_compressed(os.path.join(MG.OUT, "g_gqa_rope.npz"), **out)
    print("wrote g_gqa_rope.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
