"""``RMSNorm`` on the HIP path (reference ``model/custom_transformer.py:27-48``)."""
import torch
import torch.nn as nn

from .. import _lib


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6, elementwise_affine=True, memory_efficient=False):
        super().__init__()
        self.dim, self.eps, self.elementwise_affine = dim, eps, elementwise_affine
        if elementwise_affine:
            self.weight = nn.Parameter(torch.ones(dim))
        else:
            self.register_parameter("weight", None)

    def forward(self, x):
        if x.device.type != "cuda":
            raise _lib.AmtError("RMSNorm runs on an MI355X only; video2music_amd has no CPU fallback")
        xf = x.to(torch.float32).contiguous()
        y = torch.empty_like(xf)
        w = self.weight.detach().contiguous() if self.weight is not None else None
        _lib.call("amt_rmsnorm_fwd", _lib.ptr(xf), _lib.ptr(w), _lib.ptr(y), xf.numel() // self.dim, self.dim, float(self.eps),
                  _lib.stream_ptr())
        return y.to(x.dtype)

    def extra_repr(self):
        return f"dim={self.dim}, eps={self.eps}, elementwise_affine={self.elementwise_affine}"
