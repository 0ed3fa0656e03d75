"""Device selection, mirroring the reference's ``utilities/device.py:18-43`` API.

The reference hard-codes ``cuda:0`` (``utilities/device.py:8-9``).  Here the default device is the
rank-local GPU (``LOCAL_RANK`` under torchrun), so one process per GPU works unchanged.
"""
import os

import torch

TORCH_CPU_DEVICE = torch.device("cpu")
USE_CUDA = True


def _cuda_device():
    if torch.cuda.device_count() > 0:
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    return None


def use_cuda(cuda_bool):
    """Sets whether to use the GPU (if available)."""
    global USE_CUDA
    USE_CUDA = cuda_bool


def get_device():
    """Default device: the rank-local GPU unless ``use_cuda(False)`` or no GPU is present."""
    dev = _cuda_device()
    if (not USE_CUDA) or dev is None:
        return TORCH_CPU_DEVICE
    return dev


def cuda_device():
    return _cuda_device()


def cpu_device():
    return TORCH_CPU_DEVICE
