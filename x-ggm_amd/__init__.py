"""xggm_amd: MI355X-native (gfx950) implementation of the X-GGM training-step hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd tape,
torch.distributed/RCCL); every op on the path is a hand-written HIP kernel behind the
C-ABI declared in ``include/xggm.h`` (``x-ggm_amd/csrc``).  There is no CPU fallback:
``xggm_amd._lib`` raises if the extension is missing.
"""
__version__ = "0.1.0"
