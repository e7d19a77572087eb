import ctypes, os, sys
ROOT="/root/repo"
os.environ["XGGM_LIB"] = os.path.join(ROOT, "x-ggm_amd", "csrc", "build_stamp", "libxggm_hip.so")
sys.path.insert(0, ROOT)
import torch
from xggm_amd import _lib
torch.zeros(1, device="cuda")
out = (ctypes.c_int * 8)()
f = _lib.lib.xggm_gemm_occupancy
f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for tile in (3, 2, 1):
    for lds in (0, 16384, 32768, 40960, 65536, 73728, 81920):
        f(tile, lds, out)
        print(tile, lds, list(out)[:6])
