"""Does the BertAdam bandwidth depend on how the five streams (p, g, m, v fp32 + bf16 shadow) are placed relative
to each other?  All carved from ONE slab with a controlled stagger between consecutive buffers."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops
from tools.bench_gemm import timeit


def main():
    n = 190_600_000 // 64 * 64
    slab = torch.empty(5 * (n * 4 + (64 << 20)) // 4, device="cuda", dtype=torch.float32)
    base = slab.data_ptr()
    sqn = torch.ones(1, device="cuda")
    for stagger in (0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 3 << 20, (5 << 20) + 12288, 17 << 20):
        bufs = []
        off = (-base) % (2 << 20)  # start 2 MiB aligned
        for i in range(5):
            bufs.append(slab.view(torch.uint8)[off:off + n * 4].view(torch.float32))
            off += (n * 4 + (2 << 20) - 1) // (2 << 20) * (2 << 20) + stagger
        p, g, m, v, sh32 = bufs
        sh = sh32.view(torch.bfloat16)[:n]
        for t in (p, g, m):
            t.normal_(0, 0.01)
        v.uniform_(0, 1e-4)
        t = timeit(lambda: ops.bertadam(p, g, m, v, sh, sqn, 5.0, 1e-5, None, 0.9, 0.999, 1e-6, 0.01), n=4)
        print("stagger %9d B: %.1f us, %.0f GB/s" % (stagger, t * 1e6, 30.0 * n / t / 1e9), flush=True)


if __name__ == "__main__":
    main()
