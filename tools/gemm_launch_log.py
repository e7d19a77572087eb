"""Which GEMM launches does a training pass issue, and how many problems share each?  Wraps the C-ABI calls of one eager
iteration and prints every launch (entry point, problems, M x N x K of each).   python tools/gemm_launch_log.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = bench.parse(["--no-graph", "--no-cpu-baseline", "--no-kernel-timing", "--no-loader"] + sys.argv[1:])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model, optim, batch = bench.build(args, dev)
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd import ops, _lib
    tr = CapturedTrainer(model, optim, batch, sigma=1.0, order=args.order, use_graph=False)
    tr.iteration("rel")
    torch.cuda.synchronize()
    log = []
    real = _lib.call

    def call(name, *a):
        if name.startswith("xggm_gemm"):
            if "grouped" in name:
                arr, n = a[0], a[1]
                probs = ops._ct.cast(arr, ops._ct.POINTER(ops.GemmProblem))
                log.append((name, tuple((probs[i].M, probs[i].N, probs[i].K, probs[i].batch) for i in range(n))))
            else:
                log.append((name, ((a[3], a[4], a[5], a[11] if len(a) > 11 else 1),)))
        return real(name, *a)

    _lib.call = call
    ops.call = call
    for kind in ("plain", "rel"):
        log.append(("---- pass %s" % kind, ()))
        tr.run_pass(kind)
    torch.cuda.synchronize()
    cnt = collections.Counter()
    for name, shapes in log:
        if name.startswith("----"):
            for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
                print("   %3d x %s %s" % (v, k[0], " | ".join("%dx%dx%d%s" % (m, n, kk, "" if b == 1 else " b%d" % b) for m, n, kk, b in k[1])))
            cnt = collections.Counter()
            print(name)
            continue
        cnt[(name, shapes)] += 1
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
        print("   %3d x %s %s" % (v, k[0], " | ".join("%dx%dx%d%s" % (m, n, kk, "" if b == 1 else " b%d" % b) for m, n, kk, b in k[1])))


if __name__ == "__main__":
    main()
