"""Which activations of the fp8 step still reach an e4m3 product WITHOUT a producer-written copy (Fp8State.get falls back
to the stand-alone quantiser)?  Prints every fallback of one iteration with its site key and shape.
    python tools/exp_fp8_quantize_sites.py   (on an MI355X)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = bench.parse(["--dtype", "fp8", "--no-graph", "--no-cpu-baseline", "--no-kernel-timing", "--no-loader"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model, optim, batch = bench.build(args, dev)
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd import ops, fp8 as F
    tr = CapturedTrainer(model, optim, batch, sigma=1.0, order=args.order, use_graph=False)
    for _ in range(2):
        tr.iteration("rel")
    torch.cuda.synchronize()
    real = ops.quantize_fp8
    log = []

    def spy(x, qscale=None, amax=None):
        import traceback
        st = [f for f in traceback.extract_stack()[:-1] if "x-ggm_amd" in f.filename or "xggm_amd" in f.filename]
        log.append((tuple(x.shape), " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-4:])))
        return real(x, qscale=qscale, amax=amax)

    ops.quantize_fp8 = spy
    F.ops.quantize_fp8 = spy
    for kind in ("plain", "rel"):
        log.append(("---- pass " + kind, ""))
        tr.run_pass(kind)
    torch.cuda.synchronize()
    for shape, where in log:
        print(shape, where)


if __name__ == "__main__":
    main()
