"""How much idle GPU is there inside an iteration?  Two independent trainers (two models, two sets of graphs) replayed on two
streams at the same time against one after the other.  If two concurrent iterations take much less than twice one, the
latency-bound kernels of one chain can hide under the other's -- the potential of running the two modality streams (or two
micro-batches) as parallel branches.  A timing experiment only.
   python tools/exp_two_streams.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    args = bench.parse(["--steps", "30", "--warmup", "5", "--batch", str(batch)])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from xggm_amd.engine import CapturedTrainer
    trs = []
    for i in range(2):
        model, optim, b = bench.build(args, dev)
        trs.append(CapturedTrainer(model, optim, b, sigma=1.0, order="vqa"))
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()

    def timed(step, n=30):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return 1000 * (time.perf_counter() - t0) / n

    def one():
        with torch.cuda.stream(s[0]):
            trs[0].iteration("rel")

    def serial():
        with torch.cuda.stream(s[0]):
            trs[0].iteration("rel")
            trs[1].iteration("rel")

    def both():
        for i in range(2):
            with torch.cuda.stream(s[i]):
                trs[i].iteration("rel")

    t1 = timed(one)
    print("batch %d: one iteration                      %.3f ms" % (batch, t1), flush=True)
    print("batch %d: two iterations, one stream         %.3f ms" % (batch, timed(serial)), flush=True)
    t2 = timed(both)
    print("batch %d: two iterations, two streams        %.3f ms  (%.2f x one)" % (batch, t2, t2 / t1), flush=True)


if __name__ == "__main__":
    main()
