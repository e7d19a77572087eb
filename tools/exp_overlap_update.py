"""How much of the optimiser step (HBM-bound, ~1.2 ms per pass) hides under the NEXT pass's forward + backward when
the two run on two streams?  A timing experiment only: no dependencies between the two graphs, so the trained values are
meaningless here -- it measures what the GPU can overlap, before any engine work.
   python tools/exp_overlap_update.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = bench.parse(["--steps", "30", "--warmup", "5"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from xggm_amd.engine import CapturedTrainer, _quiet_gc
    model, optim, batch = bench.build(args, dev)
    tr = CapturedTrainer(model, optim, batch, sigma=1.0, order="vqa", use_graph=False)
    for _ in range(2):
        for kind in ("plain", "rel"):
            tr._eager_pass(kind)
    torch.cuda.synchronize()
    graphs = {}
    with _quiet_gc():
        for kind in ("plain", "rel"):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                tr._fwd_bwd(kind)
            graphs[kind] = g
        gu = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gu):
            tr._update()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def timed(step, n=40):
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return 1000 * (time.perf_counter() - t0) / n

    def serial():
        for kind in ("plain", "rel"):
            graphs[kind].replay()
            gu.replay()

    def overlapped():
        # update of pass k on the side stream, forward + backward of pass k + 1 on the main stream
        for kind in ("plain", "rel"):
            ev = torch.cuda.Event()
            ev.record(main_s)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                gu.replay()
            graphs[kind].replay()
            main_s.wait_stream(side)

    def only_fb():
        for kind in ("plain", "rel"):
            graphs[kind].replay()

    def only_u():
        gu.replay()
        gu.replay()

    print("forward + backward of both passes          %.3f ms" % timed(only_fb), flush=True)
    print("two updates                                %.3f ms" % timed(only_u), flush=True)
    print("serial (the iteration as it is)            %.3f ms" % timed(serial), flush=True)
    print("update of pass k beside pass k + 1         %.3f ms" % timed(overlapped), flush=True)


if __name__ == "__main__":
    main()
