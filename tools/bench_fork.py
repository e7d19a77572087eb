"""Does a hipGraph run two independent small kernels of forked streams concurrently?  LN forward of the
language rows [640, 768] and the vision rows [1152, 768]: serial on one stream vs forked on two."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xggm_amd import ops

dev = "cuda"
BF = torch.bfloat16
NREP = 40


def main():
    H = 768
    xs = [torch.randn(r, H, device=dev).to(BF) for r in (640, 1152)]
    res = [torch.randn_like(x) for x in xs]
    gam, bet, bias = (torch.randn(H, device=dev) for _ in range(3))
    outs = [torch.empty_like(x) for x in xs]

    def ln(i):
        ops.ln_fwd(xs[i], bias, res[i], gam, bet, 1e-12, out=outs[i], save=False)

    def serial():
        for _ in range(NREP):
            ln(0)
            ln(1)

    side = torch.cuda.Stream()

    def forked():
        main_s = torch.cuda.current_stream()
        for _ in range(NREP):
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                ln(1)
            ln(0)
            main_s.wait_stream(side)

    def only0():
        for _ in range(NREP):
            ln(0)

    for name, fn in (("lang only", only0), ("serial lang+visn", serial), ("forked lang|visn", forked)):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print("%-20s %.2f us per step (one step = LN of both streams)" % (name, e0.elapsed_time(e1) * 1e3 / (5 * NREP)), flush=True)


if __name__ == "__main__":
    main()
