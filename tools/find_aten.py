"""Which framework (ATen) kernels still run inside a training pass, and from where?  One eager iteration of the bench
model under torch.profiler with Python stacks; prints every aten op that launched a device kernel, with its stack.
    python tools/find_aten.py"""
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
    tr = CapturedTrainer(model, optim, batch, sigma=1.0, order=args.order, use_graph=False)
    tr.iteration("rel")
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        tr.iteration("rel")
        tr.iteration("node")
        torch.cuda.synchronize()
    seen = {}
    for ev in prof.events():
        if not ev.name.startswith("aten::"):
            continue
        dev_us = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        hit = any(k.name and ("at::native" in k.name or "elementwise" in k.name or "emcpy" in k.name or "copyBuffer" in k.name
                              or "fillBuffer" in k.name) for k in ev.kernels)
        # device-to-device copies are memcpy activities, not kernels: an aten::copy_ with device time but no kernel of the
        # package is one (clone / contiguous / copy_ of a tensor that was not contiguous)
        if not hit and not (ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_")
                            and dev_us > 0):
            continue
        st = [s for s in (ev.stack or []) if "ggm" in s or "bench.py" in s][:4]
        if not st:
            st = list(ev.stack or [])[:3]
        key = (ev.name, str(ev.input_shapes), tuple(st))
        seen[key] = seen.get(key, 0) + 1
    for (name, shp, st), n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print("%3d x %-14s %-44s %s" % (n, name, shp[:44], " <- ".join(s.split("/")[-1] for s in st) or "(no python frame: autograd engine)"))


if __name__ == "__main__":
    main()
