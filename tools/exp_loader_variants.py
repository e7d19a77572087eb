"""Where does the loader leg's overhead come from?  The bench's timed loop under variants of the input boundary:
  resident   no loader at all (the `value` configuration)
  current    DataLoaderX(device): producer thread, H2D on its copy stream, events, 7-field device-to-device hand-over
  nohandover the same loader running, but the trainer keeps its resident batch (no hand-over copies)
  hostonly   DataLoaderX(device=None): the producer only fills host buffers (no H2D, no events); nothing handed over
  inline     producer fills PINNED host buffers; the consumer copies them H2D into the trainer's static buffers on the
             compute stream (one copy per field, stream-ordered: no ring, no events, no device-to-device copies)"""
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
    from xggm_amd.engine import CapturedTrainer
    import random
    model, optim, batch = bench.build(args, dev)
    tr = CapturedTrainer(model, optim, batch, sigma=1.0, order="vqa")
    rng = random.Random(0)

    def br():
        return "rel" if rng.randint(1, 10) <= 5 else "node"

    def timed(step, n=30):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return 1000 * (time.perf_counter() - t0) / n

    res = {}
    res["resident"] = timed(lambda: tr.iteration(br()))
    loader, _, tmp = bench.make_loader(model, args, dev)
    it = iter(loader)
    res["current"] = timed(lambda: (tr.load_batch(bench.batch_of(next(it))), tr.iteration(br())))
    res["nohandover"] = timed(lambda: (next(it), tr.iteration(br())))
    it.close()
    # host-only producer
    from xggm_amd.tools.data_loader import DataLoaderX
    hl = DataLoaderX(loader.ds, args.batch, shuffle=True, drop_last=True, device=None, batcher=model.lxrt_encoder.batcher,
                     depth=3, seed=1, epochs=None)
    it = iter(hl)
    res["hostonly"] = timed(lambda: (next(it), tr.iteration(br())))
    # inline: pin the host-only loader's buffers and copy them from the consumer on the compute stream
    for slot in hl.host:
        for k in list(slot):
            slot[k] = slot[k].pin_memory()

    def inline_step():
        qid, feats, boxes, sent, target, adj = next(it)
        s = tr.static
        s["feats"].copy_(feats, non_blocking=True)
        s["boxes"].copy_(boxes, non_blocking=True)
        s["target"].copy_(target, non_blocking=True)
        s["adj_true"].copy_(adj, non_blocking=True)
        s["input_ids"].copy_(sent[0], non_blocking=True)
        s["input_mask"].copy_(sent[1], non_blocking=True)
        s["segment_ids"].copy_(sent[2], non_blocking=True)
        tr.iteration(br())

    res["inline (pageable ids)"] = timed(inline_step)
    it.close()
    for k, v in res.items():
        print("%-24s %.3f ms/step  (+%.3f)" % (k, v, v - res["resident"]), flush=True)


if __name__ == "__main__":
    main()
