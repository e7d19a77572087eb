"""Validation-sweep throughput (``predict``: eval forward -> logit_fc -> arg-max, src/vqa/vqacpv2.py:315-339) of the
full 9/5/5 model at the reference's evaluation batch size (512, script/vqacpv2.sh), inputs resident in HBM:
captured replay against eager launches.  Forward work: 10.6 GFLOP per sample.
usage: python tools/bench_predict.py [batch=512] [iters=20]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    from xggm_amd import param, synth
    from xggm_amd.engine import CapturedPredictor
    from xggm_amd.lxrt.modeling import BertConfig, VISUAL_CONFIG
    from xggm_amd.vqa.vqacpv2_model import VQAModel
    VISUAL_CONFIG.set_visual_dims(2048, 4)
    a = param.parse_args(["--llayers", "9", "--xlayers", "5", "--rlayers", "5"])
    torch.manual_seed(0)
    model = VQAModel(2274, args=a, config=BertConfig(30522), compute_dtype=torch.bfloat16).to("cuda")
    b = synth.vqa_batch(B, A=2274, seed=5)
    feats, boxes = torch.from_numpy(b["feats"]).cuda(), torch.from_numpy(b["boxes"]).cuda()
    sent = tuple(torch.from_numpy(b[k]).cuda() for k in ("input_ids", "input_mask", "segment_ids"))
    for use_graph in (True, False):
        pred = CapturedPredictor(model, B, use_graph=use_graph)
        for _ in range(3):
            lab, _ = pred(feats, boxes, sent)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            lab, _ = pred(feats, boxes, sent)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print("%s batch %d: %.2f ms per batch, %.0f samples/s, %.0f TFLOP/s forward (labels %s...)" % (
            "captured" if use_graph else "eager   ", B, dt * 1e3, B / dt, 10.6e9 * B / dt / 1e12, lab[:4].tolist()),
            flush=True)


if __name__ == "__main__":
    main()
