#!/usr/bin/env python3
"""Benchmark of the X-GGM training iteration on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one training iteration of the reference loop (src/vqa/vqacpv2.py:164-254): a
plain-VQA pass and a graph-generative pass, EACH a full forward + backward + grad-norm clip +
BertAdam update of the 220.8 M-parameter LXMERT(9/5/5) + GCNx2 generator + heads model, on a
synthetic VQA-CP-v2-shaped batch of 32 samples per GPU (36 objects x 2048-d, 20 tokens,
A = 2274), bf16 storage / fp32 accumulate, dropout on.  Weak scaling: 32 samples per rank,
gradients averaged over RCCL.  Prints ONE JSON line (rank 0).

Extra objects on the line:
  roofline     the dominant kernel family of the step, timed live with HIP events on the
               launch stream in an instrumented (un-captured) iteration
  cpu_baseline the CPU oracle (oracle/xggm_oracle.py, torch fp32) timed on this box's host cores
               on one iteration of the same workload (N = 1 only)
"""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=32, help="samples per GPU")
    p.add_argument("--answers", type=int, default=2274)
    p.add_argument("--delta", type=int, default=5, help="relation-branch probability delta/10")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-timing", action="store_true")
    p.add_argument("--wire", default="bf16", choices=["bf16", "f32"], help="gradient all-reduce dtype")
    p.add_argument("--seed", type=int, default=9595)
    return p.parse_args()


def build(args, device):
    from xggm_amd import param, synth
    from xggm_amd.lxrt.modeling import BertConfig, VISUAL_CONFIG
    from xggm_amd.vqa.vqacpv2_model import VQAModel
    from xggm_amd.vqa.vqacpv2 import make_optimizer
    VISUAL_CONFIG.set_visual_dims(2048, 4)
    a = param.parse_args(["--llayers", "9", "--xlayers", "5", "--rlayers", "5"])
    torch.manual_seed(args.seed)
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = VQAModel(args.answers, gnn="GCN", n_layers=2, args=a, config=BertConfig(30522), compute_dtype=dt)
    model.seed = args.seed
    model = model.to(device)
    rank = int(os.environ.get("RANK", 0))
    b = synth.vqa_batch(args.batch, A=args.answers, seed=1000 + rank)
    batch = {k: torch.from_numpy(v).to(device) for k, v in b.items() if k != "randn_adj"}
    n_iters = args.steps + args.warmup + 16
    optim = make_optimizer(model, 1e-6, 2 * n_iters)  # lr of script/vqacpv2.sh:26, t_total = 2 * iterations
    return model, optim, batch


def kernel_timing(trainer, branches):
    """time every C-ABI launch of one un-captured iteration per branch with HIP events recorded
    on the launch stream; a long sleep kernel is queued first so the host runs ahead and the
    events bracket back-to-back GPU execution, not Python latency."""
    from xggm_amd import _lib
    import xggm_amd.ops as ops_mod
    rec = []
    orig = _lib.call

    def timed(name, *a):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        work = 0.0
        if name.startswith("xggm_gemm_grouped_"):
            probs = (ops_mod.GemmProblem * a[1]).from_address(a[0].value)
            work = sum(2.0 * p.M * p.N * p.K * p.batch for p in probs)
            name = "xggm_gemm_" + name.rsplit("_", 1)[1]  # same kernel family as the single launches
        elif name.startswith("xggm_gemm_"):
            work = 2.0 * a[3] * a[4] * a[5] * a[11]
        e0.record()
        orig(name if not name.startswith("xggm_gemm_") or len(a) > 3 else "xggm_gemm_grouped_" + name.rsplit("_", 1)[1], *a)
        e1.record()
        desc = ""
        if os.environ.get("XGGM_DUMP_GEMMS") and name.startswith("xggm_gemm_"):
            if len(a) == 3:
                desc = " + ".join("%dx%dx%d%s" % (p.M, p.N, p.K, "f" if p.c_f32 else "") for p in probs)
            else:
                desc = "%dx%dx%d" % (a[3], a[4], a[5])
        rec.append((name, a, e0, e1, work, desc))

    for kind in ["plain"] + list(branches):
        torch.cuda.synchronize()
        torch.cuda._sleep(int(5e7))  # head start for the host: launches queue up behind it
        ops_mod.call = timed
        try:
            trainer._eager_pass(kind)
        finally:
            ops_mod.call = orig
        torch.cuda.synchronize()
    fam = {}
    if os.environ.get("XGGM_DUMP_GEMMS"):
        agg = {}
        for name, a, e0, e1, work, desc in rec:
            if desc:
                t = agg.setdefault(desc, [0.0, 0, 0.0])
                t[0] += e0.elapsed_time(e1); t[1] += 1; t[2] += work
        for desc, (ms, n, w) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
            log("gemm %-60s n=%3d total %.3f ms avg %.1f us %.0f TF" % (desc, n, ms, 1000 * ms / n, w / ms / 1e9))
    for name, a, e0, e1, work, desc in rec:
        ms = e0.elapsed_time(e1)
        f = fam.setdefault(name, dict(ms=0.0, n=0, flops=0.0, bytes=0.0))
        f["ms"] += ms
        f["n"] += 1
        if name.startswith("xggm_gemm_"):
            f["flops"] += work
        elif name == "xggm_bertadam_f32":
            f["bytes"] += a[5] * (16 + 12 + (2 if a[4] else 0))
        elif name == "xggm_sqnorm_f32":
            f["bytes"] += a[1] * 4
    return fam


def cpu_baseline(args):
    """the CPU oracle on ONE iteration (plain + relation pass, each fwd+bwd+clip+BertAdam) of the
    same workload; weights are random (values do not affect the timing)."""
    from oracle import shapes, xggm_oracle as O
    from xggm_amd import synth
    # the GPU box gives one GPU a 16-core CPU share; more threads than that only thrash
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    cfg = shapes.FULL
    g = torch.Generator().manual_seed(0)
    P = {}
    for k, s in shapes.model_shapes(cfg, args.answers).items():
        t = torch.randn(s, generator=g) * 0.02
        if len(s) == 1 and k.endswith("weight"):
            t = t + 1.0
        P[k] = t
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    step = {k: 0 for k in P}
    b = synth.vqa_batch(args.batch, A=args.answers, seed=1000)
    b["randn_node"] = synth.randn_nodes(args.batch, 36, 768, 0)
    b = {k: torch.from_numpy(v) for k, v in b.items()}
    t0 = time.perf_counter()
    iters = 0
    while iters < 8 and time.perf_counter() - t0 < 12.0:  # bounded sample: about 10-20 s of CPU work
        O.train_pass(P, M, V, step, b, cfg, "plain", 1e-6, 100)
        O.train_pass(P, M, V, step, b, cfg, "rel", 1e-6, 100, sigma=1.0, kl_weight=8.0, gnn="GCN")
        iters += 1
    dt = time.perf_counter() - t0
    return {"value": round(iters * args.batch / dt, 3), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d iterations (plain pass + relation-generation pass, each fwd+bwd+clip+BertAdam) at %d "
                      "samples, fp32, full 9/5/5 model, torch CPU oracle, %.1f s" % (iters, args.batch, dt)}


def pmc_traffic(family):
    """HBM-side bytes per C-ABI launch of a kernel family, from the committed PMC passes
    (profiles/r01_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this bench in separate eager
    runs, gfx950 correction applied by tools/pmc_summary.py); None when no profile covers the family."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    key = {"xggm_gemm_bf16": "gemm_", "xggm_bertadam_f32": "bertadam_kernel", "xggm_ln_bwd_bf16": "ln_bwd_kernel",
           "xggm_ln_fwd_bf16": "ln_fwd_kernel", "xggm_attn_bwd_bf16": "attn_bwd", "xggm_attn_fwd_bf16": "attn_fwd"}.get(family)
    if key is None or not os.path.exists(path):
        return None
    tot = n = 0.0
    for name, r in json.load(open(path)).items():
        if key in name:
            tot += (r["read_bytes_per_launch"] + r["write_bytes_per_launch"]) * r["launches"]
            n += r["launches"]
    return round(tot / n) if n else None


def log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    print("[bench %6.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("XGGM_SHARE_GPU"):  # several ranks on one GPU (logic rehearsal only)
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_dp = world == 1 and bool(os.environ.get("XGGM_DP_FORCE"))  # one-rank RCCL rehearsal of the N > 1 path
    if force_dp:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("XGGM_DIST_BACKEND", "nccl")  # "gloo": logic check of the N>1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd.runtime import runtime_of
    if os.environ.get("XGGM_GROUP_TILE"):  # A/B hook: pin the tile of grouped GEMM launches (1: 64x64, 2: 128x64, 3: 128x128)
        from xggm_amd import _lib
        _lib.lib.xggm_gemm_set_group_tile(int(os.environ["XGGM_GROUP_TILE"]))

    log("building the model (rank %d/%d)" % (rank, world))
    model, optim, batch = build(args, device)
    log("model on %s" % device)
    # first forward creates the arena; data parallel hooks need it
    rt = runtime_of(model)
    if world > 1 or force_dp:
        from xggm_amd.vqa.vqacpv2 import enable_data_parallel
        enable_data_parallel(model, wire_dtype=torch.bfloat16 if args.wire == "bf16" else None)
    trainer = CapturedTrainer(model, optim, batch, sigma=1.0, order="vqa", use_graph=not args.no_graph)
    log("trainer ready (hip_graph=%s)" % (not args.no_graph))
    pyrng = random.Random(args.seed)  # identical draws on every rank (src/vqa/vqacpv2.py:192)

    def branch():
        return "rel" if pyrng.randint(1, 10) <= args.delta else "node"

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.iteration(branch())
    barrier()
    log("warmup done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.iteration(branch())
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_step = 1000.0 * dt / args.steps
    value = args.batch * world * args.steps / dt
    log("timed region: %.3f ms/step, %.1f samples/s" % (ms_step, value))

    # per-branch step time (diagnostic; not part of the timed region)
    per_branch = {}
    if rank == 0 or world > 1:
        for br in ("rel", "node"):
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                trainer.iteration(br)
            barrier()
            per_branch[br] = round(1000.0 * (time.perf_counter() - t1) / 5, 3)

    # per-pass time: forward + backward + clip + BertAdam of ONE pass (SURVEY section 8d asks for both views)
    per_pass = {}
    for kind in ("plain", "rel", "node"):
        barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            trainer.run_pass(kind)
        barrier()
        per_pass[kind] = round(1000.0 * (time.perf_counter() - t1) / 5, 3)

    roofline, kernels = None, None
    if not args.no_kernel_timing:
        # EVERY rank runs the instrumented passes (they contain the gradient exchange: a rank that skipped them
        # would leave the others waiting in a collective); rank 0 reports
        log("kernel timing pass")
        fam = kernel_timing(trainer, ["rel", "node"])
    if rank == 0 and not args.no_kernel_timing:
        tot = sum(f["ms"] for f in fam.values())
        kernels = {k: {"ms": round(f["ms"], 3), "launches": f["n"], "share": round(f["ms"] / tot, 4)}
                   for k, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:8]}
        dom = max(fam, key=lambda k: fam[k]["ms"])
        f = fam[dom]
        if dom.startswith("xggm_gemm_"):
            ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(dom),
                        "launches": f["n"], "avg_us": round(1000 * f["ms"] / f["n"], 2)}
        else:
            ach = f["bytes"] / (f["ms"] * 1e-3) / 1e9 if f["bytes"] else 0.0
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": pmc_traffic(dom),
                        "launches": f["n"], "avg_us": round(1000 * f["ms"] / f["n"], 2)}
        # the optimiser is the HBM-bound half of the step: always report it too
        if "xggm_bertadam_f32" in fam and fam["xggm_bertadam_f32"]["ms"] > 0:
            fo = fam["xggm_bertadam_f32"]
            kernels["xggm_bertadam_f32"]["GB/s"] = round(fo["bytes"] / (fo["ms"] * 1e-3) / 1e9, 1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle, one iteration)")
        cpu = cpu_baseline(args)
        log("cpu baseline done: %s" % cpu["value"])

    if rank == 0:
        line = {
            "metric": "train samples/sec (36-obj VQA batch); one step = plain + GGM pass, each fwd+bwd+clip+BertAdam",
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%dxMI355X %s: VQA-CP-v2-shaped synthetic batch=%d/GPU, 36 objects x 2048, "
                                   "20 tokens, A=%d, LXMERT 9/5/5 + GCNx2, full fwd+bwd+clip+BertAdam x2 passes, "
                                   "dropout on, delta=%d branch mix" % (world, args.dtype, args.batch, args.answers,
                                                                        args.delta),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "hip_graph": not args.no_graph, "grad_wire": args.wire if world > 1 else None},
            "ms_per_step_by_branch": per_branch, "ms_per_pass": per_pass,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1 or force_dp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
