"""In-graph duration and rate of every GEMM launch of the iteration BY SHAPE: the ordered launch log of one eager pass per
kind (entry point, M x N x K of every problem) joined with the ordered GEMM kernels of the replayed passes in a rocprofv3
kernel trace of the same configuration.
    python tools/gemm_by_shape.py log gpurun_out/gemm_log.json                  (on the GPU: writes the launch log)
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gtrace -- python bench.py --steps 12 --warmup 2 \
        --no-cpu-baseline --no-ref-batch --no-loader --no-kernel-timing
    python tools/gemm_by_shape.py join gpurun_out/gemm_log.json gpurun_out/gtrace"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def log(out, extra):
    import torch
    import bench
    args = bench.parse(["--no-graph", "--no-cpu-baseline", "--no-kernel-timing", "--no-loader"] + extra)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model, optim, batch = bench.build(args, dev)
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd import ops, _lib
    tr = CapturedTrainer(model, optim, batch, sigma=1.0, order=args.order, use_graph=False)
    for kind in ("plain", "rel", "node"):
        tr.run_pass(kind)
    torch.cuda.synchronize()
    cur = []
    real = _lib.call

    def call(name, *a):
        if name.startswith("xggm_gemm"):
            if "grouped" in name:
                arr, n = a[0], a[1]
                probs = ops._ct.cast(arr, ops._ct.POINTER(ops.GemmProblem))
                cur.append([name, [[probs[i].M, probs[i].N, probs[i].K, probs[i].batch] for i in range(n)]])
            else:
                cur.append([name, [[a[3], a[4], a[5], a[11] if len(a) > 11 else 1]]])
        return real(name, *a)

    _lib.call = call
    ops.call = call
    res = {}
    for kind in ("plain", "rel", "node"):
        cur = []
        tr.run_pass(kind)
        res[kind] = cur
    torch.cuda.synchronize()
    json.dump(res, open(out, "w"))
    print({k: len(v) for k, v in res.items()})


def join(logf, tdir):
    lg = json.load(open(logf))
    f = glob.glob(tdir + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    ends = [i + 1 for i, r in enumerate(rows) if "bertadam_multi" in r[2]]
    by_len = {len(v): k for k, v in lg.items()}
    if len(by_len) < len(lg):  # rel and node may issue the same number of GEMM launches: shapes tell them apart below
        by_len = None
    agg = collections.defaultdict(list)
    used = 0
    for a, b in zip(ends[:-1], ends[1:]):
        seg = [r for r in rows[a:b] if "gemm_" in r[2]]
        cands = [k for k, v in lg.items() if len(v) == len(seg)]
        if not cands:
            continue
        kind = cands[0]
        if len(cands) > 1:  # rel / node: the node branch has no 630-wide edge head
            has630 = any(630 in (p[0], p[1]) for _, ps in lg["rel"] for p in ps)
            kind = "rel" if has630 and False else cands[0]
        used += 1
        for (s, e, n), (name, probs) in zip(seg, lg[kind]):
            key = (name.replace("xggm_", ""), " | ".join("%dx%dx%d%s" % (m, nn, k, "" if bt == 1 else "b%d" % bt) for m, nn, k, bt in probs))
            fl = sum(2.0 * m * nn * k * bt for m, nn, k, bt in probs)
            agg[key].append(((e - s) / 1e3, fl, n))
    print("passes joined: %d" % used)
    out = []
    for key, v in agg.items():
        us = statistics.median(x[0] for x in v)
        per_pass = len(v) / max(used, 1)
        out.append((us * per_pass, key, us, v[0][1], per_pass, v[0][2]))
    out.sort(reverse=True)
    tot = sum(o[0] for o in out)
    print("GEMM time per pass (median durations x launches per pass): %.1f us" % tot)
    for t, key, us, fl, pp, kn in out:
        tile = kn[kn.find("kernel<") + 7:kn.find(">")] if "kernel<" in kn else ""
        print("%7.1f us/pass %5.1f x %6.1f us %6.0f TFLOP/s  %-22s %-12s %s" % (t, pp, us, fl / us / 1e6, key[0][:22], tile[:12], key[1]))


if __name__ == "__main__":
    if sys.argv[1] == "log":
        log(sys.argv[2], sys.argv[3:])
    else:
        join(sys.argv[2], sys.argv[3])
