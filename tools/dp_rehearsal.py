"""Rehearsal of the data-parallel engine paths on ONE GPU: world size 2 over gloo, both ranks on cuda:0, a tiny
model, DIFFERENT batches per rank.  Checks (a) replicas stay identical (the exchange really happens), (b) the
overlapped three-graph path gives the same parameters as the plain two-graph path.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/dp_rehearsal.py
XGGM_GATHER_DELAY_US=3000 holds every batch of the sharded update's staged all-gather back by 3 ms: the checks must still
pass (the forward graphs wait for their batch).  XGGM_REHEARSE_NOWAIT=1 on top removes those waits -- the negative
control: the checks must then FAIL (over gloo the gathered slices reach the weights through an asynchronous device copy
behind the delay, dist.ShardedUpdate._gather_runs, so a forward graph that does not wait reads the previous step's
weights).  The same for the staged gradient exchange: XGGM_REHEARSE_NOJOIN=1 (the update does not wait for the
communication stream) and XGGM_REHEARSE_NOAFTER=1 (the exchange of a stage does not wait for the backward graph that wrote
its gradients) must make the replicas come apart -- over gloo the summed slices come back through a delayed device copy too
(dist.GradSync._all_reduce_in_place)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(overlap, rank, use_graph=True, iters=("rel", "node", "rel"), layers=(2, 2, 1), zero1=False, want="params", clip=5.0,
        fp8=False):
    from xggm_amd import synth
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd.vqa.vqacpv2 import enable_data_parallel, make_optimizer
    from test_model_gpu import build_model, batch_tensors
    cfg = dict(hidden=128, heads=2, inter=256, vocab=64, max_pos=32, feat_dim=64, l_layers=layers[0], x_layers=layers[1],
               r_layers=layers[2])
    A, B = 29, 4
    m = build_model(cfg, A, seed=5, dt=torch.bfloat16)
    bn = synth.vqa_batch(B, A=A, F=cfg["feat_dim"], vocab=cfg["vocab"], seed=100 + rank)  # rank-specific data
    b = batch_tensors(bn, "cuda")
    m(b["feats"], b["boxes"], (b["input_ids"], b["input_mask"], b["segment_ids"]))
    opt = make_optimizer(m, 1e-3, 20)
    if fp8:
        from xggm_amd.fp8 import enable_fp8
        enable_fp8(m)
    enable_data_parallel(m, wire_dtype=torch.bfloat16, overlap=overlap, zero1=zero1)
    tr = CapturedTrainer(m, opt, b, sigma=1.0, order="vqa", clip=clip, use_graph=use_graph, warmup_iters=1)
    norms = []
    for br in iters:
        outs = [tr.run_pass("plain")] if br == "plain" else tr.iteration(br)
        for out in outs:
            norms.append(out[2].clone())  # the clip norm of the pass (read after the loop: NO host synchronisation between
            # the passes -- the next pass's forward graphs are queued while the staged all-gather of this pass still runs)
    torch.cuda.synchronize()
    run.norms = norms = [float(x) for x in norms]
    g = tr.graphs.get("rel") if use_graph else None
    run.n_fwd = g[5] if g is not None and g[0] == "staged" else 0  # forward graphs in front of the backward (sharded update)
    run.names = [(n, p.numel()) for n, p in m.named_parameters()]
    arena = m.arena()
    run.info = dict(arena.info)
    run.runs = list(arena.zero1.all_runs) if getattr(arena, "zero1", None) is not None else []
    if want == "shadow":  # what the GEMMs read: must be identical on every rank after the all-gather
        return arena.shadow.float().clone()
    # what a checkpoint writes: under the sharded update state_dict() first gathers the fp32 masters of the other ranks'
    # slices (runtime._gather_before_state_dict -> ParamArena.gather_sharded_state; a collective, both ranks are here)
    if zero1:
        assert arena.zero1.stale
    sd = m.state_dict()
    if zero1:
        assert not arena.zero1.stale
    return torch.cat([sd[n].detach().float().flatten() for n, _ in m.named_parameters()])


def gather_all(v):
    """v of every rank (world size 2 by default; XGGM_REHEARSE_ONLY=sharded also runs at 4 ranks sharing the GPU)"""
    other = [torch.empty_like(v) for _ in range(dist.get_world_size())]
    dist.all_gather(other, v)
    return other


def all_equal(other):
    return all(torch.equal(other[0], o) for o in other[1:])


def report(v, rank, tag):
    other = gather_all(v)
    if rank == 0:
        d = torch.stack([(other[0] - o).abs() for o in other[1:]]).max(0).values
        print("%s: max |rank0 - rank r| = %.3e" % (tag, float(d.max())), flush=True)
        o, worst = 0, []
        for n, k in run.names:
            worst.append((float(d[o:o + k].max()), n))
            o += k
        worst.sort(reverse=True)
        print("   worst:", worst[:6], flush=True)


def main():
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    # negative controls of the staged gradient exchange (run with XGGM_GATHER_DELAY_US): the update must wait for the
    # communication stream (XGGM_REHEARSE_NOJOIN=1 removes that wait), and the exchange of a stage must wait for the
    # backward graph that wrote the stage's gradients (XGGM_REHEARSE_NOAFTER=1 removes that one); either way the replicas
    # must come apart.  Only waits that involve a trainer's communication stream are touched.
    if os.environ.get("XGGM_REHEARSE_NOJOIN") or os.environ.get("XGGM_REHEARSE_NOAFTER"):
        from xggm_amd.engine import CapturedTrainer
        comm_streams = []
        make = CapturedTrainer._comm_stream

        def registering(self):
            st = make(self)
            if all(st is not c for c in comm_streams):
                comm_streams.append(st)
            return st

        CapturedTrainer._comm_stream = registering
        if os.environ.get("XGGM_REHEARSE_NOJOIN"):
            ws = torch.cuda.Stream.wait_stream
            skipped = [0]

            def no_join(self, other):
                if any(other is c for c in comm_streams):
                    skipped[0] += 1
                    if skipped[0] in (1, 10):
                        print("rank %d: wait_stream(communication stream) skipped (%d so far)" % (rank, skipped[0]), flush=True)
                    return None
                return ws(self, other)

            torch.cuda.Stream.wait_stream = no_join
        if os.environ.get("XGGM_REHEARSE_NOAFTER"):
            we = torch.cuda.Stream.wait_event
            torch.cuda.Stream.wait_event = lambda self, ev: None if any(self is c for c in comm_streams) else we(self, ev)
    if os.environ.get("XGGM_REHEARSE_NOWAIT"):
        # negative control for the staged all-gather (run with XGGM_GATHER_DELAY_US=3000): the engine gets no events to wait
        # for; the forward graphs then read the weights of the previous step and the bit-exactness checks fail (docstring)
        from xggm_amd.dist import ShardedUpdate
        ShardedUpdate.take_pending = lambda self: []
    if os.environ.get("XGGM_REHEARSE_ONLY") == "sharded":  # the staged gather's controls (tests/test_engine_gpu.py)
        check_sharded(rank)
        dist.destroy_process_group()
        return
    if os.environ.get("XGGM_REHEARSE_ONLY") == "exchange":  # the staged gradient exchange's controls
        check(rank, (5, 4, 4), modes=(True,))  # the staged engine alone: replicas must stay identical
        dist.destroy_process_group()
        return
    report(run(False, rank, use_graph=False, iters=("rel",)), rank, "eager, 1 iteration")
    for layers in ((2, 2, 1), (5, 4, 4)):  # two cuts (three backward stages) / four cuts (five stages)
        check(rank, layers)
    check_sharded(rank)
    check_sharded_fp8(rank)
    dist.destroy_process_group()


def check_sharded_fp8(rank):
    """the sharded update under the fp8 forward (BASELINE configs[4] is defined on 8 GPUs): every rank writes the e4m3
    copies of ITS slices, the maxima behind the weight scales are MAX-reduced over the ranks, the e4m3 copies are
    gathered with the bf16 ones -- the training must equal the replicated fp8 run bit for bit (clip not binding)."""
    ref = run(True, rank, layers=(2, 2, 1), clip=1e9, fp8=True)
    got = run(True, rank, layers=(2, 2, 1), zero1=True, clip=1e9, fp8=True)
    other = gather_all(got)
    assert all_equal(other), "fp32 masters differ after gather_state (fp8)"
    same = bool(torch.equal(got, ref))
    if rank == 0:
        print("fp8 forward: sharded == replicated update bit for bit (clip not binding), 3 iterations: %s" % same, flush=True)
    assert same


def check_sharded(rank):
    """ZeRO-1: reduce-scatter (gloo: all-reduce + own slice) -> BertAdam on the own slices -> all-gather of the bf16
    weights.  The weights every rank computes with stay identical; after gather_state so do the fp32 masters; and the
    training equals the unsharded one (same averaged gradients, element-wise update; the clip norm is summed in
    another order)."""
    for overlap, use_graph in ((False, False), (False, True), (True, True)):
        sh = run(overlap, rank, use_graph=use_graph, layers=(5, 4, 4), zero1=True, want="shadow")
        other = gather_all(sh)
        same = bool(all_equal(other))
        if rank == 0:
            print("sharded update overlap=%s graphs=%s: bf16 weights identical on all %d ranks: %s (forward cut into %d + 1 graphs)"
                  % (overlap, use_graph, dist.get_world_size(), same, run.n_fwd), flush=True)
            if not same:  # which tensors, which ranks, where inside them
                for r, o_ in enumerate(other[1:], 1):
                    d = (other[0] != o_)
                    for n, (off, k, g, atomic) in sorted(run.info.items(), key=lambda kv: kv[1][0]):
                        nd = int(d[off:off + k].sum())
                        if nd:
                            idx = torch.nonzero(d[off:off + k]).flatten()
                            print("   rank %d differs from rank 0 in %s (group %s, offset %d, %d elements): %d elements, first %d last %d"
                                  % (r, n, g, off, k, nd, int(idx[0]), int(idx[-1])), flush=True)
                print("   runs:", sorted(run.runs), flush=True)
        assert same
        # with the staged exchange the forward is cut too: the all-gather of stage i + 1 runs beside forward graph i
        assert run.n_fwd == (4 if overlap and use_graph else 0), run.n_fwd
    # The sharded and the replicated update differ in ONE thing: the order the clip norm is summed in (slices' partial
    # norms, all-reduced, against one fixed-order sum over the buffer).  Everything else is element-wise on the same
    # averaged gradients.  So (a) with a clip that does not bind (coefficient exactly 1 in both) three iterations must
    # agree BIT FOR BIT; (b) with the reference's clip of 5 binding, ONE parameter-moving pass may differ by what an ulp of the
    # coefficient does to one BertAdam step: |d u / u| <= 2^-23 on u = m / (sqrt(v) + eps) + wd p, i.e. a relative
    # parameter difference below lr * 3.2 * 2^-23 / |p| ~ 1e-9 -- asserted with three orders of magnitude to spare; a
    # slice nobody updates or a run gathered from the wrong owner moves whole tensors by lr * 3.16 (>= 1e-3 relative).
    # No bound on a multi-pass trajectory with the clip binding: it would be a guess (the generation pass's column
    # arg-max amplifies an ulp unpredictably).
    ref = run(True, rank, layers=(5, 4, 4), clip=1e9)
    got = run(True, rank, layers=(5, 4, 4), zero1=True, clip=1e9)
    other = gather_all(got)
    assert all_equal(other), "fp32 masters differ after gather_state"
    same = bool(torch.equal(got, ref))
    if rank == 0:
        print("sharded == replicated update bit for bit (clip not binding), 3 iterations: %s" % same, flush=True)
        if not same:
            o = 0
            for n, k in run.names:
                dd = (got[o:o + k] - ref[o:o + k]).abs()
                if float(dd.max()) > 0:
                    print("   differs: %s max %.3e, %d of %d elements" % (n, float(dd.max()), int((dd > 0).sum()), k), flush=True)
                o += k
    assert same
    # eager, so that no warm-up passes precede the two that are compared: pass 1 runs at schedule value 0 (it only fills
    # the moments), pass 2 moves the parameters; plain passes only -- no discrete step anywhere in them
    ref = run(True, rank, layers=(5, 4, 4), iters=("plain", "plain"), use_graph=False)
    n_ref = run.norms
    got = run(True, rank, layers=(5, 4, 4), zero1=True, iters=("plain", "plain"), use_graph=False)
    d = float((got - ref).double().norm() / ref.double().norm())
    if rank == 0:
        print("   clip norm of the pass, replicated: %.6f, sharded: %.6f" % (n_ref[-1], run.norms[-1]), flush=True)
        print("sharded vs replicated update, one pass with the clip binding: relative parameter difference %.2e" % d, flush=True)
    assert min(n_ref) > 5.0, "the clip was meant to bind in this check"
    assert d < 1e-6


def check(rank, layers, modes=(False, True)):
    out = {}
    for overlap in modes:
        v = run(overlap, rank, layers=layers)
        if os.environ.get("XGGM_DEBUG_RUNS"):
            print("rank %d overlap=%s clip norms %s" % (rank, overlap, " ".join("%.6f" % x for x in run.norms)), flush=True)
        same = bool(all_equal(gather_all(v)))
        out[overlap] = v
        if rank == 0:
            print("layers %s overlap=%s: replicas identical: %s, |params| = %.6f" % (layers, overlap, same, float(v.double().norm())),
                  flush=True)
        assert same, "replicas diverged: the gradient exchange did not happen"
    if len(modes) < 2:
        return
    d = float((out[True] - out[False]).double().norm() / out[False].double().norm())
    if rank == 0:
        print("overlapped vs plain exchange: relative parameter difference %.2e" % d, flush=True)
    if d > 0 and os.environ.get("XGGM_DEBUG_RUNS"):
        again = run(False, rank, layers=layers)
        if rank == 0:
            print("   plain vs plain again: %.2e" % float((out[False] - again).double().norm() / again.double().norm()), flush=True)
            o = 0
            for n, k in run.names:
                dd = (out[False][o:o + k] - out[True][o:o + k]).abs()
                if float(dd.max()) > 0:
                    print("   differs: %s max %.3e, %d elements" % (n, float(dd.max()), int((dd > 0).sum())), flush=True)
                o += k
    assert d == 0.0, "the staged exchange must train exactly like the plain one (no scheduling-dependent sums left)"


if __name__ == "__main__":
    main()
