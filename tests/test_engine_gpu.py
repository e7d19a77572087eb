"""The captured engine (what bench.py times): hipGraph replays of the three pass kinds against the same passes launched
eagerly, and size-independent properties of the full-size iteration (BASELINE configs[1]: 9/5/5 LXMERT + GCN x 2, 32
samples) that the oracle is too slow to check element by element."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from xggm_amd import synth  # noqa: E402
from helpers import batch_tensors, rel_err  # noqa: E402

DEV = "cuda"
BF16 = torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _tiny(seed_w, seed_rt, layers=(3, 2, 2), vocab=None, lr=1e-4):
    """a small model + its optimiser.  The comparisons below are EXACT (every reduction on the path has a fixed order:
    no floating-point atomics), so they do not depend on the trajectory being tame -- lr 1e-4 (x 4 for the heads) still
    keeps the tiny model's loss decreasing, which the loader test asserts, so that a failure reads as what it is."""
    from oracle import shapes
    from test_model_gpu import build_model
    from xggm_amd.vqa.vqacpv2 import make_optimizer
    cfg = dict(shapes.TINY, l_layers=layers[0], x_layers=layers[1], r_layers=layers[2])
    if vocab is not None:
        cfg["vocab"] = vocab
    m = build_model(cfg, 29, seed=seed_w, dt=BF16)
    m.seed = seed_rt
    return cfg, m, make_optimizer(m, lr, 40)


def _same_state(m1, m2):
    """weights, gradients, moments, step counters and the Philox state of two models, bit for bit"""
    from xggm_amd.runtime import runtime_of
    a1, a2 = runtime_of(m1).arena, runtime_of(m2).arena
    bad = [k for k in ("params", "grads", "m", "v", "shadow") if not torch.equal(getattr(a1, k), getattr(a2, k))]
    if bad:  # which arena groups?
        for k in bad:
            for g, G in a1.groups.items():
                if not torch.equal(getattr(a1, k)[G.start:G.end], getattr(a2, k)[G.start:G.end]):
                    d = (getattr(a1, k)[G.start:G.end].float() - getattr(a2, k)[G.start:G.end].float()).abs()
                    bad.append("%s/%s: %d elements, max %.3e" % (k, g, int((d > 0).sum()), float(d.max())))
    assert not bad, bad
    assert a1.steps.tolist() == a2.steps.tolist() and runtime_of(m1).rng.tolist() == runtime_of(m2).rng.tolist()


def test_replays_survive_what_eager_code_does_between_them():
    """a replay runs no Python of the pass, so host-side bookkeeping the captured kernels rely on has to be re-established
    by ``run_pass``: (a) after an EAGER pass that touched word-table rows the device-side row list does not name (two
    look-ups in one pass: the list is dropped), the captured clear -- which only knows the list -- must not leave those
    rows in the gradient; (b) an evaluation forward between two replays changes nothing.  (A loader batch shorter than
    the captured one: test_prefetching_loader_feeds_the_captured_trainer.)"""
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd.runtime import runtime_of
    from xggm_amd.vqa.vqacpv2 import BCEWithLogitsLoss
    B, A = 4, 29
    cfg, m1, o1 = _tiny(5, 11, vocab=512)
    _, m2, o2 = _tiny(5, 11, vocab=512)
    b = [batch_tensors(synth.vqa_batch(B, A=A, F=cfg["feat_dim"], vocab=cfg["vocab"], seed=s), DEV) for s in (3, 4, 5)]
    t1 = CapturedTrainer(m1, o1, b[0], sigma=1.0, warmup_iters=1)
    t2 = CapturedTrainer(m2, o2, b[0], sigma=1.0, warmup_iters=1)
    for t in (t1, t2):
        t.load_batch(b[1])
        t.iteration("rel")
    _same_state(m1, m2)
    # (a) on the first twin only: an eager double look-up over OTHER ids, gradients accumulated, no update
    rt, arena = runtime_of(m1), runtime_of(m1).arena
    assert arena.row_list is not None and arena.row_list.clean
    bce = BCEWithLogitsLoss()
    m1.zero_grad()
    for k in (2, 0):
        sent = (b[k]["input_ids"], b[k]["input_mask"], b[k]["segment_ids"])
        _, _, x = m1(b[k]["feats"], b[k]["boxes"], sent)
        rt.backward(bce(m1.logit_fc(x), b[k]["target"], scale=A))
    assert not arena.row_list.clean
    # (b) on both: an evaluation forward
    for m in (m1, m2):
        m.eval()
        with torch.no_grad():
            m(b[2]["feats"], b[2]["boxes"], (b[2]["input_ids"], b[2]["input_mask"], b[2]["segment_ids"]))
        m.train()
    for t in (t1, t2):
        t.load_batch(b[1])
        t.iteration("node")
    wt = m1.lxrt_encoder.model.bert.embeddings.word_embeddings.weight
    used = torch.zeros(wt.shape[0], dtype=torch.bool, device=DEV)
    used[b[1]["input_ids"].view(-1)] = True
    assert float(wt.grad[~used].abs().max()) == 0.0
    _same_state(m1, m2)


@pytest.mark.parametrize("order", ["vqa", "gqa"])
def test_captured_replay_equals_eager_passes(order):
    """CapturedTrainer: each pass kind is captured once and replayed; the replays must train the model exactly as the
    same sequence of eagerly launched passes does (same Philox stream: dropout on), including the warm-up passes the
    constructor runs before capturing, batches swapped in through ``load_batch``, and both iteration orders."""
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd.runtime import runtime_of
    from xggm_amd.vqa.vqacpv2 import plain_pass, ggm_pass, BCEWithLogitsLoss
    B, A = 4, 29
    cfg, m1, o1 = _tiny(5, 11)
    _, m2, o2 = _tiny(5, 11)
    batches = [batch_tensors(synth.vqa_batch(B, A=A, F=cfg["feat_dim"], vocab=cfg["vocab"], seed=s), DEV) for s in (3, 4)]
    branches = ["rel", "node", "rel"]
    klw = 8.0 if order == "vqa" else 12.0

    tr = CapturedTrainer(m1, o1, batches[0], sigma=1.0, order=order, warmup_iters=1)
    cap = []
    for i, br in enumerate(branches):
        tr.load_batch(batches[i % 2])
        (lp, _, _), (lg, _, _) = tr.iteration(br)
        cap += [float(lp), float(lg)]

    # the same schedule, launched eagerly on the twin
    bce = BCEWithLogitsLoss()
    rt2 = runtime_of(m2)
    m2.train()

    def run(kind, b):
        sent = (b["input_ids"], b["input_mask"], b["segment_ids"])
        if kind == "plain":
            out = plain_pass(m2, o2, bce, b["feats"], b["boxes"], sent, b["target"])
        else:
            out = ggm_pass(m2, o2, bce, b["feats"], b["boxes"], sent, b["target"], b["adj_true"], kind, 1.0, klw)
        rt2.advance()
        return float(out[0])

    for kind in ("plain", "rel", "node"):  # the constructor's warm-up iteration
        run(kind, batches[0])
    eager = []
    for i, br in enumerate(branches):
        b = batches[i % 2]
        if order == "vqa":
            eager += [run("plain", b), run(br, b)]
        else:
            g = run(br, b)
            eager += [run("plain", b), g]
    # the same kernels on the same data in the same order, and no reduction whose order depends on scheduling: a replay
    # IS the eager pass, bit for bit -- losses, weights, gradients, moments, step counters, Philox state
    assert cap == eager, (cap, eager)
    _same_state(m1, m2)


def test_full_size_iteration_properties():
    """BASELINE configs[1] on the captured engine (full 9/5/5 model, 32 samples, bf16, dropout on): losses and
    parameters stay finite; the clip norm the pass reports (GEMM slot table + norm pass) equals the fp64 norm of the
    gradient buffer; the first pass only changes what the warm-up schedule lets it change (lr factor 0 at step 0);
    an engine built the same way reproduces the loss trajectory."""
    import types
    import bench
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd.runtime import runtime_of
    args = bench.parse(["--steps", "4", "--warmup", "0"])  # the defaults ARE configs[1]: 32 samples, A = 2274, bf16
    assert (args.batch, args.answers, args.dtype, args.order, args.seed) == (32, 2274, "bf16", "vqa", 9595)
    traj = []
    for rep in range(2):
        model, optim, batch = bench.build(args, torch.device("cuda", 0))
        rt = runtime_of(model)
        tr = CapturedTrainer(model, optim, batch, sigma=1.0, order="vqa", warmup_iters=1)
        arena = rt.arena
        losses = []
        for br in ("rel", "node"):
            (lp, _, tp), (lg, _, tg) = tr.iteration(br)
            losses += [float(lp), float(lg)]
            # after a replay the gradient buffer holds the last (GGM) pass's gradients: every group but the head of
            # the other branch (a replay leaves no ``.grad`` attributes behind to ask)
            act = [G for g, G in arena.groups.items() if g != ("node_fc" if br == "rel" else "encoder_adj")]
            want = float(torch.sqrt(sum(float((arena.grads[G.start:G.end].double() ** 2).sum()) for G in act)
                                    * torch.ones((), dtype=torch.float64)))
            assert abs(float(tg) - want) < 1e-4 * want, (br, float(tg), want)
            assert float(tp) > 0
        assert all(np.isfinite(losses)) and all(x > 0 for x in losses)
        assert bool(torch.isfinite(arena.params).all()) and bool(torch.isfinite(arena.m).all())
        assert float(arena.sq_slots.sum()) > 0  # the encoder's weight gradients went through the slot table
        assert min(arena.steps.tolist()) >= 2
        traj.append(losses)
        del tr, model, optim
        torch.cuda.empty_cache()
    assert traj[0] == traj[1], traj  # reproducible to the bit (src/param.py:129-132: the reference run is reproducible too)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_data_parallel_ranks_on_one_gpu_stay_identical():
    """the data-parallel engine end to end, as far as one GPU allows: two ranks (gloo rendezvous on 127.0.0.1, both on
    cuda:0, DIFFERENT batches per rank) run eager and captured iterations with the plain and with the overlapped,
    five-stage gradient exchange; ``tools/dp_rehearsal.py`` asserts that the replicas stay bit-identical (the exchange
    really happens) and that the overlapped exchange trains exactly like the plain one."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "tools", "dp_rehearsal.py")], capture_output=True, text=True, timeout=600,
                       env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("replicas identical: True") == 4 and "replicas identical: False" not in r.stdout
    # the staged (overlapped) exchange trains exactly like the plain one: bit for bit, the rehearsal asserts it
    import re
    ds = [float(v) for v in re.findall(r"overlapped vs plain exchange: relative parameter difference ([0-9.e+-]+)", r.stdout)]
    assert len(ds) == 2 and max(ds) == 0.0, ds
    # the sharded update (ZeRO-1) on the wire arena: eager, two-graph and staged engines
    assert r.stdout.count("bf16 weights identical on all 2 ranks: True") == 3
    assert "sharded == replicated update bit for bit (clip not binding), 3 iterations: True" in r.stdout
    assert "sharded vs replicated update, one pass with the clip binding" in r.stdout
    assert "fp8 forward: sharded == replicated update bit for bit (clip not binding), 3 iterations: True" in r.stdout


@pytest.mark.parametrize("nowait", [0, 1])
def test_staged_all_gather_of_the_sharded_update_is_waited_for(nowait):
    """ZeRO-1: the bf16 weights come back stage by stage on the communication stream beside the next pass's forward
    graphs, and forward graph i waits for batch i only (dist.ShardedUpdate.gather_begin, CapturedTrainer.run_pass).  With
    every batch held back 50 ms the two-rank rehearsal must still train bit for bit like the replicated update -- and it
    must NOT when the engine is denied the events (the negative control: this is what shows that the rehearsal, whose
    passes follow each other without a host synchronisation, catches a forward that reads the previous step's weights)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", XGGM_GATHER_DELAY_US="50000", XGGM_REHEARSE_ONLY="sharded")
    if nowait:
        env["XGGM_REHEARSE_NOWAIT"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "tools", "dp_rehearsal.py")], capture_output=True, text=True, timeout=600,
                       env=env, cwd=root)
    ok = "sharded == replicated update bit for bit (clip not binding), 3 iterations: True"
    bad = "sharded == replicated update bit for bit (clip not binding), 3 iterations: False"
    assert "bf16 weights identical on all 2 ranks: True (forward cut into 4 + 1 graphs)" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    if nowait:
        # a CONTROL, not a requirement on the product: whether the unwaited forward really overtakes the delayed copies
        # depends on the box (it did on every box of the round); where it does not, the control says nothing
        if r.returncode == 0 and ok in r.stdout:
            pytest.skip("negative control inconclusive on this box: the delayed gather landed before the next forward ran")
        assert r.returncode != 0 and bad in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    else:
        assert r.returncode == 0 and ok in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_sharded_update_at_four_ranks_equals_the_replicated_one():
    """FOUR ranks share the one GPU (gloo; the box allows six processes on the card) and run the
    sharded update of tools/dp_rehearsal.py -- quarter slices of every matrix run, the staged all-gather from four
    owners beside the next forward's graphs: the bf16 weights must be identical on all four ranks under the eager, the
    two-graph and the staged engine, and three iterations must equal the replicated update bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", XGGM_REHEARSE_ONLY="sharded")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "tools", "dp_rehearsal.py")], capture_output=True, text=True, timeout=900,
                       env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("bf16 weights identical on all 4 ranks: True") == 3, r.stdout[-2000:]
    assert "sharded == replicated update bit for bit (clip not binding), 3 iterations: True" in r.stdout


@pytest.mark.parametrize("broken", ["", "NOJOIN", "NOAFTER"])
def test_staged_gradient_exchange_of_the_data_parallel_engine_is_waited_for(broken):
    """The overlapped exchange: the collectives of backward stage k run on the communication stream under graph k + 1.  Two
    waits hold it together -- the exchange of a stage waits for the graph that wrote the stage's gradients, the update
    waits for the communication stream.  Over gloo the summed slices come back through a device copy behind a 50 ms delay
    (dist.GradSync._all_reduce_in_place), so each wait can be shown to matter: with both in place the replicas of the
    two-rank rehearsal stay bit-identical, with either removed (tools/dp_rehearsal.py patches it out) they come apart."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", XGGM_GATHER_DELAY_US="50000", XGGM_REHEARSE_ONLY="exchange")
    if broken:
        env["XGGM_REHEARSE_" + broken] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "tools", "dp_rehearsal.py")], capture_output=True, text=True, timeout=600,
                       env=env, cwd=root)
    if broken:
        if r.returncode == 0 and "overlap=True: replicas identical: True" in r.stdout:  # a control: see the gather's twin above
            pytest.skip("negative control inconclusive on this box: the delayed sums landed before their reader ran")
        assert r.returncode != 0 and "overlap=True: replicas identical: False" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    else:
        assert r.returncode == 0 and "overlap=True: replicas identical: True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("zero1", [0, 1])
def test_bench_on_a_one_rank_rccl_group(zero1):
    """bench.py with XGGM_DP_FORCE=1: a real ``nccl`` (RCCL) process group of one rank, so the staged capture, the
    collectives on RCCL's stream between the replayed stage graphs and the watchdog thread are all live on one GPU;
    the JSON line must come out with the data-parallel configuration.  ``--zero1 1``: the sharded update's nccl
    branches (reduce-scatter of the stages, ``gather_begin`` / ``take_pending`` under the next forward,
    ``exchange_norm``) run on the one rank as well -- the opt-in path is exercised on RCCL before anyone measures it."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XGGM_DP_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-kernel-timing", "--no-ref-batch", "--zero1", str(zero1)], capture_output=True, text=True,
                       timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0 and line["config"]["hip_graph"] is True
    assert line["config"]["zero1"] is bool(zero1) and line["config"]["backend"] == "nccl"
    assert set(line["ms_per_pass"]) == {"plain", "rel", "node"} and all(v > 0 for v in line["ms_per_pass"].values())


def _run_bench(args, env_extra=None, timeout=1200):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):  # a launcher's variables must not leak into the child
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("order,zero1", [("vqa", 1), ("gqa", 0)])
def test_bench_gpus_2_spawns_two_ranks(order, zero1):
    """`python bench.py --gpus 2` with no launcher: the parent spawns the two ranks itself (before any GPU call) and
    rank 0 prints the line.  Here both ranks share the one GPU of the box and exchange over gloo (XGGM_SHARE_GPU,
    XGGM_DIST_BACKEND): the process group, the staged exchange, the max-over-ranks timing and the JSON are those
    of the real N = 2 run, only the transport differs.  ``gqa``: BASELINE configs[2]'s data-parallel workload
    (GQA-OOD-shaped: GGM pass first, KL x 12, A = 1842) in its default configuration (replicated update)."""
    extra = ["--order", "gqa"] if order == "gqa" else []
    line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline",
                       "--no-kernel-timing", "--zero1", str(zero1)] + extra, {"XGGM_DIST_BACKEND": "gloo", "XGGM_SHARE_GPU": "1"})
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["config"]["backend"] == "gloo"
    assert line["config"]["zero1"] is bool(zero1) and line["config"]["order"] == order
    assert ("A=1842" in line["config"]["workload"]) == (order == "gqa")
    assert line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["value_with_loader"]["value"] > 0 and line["value_ref_batch"] is None


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=root)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_gqa_order_line():
    """BASELINE configs[2] per GPU: GQA-OOD-shaped workload (GGM pass first, KL x 12, A = 1842; src/gqa/gqa_ood.py:165-292)"""
    line = _run_bench(["--gpus", "1", "--order", "gqa", "--answers", "1842", "--steps", "3", "--warmup", "1",
                       "--no-cpu-baseline"])
    assert line["n_gpus"] == 1 and line["config"]["order"] == "gqa" and "A=1842" in line["config"]["workload"]
    assert "GGM pass first" in line["config"]["workload"] and line["value"] > 0
    assert line["roofline"]["frac"] > 0 and line["value_with_loader"]["value"] > 0


def test_bench_fp8_line():
    """BASELINE configs[4] per GPU: e4m3 operands in the forward QKV / attention-output / FFN products, bf16 GNN and
    backward (src/lxrt/modeling.py:345-347, 428-431, 441-445 are the products that change type)"""
    line = _run_bench(["--dtype", "fp8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-loader", "--no-ref-batch"])
    assert line["dtype"] == "fp8" and line["n_gpus"] == 1 and line["value"] > 0 and "fp8" in line["config"]["workload"]
    assert "xggm_gemm_grouped_fp8e4m3" in line["kernels"] or any("fp8" in k for k in line["kernels"]), line["kernels"]


def test_bench_c4_stress_line():
    """BASELINE configs[3]: 64 objects x 64 adjacency, batch 64, aggregate kernel against the HBM roofline"""
    line = _run_bench(["--workload", "c4", "--steps", "10", "--warmup", "2"])
    assert line["config"]["global_batch"] == 64 and "64 objects" in line["config"]["workload"]
    rf = line["roofline"]
    assert rf["kernel"] == "xggm_aggregate_bf16" and rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["achieved"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["value"] > line["cpu_baseline"]["value"]


@pytest.mark.parametrize("handover", ["ring", "inline"])
def test_prefetching_loader_feeds_the_captured_trainer(tmp_path, handover):
    """shard -> VQATorchDataset -> DataLoaderX(device=cuda, batcher) -> CapturedTrainer: what arrives in the trainer's
    static input buffers equals the host items (features as their bf16 values), question strings have become the
    tokenised triple, batches keep arriving while earlier ones are still in use on the GPU, and training on loader
    batches leaves the engine in the state of a twin fed the same batches as plain tensors -- bit for bit.
    ``ring``: the producer copies to a device ring on its copy stream, ``load_batch`` hands over field by field;
    ``inline``: the producer fills pinned slots, ``load_packed`` ships a slot with ONE stream-ordered copy."""
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd.tools.shards import ShardWriter
    from xggm_amd.tools.data_loader import DataLoaderX
    from xggm_amd.vqa.vqacpv2_data import VQADataset, VQATorchDataset
    from xggm_amd.lxrt.entry import SentenceBatcher
    from xggm_amd.lxrt.tokenization import BertTokenizer
    from helpers import GOLDEN
    import os
    B, A, n_img = 4, 29, 12
    cfg, m, opt = _tiny(5, 11, vocab=96)  # the test vocabulary has 81 entries: every id inside the embedding table
    _, m2, opt2 = _tiny(5, 11, vocab=96)
    tok = BertTokenizer(os.path.join(GOLDEN, "vocab_small.txt"), do_lower_case=True)
    words = [w for w in open(os.path.join(GOLDEN, "vocab_small.txt")).read().split() if w.isalpha()][:40]
    rng = np.random.default_rng(0)
    w = ShardWriter(str(tmp_path / "train_obj36.xgs"), n_objects=36, feat_dim=cfg["feat_dim"])
    data = []
    bsrc = synth.vqa_batch(n_img, A=A, F=cfg["feat_dim"], vocab=cfg["vocab"], seed=21)
    for i in range(n_img):
        w.add(i, bsrc["feats"][i], bsrc["boxes"][i] * 0.99, 1.0, 1.0, bsrc["adj_true"][i])
        data.append({"question_id": i, "image_id": i, "label": [int(bsrc["target"][i].argmax())], "score": [1.0],
                     "question": " ".join(rng.choice(words, size=int(rng.integers(3, 9))))})
    l2a = ["a%d" % k for k in range(A)]
    ts = VQATorchDataset(VQADataset("train", data=data, ans2label={a: k for k, a in enumerate(l2a)}, label2ans=l2a),
                         shard=w.close())
    batcher = SentenceBatcher(tok, 20)
    it0 = iter(DataLoaderX(ts, B, device=DEV, batcher=batcher))
    first = next(it0)
    batch0 = dict(feats=first[1].clone(), boxes=first[2].clone(), input_ids=first[3][0].clone(), input_mask=first[3][1].clone(),
                  segment_ids=first[3][2].clone(), target=first[4].clone(), adj_true=first[5].clone())
    it0.close()  # an abandoned iterator would keep its producer thread parked on the queue
    loader = DataLoaderX(ts, B, device=DEV, batcher=batcher, depth=2, handover=handover)
    tr = CapturedTrainer(m, opt, batch0, warmup_iters=1, packed_spec=loader.spec if handover == "inline" else None)
    tr2 = CapturedTrainer(m2, opt2, batch0, warmup_iters=1)
    held, losses, losses2 = [], [], []
    it = iter(loader)
    for k, item in enumerate(it):
        qid, feats, boxes, sent, target, adj = item
        assert feats.dtype == torch.bfloat16 and feats.is_cuda == (handover == "ring") and sent[0].is_cuda == feats.is_cuda
        host = [ts[k * B + b] for b in range(B)]
        assert qid == [h[0] for h in host]
        assert torch.equal(feats.float().cpu(), torch.from_numpy(np.stack([h[1] for h in host])))
        assert torch.equal(boxes.cpu(), torch.from_numpy(np.stack([h[2] for h in host])))
        assert torch.equal(target.cpu(), torch.stack([h[4] for h in host]))
        assert torch.equal(adj.cpu(), torch.from_numpy(np.stack([h[5] for h in host])))
        if handover == "ring":
            bt = dict(feats=feats, boxes=boxes, input_ids=sent[0], input_mask=sent[1], segment_ids=sent[2], target=target,
                      adj_true=adj)
            tr.load_batch(bt)
        else:
            tr.load_packed(it)  # one copy of the pinned slot into the flat static buffer
            bt = {kk: v.clone() for kk, v in tr.static.items()}
            want = dict(feats=feats, boxes=boxes, input_ids=sent[0], input_mask=sent[1], segment_ids=sent[2], target=target,
                        adj_true=adj)
            for kk, v in want.items():  # (reads the pinned views: the slot is not rewritten before the next next())
                assert torch.equal(bt[kk].cpu(), v), kk
            feats = bt["feats"]
        (lp, _, _), (lg, _, _) = tr.iteration("rel")
        losses.append((float(lp), float(lg)))
        # the same batch as independent tensors through a second, identical engine
        tr2.load_batch({kk: v.clone() for kk, v in bt.items()})
        (lp2, _, _), (lg2, _, _) = tr2.iteration("rel")
        losses2.append((float(lp2), float(lg2)))
        for kk, v in bt.items():  # every static input buffer of both engines holds this batch, bit for bit
            assert torch.equal(tr.static[kk], v) and torch.equal(tr2.static[kk], v), kk
        _same_state(m, m2)
        held.append((feats, feats.clone()))
    # two engines that started equal and saw the same batches -- one through the loader's pinned ring and hand-over,
    # one as cloned tensors -- are in the same state BIT FOR BIT after every iteration (checked below, per iteration,
    # on weights, gradients, moments, counters): a batch mixed from two sample sets, a hand-over copy that raced a
    # replay or a buffer rewritten under a queued copy would show here, and nothing else can (every reduction on the
    # path has a fixed order)
    assert len(losses) == n_img // B and losses == losses2, (losses, losses2)
    if handover == "inline":
        # a batch shorter than the captured one fills only the head of its slot: refused, not trained on with a stale tail
        assert it.rows == B
        it.rows = B - 1
        with pytest.raises(ValueError, match="holds %d samples" % (B - 1)):
            tr.load_packed(it)
    assert losses[-1][0] < losses[0][0], losses  # lr 1e-4: the tiny model learns; the run is not a chaotic one
    torch.cuda.synchronize()


def _small_shard(tmp_path, n_img, F, A, vocab, seed=21):
    """a synthetic shard + dataset mirror of ``n_img`` images, one question each"""
    from xggm_amd.tools.shards import ShardWriter
    from xggm_amd.vqa.vqacpv2_data import VQADataset, VQATorchDataset
    from helpers import GOLDEN
    import os
    words = [w for w in open(os.path.join(GOLDEN, "vocab_small.txt")).read().split() if w.isalpha()][:40]
    rng = np.random.default_rng(0)
    w = ShardWriter(str(tmp_path / "train_obj36.xgs"), n_objects=36, feat_dim=F)
    bsrc = synth.vqa_batch(n_img, A=A, F=F, vocab=vocab, seed=seed)
    data = []
    for i in range(n_img):
        w.add(i, bsrc["feats"][i], bsrc["boxes"][i] * 0.99, 1.0, 1.0, bsrc["adj_true"][i])
        data.append({"question_id": i, "image_id": i, "label": [int(bsrc["target"][i].argmax())], "score": [1.0],
                     "question": " ".join(rng.choice(words, size=int(rng.integers(3, 9))))})
    l2a = ["a%d" % k for k in range(A)]
    return VQATorchDataset(VQADataset("train", data=data, ans2label={a: k for k, a in enumerate(l2a)}, label2ans=l2a),
                           shard=w.close())


@pytest.mark.parametrize("handover", ["ring", "inline"])
def test_loader_ring_survives_a_gpu_that_lags_the_host(tmp_path, handover):
    """a slot's PINNED buffer must not be rewritten while the host-to-device copy out of it is still queued.
    The GPU is parked behind a ~80 ms spin kernel while the host races through every batch of a depth-2 ring (each
    batch is only cloned on the stream -- nothing synchronises); the clones, read after the GPU has caught up, must be
    the host items of THEIR batch.  Before the producer waited for the slot's copy event, batch k came out holding
    rows of batch k + depth."""
    from xggm_amd.tools.data_loader import DataLoaderX
    from xggm_amd.lxrt.entry import SentenceBatcher
    from xggm_amd.lxrt.tokenization import BertTokenizer
    from helpers import GOLDEN
    import os
    B, A, n_img, F = 4, 29, 40, 64
    ts = _small_shard(tmp_path, n_img, F, A, 96)
    tok = BertTokenizer(os.path.join(GOLDEN, "vocab_small.txt"), do_lower_case=True)
    batcher = SentenceBatcher(tok, 20)
    want = [[ts[k * B + b] for b in range(B)] for k in range(n_img // B)]
    ref_ids = [SentenceBatcher(tok, 20).host_batch([h[3] for h in hs]).clone() for hs in want]
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2e8))  # ~80 ms: every copy below queues up behind this
    got = []
    if handover == "ring":
        for k, (qid, feats, boxes, sent, target, adj) in enumerate(DataLoaderX(ts, B, device=DEV, batcher=batcher, depth=2)):
            got.append((feats.clone(), boxes.clone(), torch.stack(sent).clone(), target.clone(), adj.clone()))
    else:
        # the consumer's own stream-ordered copy of the pinned slot (what CapturedTrainer.load_packed does): the slot
        # must not be rewritten before that queued copy has run
        from xggm_amd.tools.data_loader import packed_like
        loader = DataLoaderX(ts, B, device=DEV, batcher=batcher, depth=2, handover="inline")
        flat, view = packed_like(loader.spec, torch.device(DEV, torch.cuda.current_device()))
        it = iter(loader)
        for item in it:
            flat.copy_(it.flat, non_blocking=True)
            it.mark_copied()
            got.append((view["feats"].clone(), view["boxes"].clone(), view["ids"].clone(), view["target"].clone(),
                        view["adj"].clone()))
    torch.cuda.synchronize()
    assert len(got) == n_img // B
    for k, (feats, boxes, ids3, target, adj) in enumerate(got):
        hs = want[k]
        assert torch.equal(feats.float().cpu(), torch.from_numpy(np.stack([h[1] for h in hs]))), k
        assert torch.equal(boxes.cpu(), torch.from_numpy(np.stack([h[2] for h in hs]))), k
        assert torch.equal(ids3.cpu(), ref_ids[k]), k
        assert torch.equal(target.cpu(), torch.stack([h[4] for h in hs])), k
        assert torch.equal(adj.cpu(), torch.from_numpy(np.stack([h[5] for h in hs]))), k


def test_learning_rate_edit_reaches_replayed_graphs_and_split_param_groups_are_refused():
    """BertAdam reads the learning rate from a device table: editing ``param_groups[i]['lr']`` + ``sync_hyper()``
    between replays changes the captured update (lr 0 freezes the parameters, the original lr moves them again);
    an optimiser whose param_groups cut through an arena group (bias / LayerNorm in a no-decay group) is refused."""
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd.lxrt.optimization import BertAdam
    from xggm_amd.vqa.vqacpv2 import plain_pass, BCEWithLogitsLoss
    B, A = 4, 29
    cfg, m, opt = _tiny(5, 11)
    batch = batch_tensors(synth.vqa_batch(B, A=A, F=cfg["feat_dim"], vocab=cfg["vocab"], seed=3), DEV)
    tr = CapturedTrainer(m, opt, batch, warmup_iters=1)
    tr.iteration("rel")
    w = m.logit_fc[3].weight
    lrs = [pg["lr"] for pg in opt.param_groups]
    for pg in opt.param_groups:
        pg["lr"] = 0.0
    opt.sync_hyper()
    before = w.detach().clone()
    tr.iteration("rel")
    assert torch.equal(w.detach(), before)
    for pg, lr in zip(opt.param_groups, lrs):
        pg["lr"] = lr
    opt.sync_hyper()
    tr.iteration("rel")
    assert not torch.equal(w.detach(), before)
    # a no-decay group that takes the biases out of every arena group
    _, m2, _ = _tiny(5, 11)
    decay = [p for n, p in m2.named_parameters() if not n.endswith("bias")]
    no_decay = [p for n, p in m2.named_parameters() if n.endswith("bias")]
    bad = BertAdam([{"params": decay}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-3, warmup=0.1, t_total=10)
    sent = (batch["input_ids"], batch["input_mask"], batch["segment_ids"])
    with pytest.raises(ValueError, match="param_groups"):
        plain_pass(m2, bad, BCEWithLogitsLoss(), batch["feats"], batch["boxes"], sent, batch["target"])


def test_training_state_restored_under_live_graphs(tmp_path):
    """load_training_state writes weights, moments, step counters and the Philox state IN PLACE, so an engine whose
    graphs were captured before the restore continues exactly where the saved run stood: trainer A runs two
    iterations, saves, runs a third; trainer B (other weights, other dropout seed, graphs already captured and
    replayed) loads the snapshot and runs one iteration -- same losses, same parameters."""
    from xggm_amd.engine import CapturedTrainer
    from xggm_amd.vqa.vqacpv2 import save_training_state, load_training_state
    B, A = 4, 29
    cfg, ma, oa = _tiny(5, 11)
    _, mb, ob = _tiny(77, 99)
    batch = batch_tensors(synth.vqa_batch(B, A=A, F=cfg["feat_dim"], vocab=cfg["vocab"], seed=3), DEV)
    ta = CapturedTrainer(ma, oa, batch, warmup_iters=1)
    tb = CapturedTrainer(mb, ob, batch, warmup_iters=1)
    tb.iteration("rel")  # B has a history of its own
    for br in ("rel", "node"):
        ta.iteration(br)
    path = str(tmp_path / "state.pth")
    save_training_state(path, ma, oa, iteration=2)
    (lp_a, _, _), (lg_a, _, _) = ta.iteration("rel")
    want = [float(lp_a), float(lg_a)]
    assert load_training_state(path, mb, ob) == {"iteration": 2}
    (lp_b, _, _), (lg_b, _, _) = tb.iteration("rel")
    got = [float(lp_b), float(lg_b)]
    assert got == want, (got, want)  # exact resume: the restored run continues bit for bit
    sa, sb = ma.state_dict(), mb.state_dict()
    assert all(torch.equal(sb[k], sa[k]) for k in sa)
    assert oa.state_dict()["state"][0]["step"] == ob.state_dict()["state"][0]["step"]


def test_no_framework_kernel_inside_a_training_pass():
    """every device kernel a pass launches is one of the package's own: the plain, the relation
    and the node-generation pass of a small model run eagerly under torch.profiler -- no aten op may launch a kernel
    (at::native fills, adds, copies: F.pad of odd-width logit gradients, autograd's sums where a tensor feeds several
    consumers, select backward of the pooler, zero 'gradients' of non-differentiable outputs all used to)."""
    import os
    if os.environ.get("XGGM_POISON_EMPTY"):
        pytest.skip("the poisoned torch.empty of conftest.py fills every buffer with a framework kernel")
    from torch.profiler import ProfilerActivity, profile
    from xggm_amd.engine import CapturedTrainer
    B, A = 4, 29
    cfg, m, opt = _tiny(5, 11)
    batch = batch_tensors(synth.vqa_batch(B, A=A, F=cfg["feat_dim"], vocab=cfg["vocab"], seed=3), DEV)
    tr = CapturedTrainer(m, opt, batch, sigma=1.0, order="vqa", use_graph=False)
    tr.iteration("rel")
    tr.iteration("node")
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        tr.iteration("rel")
        tr.iteration("node")
        torch.cuda.synchronize()
    bad = []
    for ev in prof.events():
        if ev.name.startswith("aten::") and any(k.name for k in ev.kernels):
            bad.append((ev.name, str(ev.input_shapes)[:60], [k.name[:50] for k in ev.kernels][:2]))
    assert not bad, bad[:8]
