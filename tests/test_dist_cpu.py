"""world_size-2 gloo tests (CPU) of the data-parallel host logic: bucketed averaging of a flat
gradient buffer, bf16 wire compression, branch agreement, batch sharding.  The HIP kernels are
not involved (they need a GPU); this covers the N>1 path's bookkeeping by construction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xggm_amd.dist import GradSync, sync_branch, shard_batch, ranges_to_buckets
        n = 10_000
        base = torch.arange(n, dtype=torch.float32)
        g = base * (rank + 1)              # rank r holds (r+1) * base
        ranges = [(8, 1000), (2000, 9992)]  # two "active groups"; the rest must stay untouched
        gs = GradSync(g, bucket_elems=1024)
        gs.sync(ranges)
        mean = base * (sum(range(1, world + 1)) / world)
        ok = True
        for s, e in ranges:
            ok &= torch.allclose(g[s:e], mean[s:e])
        ok &= torch.equal(g[:8], base[:8] * (rank + 1)) and torch.equal(g[1000:2000], base[1000:2000] * (rank + 1))
        # bf16 on the wire: result within bf16 rounding of the mean
        g2 = base * (rank + 1)
        GradSync(g2, wire_dtype=torch.bfloat16, bucket_elems=4096).sync([(0, n)])
        ok &= bool(((g2 - mean).abs() <= 8e-3 * mean.abs() + 1e-6).all())
        # many small buckets: both wire buffers are reused several times; untouched gaps stay untouched
        g3 = base * (rank + 1)
        GradSync(g3, wire_dtype=torch.bfloat16, bucket_elems=700).sync(ranges)
        for s, e in ranges:
            ok &= bool(((g3[s:e] - mean[s:e]).abs() <= 8e-3 * mean[s:e].abs() + 1e-6).all())
        ok &= torch.equal(g3[:8], base[:8] * (rank + 1)) and torch.equal(g3[1000:2000], base[1000:2000] * (rank + 1))
        # every rank follows rank 0's branch
        br = sync_branch(rank == 0, "cpu")
        ok &= (br is True)
        # sharding
        batch = {"x": torch.arange(8).view(8, 1), "sent": "keep"}
        sh = shard_batch(batch, rank, world)
        ok &= sh["x"].flatten().tolist() == list(range(rank * 4, rank * 4 + 4)) and sh["sent"] == "keep"
        ok &= ranges_to_buckets([(0, 10), (20, 25)], 4) == [(0, 4), (4, 8), (8, 10), (20, 24), (24, 25)]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_grad_sync_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_is_noop():
    from xggm_amd.dist import GradSync
    g = torch.ones(16)
    GradSync(g).sync([(0, 16)])
    assert torch.equal(g, torch.ones(16))


def test_stage_ranges_tile_the_full_size_arena():
    """dist.stage_ranges on the layout of the full 9/5/5 model (names and sizes from oracle.shapes, offsets laid out
    the way ParamArena does): the four backward stages tile every active range exactly, the cross-layer matrices of
    the last two x-layers and all head groups are in stage 0, nothing of enc_main's vector region before the last."""
    from oracle import shapes
    from xggm_amd import arena as A
    from xggm_amd.dist import stage_ranges

    class G:
        pass

    class FakeModel:
        _enc_tail_prefixes = tuple("lxrt_encoder.model.bert.encoder.x_layers.4." + s
                                   for s in ("visn_self_att.", "visn_inter.", "visn_output."))

    class FakeParam:
        def __init__(self, shape):
            self.shape = tuple(shape)

        def dim(self):
            return len(self.shape)

        def numel(self):
            n = 1
            for d in self.shape:
                n *= d
            return n

    named = [(k, FakeParam(v)) for k, v in shapes.model_shapes(shapes.FULL, 2274).items()]
    _, groups, info, _ = A.layout(named, A.default_group_of, FakeModel)
    fake = G()
    fake.groups, fake.info = groups, info
    active = [(g.start, g.end) for g in groups.values()]
    pre = "lxrt_encoder.model.bert.encoder."

    def check(st, n):
        assert len(st) == n and all(st)
        tiles = sorted(r for s_ in st for r in s_)
        assert all(a[1] <= b[0] for a, b in zip(tiles, tiles[1:]))
        assert sum(e - s_ for s_, e in tiles) == sum(e - s_ for s_, e in active)

        def stage_of(name):
            o = info[name][0]
            return next(k for k, rs in enumerate(st) if any(a <= o < b for a, b in rs))

        return stage_of, [sum(e - s_ for s_, e in rs) for rs in st]

    # cuts above the embeddings, after two layer pairs, before the first and the second-to-last cross layer
    stage_of, sizes = check(stage_ranges(fake, active, dict(pair_cut=2, x_mid=3, emb_cut=True), 5), 5)
    assert stage_of(pre + "x_layers.4.visual_attention.att.query.weight") == 0
    assert stage_of(pre + "x_layers.3.lang_inter.dense.weight") == 0
    assert stage_of(pre + "x_layers.2.lang_inter.dense.weight") == 1
    assert stage_of(pre + "x_layers.0.visual_attention.att.query.weight") == 1
    assert stage_of(pre + "layer.8.output.dense.weight") == 2 and stage_of(pre + "r_layers.2.output.dense.weight") == 2
    assert stage_of(pre + "layer.1.output.dense.weight") == 3 and stage_of(pre + "r_layers.0.output.dense.weight") == 3
    assert stage_of(pre + "visn_fc.visn_fc.weight") == 4
    assert stage_of("logit_fc.3.weight") == 0 and stage_of("generator.gnn_layers.0.gnn_layers.0.ctx_layer.weight") == 0
    assert stage_of(pre + "x_layers.4.visual_attention.output.LayerNorm.weight") == 4  # vectors: only after the last stage
    assert stage_of("lxrt_encoder.model.bert.embeddings.word_embeddings.weight") == 4
    # 46 / 50 / 71 / 28 / 26 M parameters: the exposed tail is the embedding tables + vectors + visn_fc
    assert sum(sizes) > 2.2e8 and all(4e7 < z < 8e7 for z in sizes[:3]) and all(2e7 < z < 3.2e7 for z in sizes[3:])
    # without the cut above the embeddings the two lowest regions are final together
    stage_of, sizes4 = check(stage_ranges(fake, active, dict(pair_cut=2, x_mid=3), 4), 4)
    assert stage_of(pre + "layer.1.output.dense.weight") == 3 and stage_of(pre + "visn_fc.visn_fc.weight") == 3
    assert sizes4[:3] == sizes[:3] and sizes4[3] == sizes[3] + sizes[4]
    # a forward that recorded another number of cuts than the layout predicts: everything after the last stage
    st = stage_ranges(fake, active, dict(pair_cut=2, x_mid=3, emb_cut=True), 3)
    assert [len(x) for x in st[:2]] == [0, 0] and sum(e - s_ for s_, e in st[2]) == sum(e - s_ for s_, e in active)


def _full_size_arena():
    """the flat-buffer layout of the full 9/5/5 model (names and sizes from oracle.shapes), without any tensor"""
    from oracle import shapes
    from xggm_amd import arena as A

    class G:
        pass

    class FakeModel:
        _enc_tail_prefixes = tuple("lxrt_encoder.model.bert.encoder.x_layers.4." + s
                                   for s in ("visn_self_att.", "visn_inter.", "visn_output."))

    class FakeParam:
        def __init__(self, shape):
            self.shape = tuple(shape)

        def dim(self):
            return len(self.shape)

        def numel(self):
            n = 1
            for d in self.shape:
                n *= d
            return n

    named = [(k, FakeParam(v)) for k, v in shapes.model_shapes(shapes.FULL, 2274).items()]
    _, groups, info, total = A.layout(named, A.default_group_of, FakeModel)
    fake = G()
    fake.groups, fake.info, fake.total = groups, info, total
    return fake


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_slices_tile_the_full_size_arena(world):
    """the slicing of the sharded update on the REAL layout at 4 and 8 ranks (the other tests run it
    on a 2048-element toy arena at world size 2).  For every backward stage of the five-stage cut: the matrix runs are
    whole 256-element chunks, every run divides by the world size, the ranks' own slices tile it without gap or
    overlap, every slice starts on a 32-element (64-byte bf16) border -- so a 4-element quad of the update kernel never
    straddles two chunks of the e4m3 scale-id table, which is indexed by absolute element offset
    (loss_optim.hip: w8_id[(elem0 + 4 i) >> 8]) -- and the batches, walked in reverse, visit the encoder in forward
    order (what the staged all-gather under the next forward relies on)."""
    from xggm_amd import arena as A
    from xggm_amd.dist import ShardedUpdate, stage_ranges
    fake = _full_size_arena()
    active = [(g.start, g.end) for g in fake.groups.values()]
    stages = stage_ranges(fake, active, dict(pair_cut=2, x_mid=3, emb_cut=True), 5)
    pair_cut, x_mid = A.encoder_cuts(list(fake.info))
    assert (pair_cut, x_mid) == (2, 3)
    by_offset = sorted((o, k, n) for n, (o, k, g, atomic) in fake.info.items())
    n_mat = n_own = 0
    first_region = []
    for rank in range(world):
        z = ShardedUpdate.__new__(ShardedUpdate)  # the slicing logic only: no process group, no tensors
        z.arena, z.world, z.rank = fake, world, rank
        regions = []
        for st in stages:
            mats, vecs = z.split(st)
            assert sum(e - s for s, e in mats + vecs) == sum(e - s for s, e in st)
            for a, b in mats:
                assert a % A.ALIGN_MAT == 0 and (b - a) % A.ALIGN_MAT == 0, (a, b)
                o0, o1 = z.own((a, b))
                c = (b - a) // world
                assert (o0, o1) == (a + rank * c, a + (rank + 1) * c) and c * world == b - a
                assert o0 % 32 == 0 and c % 32 == 0
                # runs of encoder matrices carry e4m3 copies under the fp8 forward: xggm_bertadam_ex takes the scale-table
                # entry per 256-element chunk and per WAVE (elem0 % 256 == 0 is a checked argument), so there a slice must
                # be whole chunks -- every encoder matrix is a multiple of 8 * 256 elements, hence so is every run
                if any(G.name in ("enc_main", "enc_tail") and G.start <= a and b <= G.vec_start for G in fake.groups.values()):
                    assert o0 % A.ALIGN_MAT == 0 and c % A.ALIGN_MAT == 0, (a, b, world)
                if rank == 0:
                    n_mat += b - a
                n_own += o1 - o0
            # forward regions of the encoder matrices this stage exchanges
            rs = [A.region_of(n, pair_cut, x_mid) for o, k, n in by_offset
                  if ".encoder." in n and any(a <= o < b for a, b in mats)]
            regions.append((min(rs), max(rs)) if rs else None)
        if rank == 0:
            first_region = regions
    assert n_own == n_mat and n_mat > 1.9e8  # the ranks' slices add up to every matrix element, 88 % of the model
    # backward stage k + 1 holds LOWER forward regions than stage k: reversed = the order the next forward reads them
    enc = [r for r in first_region if r is not None]
    assert all(lo_next <= lo and hi_next <= lo for (lo, hi), (lo_next, hi_next) in zip(enc, enc[1:])), first_region
    assert enc[0][1] == 4 and enc[-1][0] <= 1


def _sharded_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xggm_amd.dist import ShardedUpdate, GradSync

        class Grp:
            def __init__(self, start, vec_start, end):
                self.start, self.vec_start, self.end = start, vec_start, end

        class Arena:
            pass

        a = Arena()
        a.groups = {"g0": Grp(0, 1024, 1200), "g1": Grp(1280, 1792, 1800)}  # matrices on 256-element chunks
        n = 2048
        a.total = n
        base = torch.arange(n, dtype=torch.float32) / 64.0
        a.grads = base * (rank + 1)                       # vector gradients (fp32, cast in at the exchange)
        a.wire = (base * (rank + 1)).to(torch.bfloat16)   # matrix gradients are born here
        a.wire[1024:1200] = 0
        a.wire[1792:1800] = 0
        a.shadow = torch.full((n,), float(rank + 1), dtype=torch.bfloat16)
        a.params = torch.full((n,), float(rank + 1))
        a.m = torch.full((n,), 10.0 * (rank + 1))
        a.v = torch.full((n,), 100.0 * (rank + 1))
        z = ShardedUpdate(a)
        ok = z.world == world and z.rank == rank and a.grad_scale == 1.0 / world
        mean = base * sum(range(1, world + 1))  # the exchange SUMS; the consumers apply arena.grad_scale = 1 / world
        # two "stages": part of g0's matrices first, the rest + everything else second
        st0, st1 = [(0, 512)], [(512, 1200), (1280, 1800)]
        h0 = z.begin(st0)
        h1 = z.begin(st1)
        z.finish(h0)
        z.finish(h1)
        ok &= z.runs == [(0, 512), (512, 1024), (1280, 1792)]
        # ... and per exchange call (= per backward stage): the staged all-gather walks these batches in REVERSE (the last
        # backward stage holds the lowest layers, which the next forward reads first; ShardedUpdate.gather_begin)
        ok &= z.batches == [[(0, 512)], [(512, 1024), (1280, 1792)]] and z.pending == []
        z.begin([])  # a stage without gradients keeps its place
        ok &= len(z.batches) == 3 and z.batches[2] == []
        z.batches.pop()
        for run in z.runs:
            o0, o1 = z.own(run)
            ok &= (o1 - o0) * world == run[1] - run[0] and o0 == run[0] + rank * (o1 - o0)
            ok &= bool(((a.wire[o0:o1].float() - mean[o0:o1]).abs() <= 8e-3 * mean[o0:o1].abs() + 1e-6).all())
        for s, e in ((1024, 1200), (1792, 1800)):  # vector ranges: averaged on every rank, from the fp32 gradients
            ok &= bool(((a.wire[s:e].float() - mean[s:e]).abs() <= 8e-3 * mean[s:e].abs() + 1e-6).all())
        own, vec = z.norm_spans(st0 + st1)
        ok &= own == [z.own(r) for r in z.runs] and vec == [(1024, 1200), (1792, 1800)]
        # the norm: own slices summed locally, one scalar all-reduced, vector ranges added by everyone
        sq = sum((a.wire[o0:o1].double() ** 2).sum() for o0, o1 in own).float().reshape(1)
        z.exchange_norm(sq)
        full = sum((a.wire[s:e].double() ** 2).sum() for s, e in z.runs)
        # (gloo path: every rank holds the whole averaged run, so the reference is computable locally)
        ok &= abs(float(sq) - float(full)) <= 1e-4 * float(full)
        # all-gather of the updated weights: every slice arrives from its owner
        for run in z.runs:
            o0, o1 = z.own(run)
            a.shadow[o0:o1] = 50.0 + rank
        ok &= z.stale is False
        z.gather()
        ok &= z.stale is True  # an update has run: the other ranks' slices of the fp32 state are out of date here
        for run in z.runs:
            c = (run[1] - run[0]) // world
            for r in range(world):
                ok &= bool((a.shadow[run[0] + r * c:run[0] + (r + 1) * c] == 50.0 + r).all())
        ok &= bool((a.shadow[1024:1280] == rank + 1).all())  # nothing outside the runs is touched
        # what a checkpoint / shadow refresh does first (ParamArena.gather_sharded_state): gather iff stale
        from xggm_amd.arena import ParamArena
        a.zero1 = z
        ParamArena.gather_sharded_state(a)
        ok &= z.stale is False
        for run in z.runs:
            c = (run[1] - run[0]) // world
            for r in range(world):
                sl = slice(run[0] + r * c, run[0] + (r + 1) * c)
                ok &= bool((a.params[sl] == r + 1).all()) and bool((a.m[sl] == 10.0 * (r + 1)).all()) and bool((a.v[sl] == 100.0 * (r + 1)).all())
        a.params[:] = -1.0
        ParamArena.gather_sharded_state(a)  # nothing updated since: no second gather
        ok &= bool((a.params == -1.0).all())
        # a rank that enters the state gather ALONE (``if rank == 0: save(state_dict())``) gets an error that
        # says what to do, not a hung group; the next symmetric call works again
        z.stale = True
        if rank == 0:
            os.environ["XGGM_COLLECTIVE_TIMEOUT"] = "1.5"
            try:
                z.gather_state()
                ok = False
            except RuntimeError as e:
                ok &= "COLLECTIVE" in str(e) and "1 of %d ranks" % world in str(e)
            os.environ.pop("XGGM_COLLECTIVE_TIMEOUT")
            ok &= z.stale is True
        else:
            z._calls = getattr(z, "_calls", 0) + 1  # (the call rank 0 made alone)
        dist.barrier()
        z.gather_state()
        ok &= z.stale is False
        z.reset()
        ok &= z.runs == []
        # the plain in-place exchange (no sharding) on the same arena
        a.wire = (base * (rank + 1)).to(torch.bfloat16)
        a.wire[1200:1280] = 0  # the alignment gap between two groups: never written, zeros on every rank
        a.wire[1800:] = 7.0
        g = GradSync(a.grads, wire_dtype=torch.bfloat16, arena=a)
        g.sync([(0, 1200), (1280, 1800)])  # one collective: the two ranges merge across the gap
        ok &= g.merged([(0, 1200), (1280, 1800), (4000, 4100)]) == [(0, 1800), (4000, 4100)]
        ok &= bool(((a.wire[:1200].float() - mean[:1200]).abs() <= 8e-3 * mean[:1200].abs() + 1e-6).all())
        ok &= bool((a.wire[1200:1280] == 0).all()) and bool((a.wire[1800:] == 7.0).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_update_is_opt_in():
    """the sharded update (ZeRO-1 + staged all-gather) has never been measured on more than one real GPU
    (SCALE_r03 is a skipped record; the one-rank RCCL run is slower with it), so ``bench.py --gpus N`` replicates the
    update at every N until a scaling record says otherwise; ``--zero1 1`` turns it on, ``--zero1 0`` off."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    assert [bench.zero1_default(w) for w in (1, 2, 4, 8)] == [False, False, False, False]
    assert bench.parse(["--zero1", "1"]).zero1 == 1 and bench.parse(["--zero1", "0"]).zero1 == 0 and bench.parse([]).zero1 is None


def test_sharded_update_bookkeeping_world2():
    """ZeRO-1 host logic over gloo (world size 2, CPU tensors): runs cut at the groups' matrix / vector borders, the
    own slices, the in-place exchange on the wire arena (matrices reduced where the GEMMs wrote them, vectors cast in
    from fp32), the scalar norm exchange, the all-gather of the updated weights and of the optimiser state."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_sharded_update_bookkeeping_world4():
    """the same worker at FOUR ranks over gloo: quarter slices, the reduce-scatter emulation, the
    scalar norm exchange, the all-gather of shadow weights and optimiser state from four owners"""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == list(range(world)) and all(ok for _, ok in res), res


def _sparse_table_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xggm_amd.dist import GradSync, ShardedUpdate

        class Grp:
            def __init__(self, start, vec_start, end):
                self.start, self.vec_start, self.end = start, vec_start, end

        ok = True
        for cls in (GradSync, ShardedUpdate):
            class Arena:
                pass
            a = Arena()
            V, H = 40, 8                     # the "word table": 40 rows of 8, inside the vector region of g0
            t0 = 1024 + 16
            a.groups = {"g0": Grp(0, 1024, 1024 + 16 + V * H + 24)}
            n = 2048
            a.total = n
            gen = torch.Generator().manual_seed(7 + rank)
            a.grads = torch.zeros(n)
            a.grads[1024:t0] = torch.randn(16, generator=gen)                 # other vectors in front of the table
            a.grads[t0 + V * H:t0 + V * H + 24] = torch.randn(24, generator=gen)  # ... and behind it
            ids = torch.tensor([[3, 7, 7, 0], [11, 3, 39, 0]]) if rank == 0 else torch.tensor([[5, 7, 20, 0], [39, 39, 2, 0]])
            tab = a.grads[t0:t0 + V * H].view(V, H)
            for i in ids.reshape(-1).tolist():  # the local scatter-add of the embedding backward (row 0 = padding too)
                tab[i] += torch.randn(H, generator=gen)
            a.wire = torch.zeros(n, dtype=torch.bfloat16)
            a.wire[:1024] = (torch.randn(1024, generator=gen)).to(torch.bfloat16)  # matrix gradients are born here
            a.shadow = torch.zeros(n, dtype=torch.bfloat16)
            a.params, a.m, a.v = torch.zeros(n), torch.zeros(n), torch.zeros(n)
            # reference: the dense exchange of the same buffers
            ref_w = a.wire.clone()
            ref_w[1024:a.groups["g0"].end] = a.grads[1024:a.groups["g0"].end].to(torch.bfloat16)
            dense = ref_w.clone()
            dist.all_reduce(dense, op=dist.ReduceOp.SUM)
            gs = cls(a) if cls is ShardedUpdate else cls(a.grads, None, torch.bfloat16, arena=a)
            gs.set_sparse_table(t0, V, H, lambda: ids)
            gs.sync([(0, a.groups["g0"].end)])
            got = a.wire
            touched = sorted(set(ids.reshape(-1).tolist()) | set([3, 7, 0, 11, 39, 5, 20, 2]))
            # touched rows: the sum of both ranks' rows (bf16 sums of two terms are exact up to one rounding);
            # untouched rows: zero; everything outside the table: as the dense exchange
            ok &= bool(torch.equal(got[:t0], dense[:t0])) and bool(torch.equal(got[t0 + V * H:], dense[t0 + V * H:]))
            gt, dt_ = got[t0:t0 + V * H].view(V, H).float(), dense[t0:t0 + V * H].view(V, H).float()
            ok &= bool(((gt - dt_).abs() <= 1e-2 * dt_.abs() + 1e-6).all())
            un = [r for r in range(V) if r not in touched]
            ok &= float(gt[un].abs().max()) == 0.0 and float(gt[touched].abs().max()) > 0.0
            # both ranks hold the same table afterwards
            both = [torch.empty_like(got) for _ in range(world)]
            dist.all_gather(both, got)
            ok &= bool(torch.equal(both[0][t0:t0 + V * H], both[1][t0:t0 + V * H]))

            def next_pass(ids_now):
                """a new backward: the table gradient holds only this pass's rows; returns the dense reference table"""
                if cls is ShardedUpdate:
                    gs.reset()
                a.grads[t0:t0 + V * H] = 0
                for i in ids_now.reshape(-1).tolist():
                    tab[i] += torch.randn(H, generator=gen)
                d = a.grads[t0:t0 + V * H].to(torch.bfloat16)
                dist.all_reduce(d, op=dist.ReduceOp.SUM)
                return d.view(V, H).float()

            # second pass, OTHER tokens: the wire's table is written only through the row path (cast_vectors skips it),
            # so the rows of the first pass must have been put back to zero
            ids2 = torch.tensor([[1, 7, 30, 0]]) if rank == 0 else torch.tensor([[31, 1, 0, 0]])
            ids = ids2
            d2 = next_pass(ids2)
            gs.sync([(0, a.groups["g0"].end)])
            gt = a.wire[t0:t0 + V * H].view(V, H).float()
            now = {0, 1, 7, 30, 31}
            ok &= bool(((gt - d2).abs() <= 1e-2 * d2.abs() + 1e-6).all())
            ok &= float(gt[[r for r in range(V) if r not in now]].abs().max()) == 0.0 and float(gt[[1, 7, 30, 31]].abs().min(1)[0].min()) >= 0.0
            ok &= float(gt[[3, 11, 39, 5, 20, 2]].abs().max()) == 0.0  # touched in the first pass only
            # third pass: the table was looked up twice since zero_grad (gradient accumulation): the last ids do not
            # cover every touched row -> the whole table goes the dense way
            a.emb_uses = 2
            ids = torch.tensor([[9, 0, 0, 0]])
            d3 = next_pass(torch.tensor([[9, 17, 25, 0]]) if rank == 0 else torch.tensor([[18, 9, 0, 0]]))
            gs.sync([(0, a.groups["g0"].end)])
            gt = a.wire[t0:t0 + V * H].view(V, H).float()
            ok &= bool(((gt - d3).abs() <= 1e-2 * d3.abs() + 1e-6).all()) and float(gt[[17, 25, 18]].abs().max()) > 0.0
            # fourth pass, sparse again: what the dense pass left in the table is cleared first
            a.emb_uses = 1
            ids = torch.tensor([[4, 0, 0, 0]]) if rank == 0 else torch.tensor([[6, 4, 0, 0]])
            d4 = next_pass(ids)
            gs.sync([(0, a.groups["g0"].end)])
            gt = a.wire[t0:t0 + V * H].view(V, H).float()
            ok &= bool(((gt - d4).abs() <= 1e-2 * d4.abs() + 1e-6).all())
            ok &= float(gt[[r for r in range(V) if r not in (0, 4, 6)]].abs().max()) == 0.0
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sparse_word_table_exchange_world2():
    """data parallelism, the word-embedding gradient: the ranks exchange the rows of the tokens of the step (ids
    all-gathered, rows gathered into a compact bf16 buffer, all-reduced, written back) instead of the whole table --
    same result as the dense exchange on the touched rows, zeros elsewhere, identical on both ranks, the ranges around
    the table untouched by the special path; for the replicated and for the sharded exchange."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_table_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
