"""Data parallelism for the X-GGM training step: one process per GPU, gradients of the flat
arena all-reduced (averaged) over RCCL/xGMI between backward and the fused clip+BertAdam.

The path is purely data parallel: samples are independent through the
whole step and every loss is a batch mean, so equal local batches + gradient AVERAGING
reproduce the single-process result.  The reference has no collective at all (only an
in-process nn.DataParallel on the encoder, src/lxrt/entry.py:183-184); this is new work, not
a port.

Because parameters live in one contiguous fp32 buffer, a "bucket" is simply a slice of it:
no packing copies.  Buckets are all-reduced asynchronously in arena order; on the wire they
can be compressed to bf16 (halves the bytes on the 7 x ~153 GB/s xGMI links, the binding
resource at 883 MB of fp32 gradients per pass).  Works on CPU tensors with the ``gloo``
backend too, which is how the logic is tested without GPUs (tests/test_dist_cpu.py).

Ranks must take the same host-side branch (relation vs node generation,
src/vqa/vqacpv2.py:192): ``sync_branch`` broadcasts rank 0's draw.
"""
import os

import torch
import torch.distributed as dist

DEFAULT_BUCKET = 32 * 1024 * 1024  # elements (128 MB fp32 / 64 MB bf16 on the wire)


class _Done:
    """work handle of a collective that has already completed"""

    def wait(self):
        return True


def all_reduce_sum(t, group, backend, world, async_op=False):
    """SUM all-reduce whose result is the SAME BITS on every rank.  RCCL gives that (every element is reduced along one
    fixed path and the result broadcast).  gloo does not once more than two ranks add floating-point numbers: its ring
    accumulates in a rank-dependent order, and the four-rank rehearsal (tools/dp_rehearsal.py, four processes on one
    GPU) found a few elements per bias / LayerNorm / embedding tensor an ulp apart between replicas after ONE update.
    gloo is this package's test transport, so there the sum is taken in rank order from an all-gather: identical on
    every rank, accumulated in fp32 and rounded once.  Two ranks: a + b == b + a, the plain all-reduce stays."""
    if backend == "nccl" or world <= 2:
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous(), group=group)
    acc = parts[0].to(torch.float32, copy=True)
    for p in parts[1:]:
        acc += p
    t.copy_(acc)
    return _Done() if async_op else None


def ranges_to_buckets(ranges, bucket_elems=DEFAULT_BUCKET):
    """split [(start, end)] element ranges into chunks of at most ``bucket_elems``."""
    out = []
    for s, e in ranges:
        while s < e:
            n = min(bucket_elems, e - s)
            out.append((s, s + n))
            s += n
    return out


class GradSync:
    """all-reduce (average) slices of a flat gradient buffer across the process group."""

    def __init__(self, flat_grads, group=None, wire_dtype=None, bucket_elems=DEFAULT_BUCKET, arena=None):
        """``arena`` with a wire buffer (ParamArena.enable_wire): the exchange runs IN PLACE on ``arena.wire`` -- the
        matrix gradients are already there (written by the weight-gradient GEMMs), only the vector ranges are cast
        in -- and nothing is copied back: the norm pass and the update read the wire."""
        self.g = flat_grads
        self.arena = arena if (arena is not None and getattr(arena, "wire", None) is not None) else None
        self.group = group
        if self.arena is not None:
            self.arena.grad_scale = 1.0 / (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # XGGM_DP_FORCE=1: issue the collectives even in a one-rank group (rehearses the RCCL stream / graph
        # interplay on a single GPU: the reduction itself is then the identity)
        import os
        self.force = bool(os.environ.get("XGGM_DP_FORCE")) and dist.is_initialized()
        self.wire_dtype = wire_dtype
        self.bucket_elems = bucket_elems
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self._wire = None
        self._flight = None

    # ---- in-place exchange on the bf16 wire arena
    def vec_parts(self, ranges):
        """the pieces of ``ranges`` that lie in a group's vector region (gradients accumulated in fp32)"""
        out = []
        for s, e in ranges:
            for G in self.arena.groups.values():
                a, b = max(s, G.vec_start), min(e, G.end)
                if b > a:
                    out.append((a, b))
        return out

    def cast_vectors(self, ranges, skip_table=False):
        """fp32 vector gradients -> the bf16 wire.  ``skip_table``: leave the word-embedding table out (94 MB read and
        47 MB written per pass for the ~640 rows that matter): its rows go on the wire through ``_finish_table``."""
        a = self.arena
        parts = self.vec_parts(ranges)
        if skip_table:
            o, e_t = self.table[0], self.table[0] + self.table[1] * self.table[2]
            parts = [q for s, e in parts for q in ((s, min(e, o)), (max(s, e_t), e)) if q[1] > q[0]]
        for s, e in parts:
            if a.wire.is_cuda:
                from . import ops
                ops.cast_bf16(a.grads[s:e], a.wire[s:e])
            else:
                a.wire[s:e].copy_(a.grads[s:e])

    # ---- the word-embedding table: only the rows some rank touched go on the links
    def set_sparse_table(self, o, rows, H, ids_of):
        """the gradient of an embedding table (arena elements [o, o + rows * H), fp32, accumulated by scatter-adds) is
        zero outside the rows of the tokens of this step: instead of all-reducing 30522 x 768 values (47 MB of bf16,
        the largest piece of the LAST, exposed stage of the exchange) the ranks all-gather their token ids (world x
        B x 20 int64), gather those rows of their own gradient into a compact bf16 buffer (world x 640 rows, 7.9 MB at
        8 ranks), all-reduce THAT, and write the summed rows back over the wire arena's table.  Identical on every
        rank (the sum is the collective's); ``ids_of()`` returns this pass's [B, T] token ids on the device."""
        self.table = (int(o), int(rows), int(H), ids_of)

    def _without_table(self, ranges):
        """ranges minus the table's; whether the table was inside"""
        t = getattr(self, "table", None)
        if t is None or self.arena is None or self.world == 1 and not self.force:
            return ranges, False
        if getattr(self.arena, "emb_uses", 1) != 1:
            # the table was looked up more than once since zero_grad() (gradient accumulation, a second forward): the ids
            # of the LAST forward do not cover every touched row -- dense exchange of the whole table
            self._table_dirty = True
            return ranges, False
        o, e = t[0], t[0] + t[1] * t[2]
        out, hit = [], False
        for s, f in ranges:
            if f <= o or s >= e:
                out.append((s, f))
                continue
            if s > o or f < e:
                self._table_dirty = True
                return ranges, False  # only part of the table in these ranges: no special path
            hit = True
            if s < o:
                out.append((s, o))
            if f > e:
                out.append((e, f))
        return out, hit

    def _table_path(self, ranges):
        """``_without_table`` + the bookkeeping of the wire's table: after a pass that went the dense way (the whole
        table was cast onto the wire) the next sparse pass clears the table once, because its row-wise zeroing only
        knows the rows of the previous SPARSE pass"""
        dense, sparse = self._without_table(ranges)
        if sparse and getattr(self, "_table_dirty", False):
            o, V, H, _ = self.table
            self.arena.wire[o:o + V * H].zero_()
            self._table_dirty, self._prev_ids = False, None
        return dense, sparse

    def _begin_table(self):
        o, V, H, ids_of = self.table
        ids = ids_of().reshape(-1).contiguous()
        n = ids.numel()
        ids_all = torch.empty(self.world * n, dtype=torch.int64, device=ids.device)
        if self.backend == "nccl":
            dist.all_gather_into_tensor(ids_all, ids, group=self.group)
        else:
            dist.all_gather([ids_all[i * n:(i + 1) * n] for i in range(self.world)], ids, group=self.group)
        a = self.arena
        g = a.grads[o:o + V * H].view(V, H)
        if g.is_cuda:
            from . import ops
            rows = ops.gather_rows(g, ids_all)
        else:
            rows = g[ids_all].to(torch.bfloat16)
        work = all_reduce_sum(rows, self.group, self.backend, self.world, async_op=True)
        return (rows, ids_all, work)

    def _finish_table(self, h):
        rows, ids_all, work = h
        work.wait()
        o, V, H, _ = self.table
        w = self.arena.wire[o:o + V * H].view(V, H)
        # the wire's table is written ONLY here (cast_vectors skips it): the rows the previous pass touched go back to
        # zero first, then this pass's summed rows land; every other row has been zero since the arena was built
        prev = getattr(self, "_prev_ids", None)
        if w.is_cuda:
            from . import ops
            if prev is not None:
                if getattr(self, "_zero_rows", None) is None or self._zero_rows.shape[0] < prev.numel():
                    self._zero_rows = torch.zeros((prev.numel(), H), device=w.device, dtype=w.dtype)
                ops.scatter_rows(self._zero_rows[:prev.numel()], prev, w)
            ops.scatter_rows(rows, ids_all, w)
        else:
            if prev is not None:
                w[prev] = 0
            w[ids_all] = rows
        self._prev_ids = ids_all

    @staticmethod
    def merged(ranges):
        """touching ranges as one -- also across the < 256-element alignment gap between two arena groups, which no
        kernel ever writes (zeros on every rank): fewer, larger collectives"""
        out = []
        for s, e in sorted(r for r in ranges if r[1] > r[0]):
            if out and 0 <= s - out[-1][1] < 256:
                out[-1] = (out[-1][0], e)
            else:
                out.append((s, e))
        return out

    def _begin_inplace(self, ranges):
        # SUM, not AVG: the norm pass and the update take the average (``arena.grad_scale`` = 1 / world) on the way
        # in.  Same bytes on the links and no pre-multiply kernel (on a one-rank group RCCL then launches nothing).
        dense, sparse = self._table_path(ranges)
        self.cast_vectors(ranges, skip_table=sparse)
        w = self.arena.wire
        tab = self._begin_table() if sparse else None
        rs = self.merged(dense)
        works, staged = [], []
        for s, e in rs:
            wk, st = self._all_reduce_in_place(w[s:e])
            works.append(wk)
            if st is not None:
                staged.append(st)
        return ("inplace", rs, works, True, tab, staged)

    def _all_reduce_in_place(self, t):
        """SUM all-reduce of the slice ``t`` where it lies; returns (work, staging pair or None).  gloo on DEVICE tensors
        (tools/dp_rehearsal.py) reduces a copy and lets ``_finish_inplace`` bring the result back with a device copy on the
        current stream, behind the test hook's delay: the host blocks inside a gloo collective, so reduced in place the sums
        would always be there before the next graph is launched and an update that forgot to wait for the exchange (or an
        exchange that started before the backward stage had written its gradients) could not be caught."""
        if self.backend != "nccl" and t.is_cuda:
            st = t.clone()
            return all_reduce_sum(st, self.group, self.backend, self.world, async_op=True), (t, st)
        return all_reduce_sum(t, self.group, self.backend, self.world, async_op=True), None

    def _finish_inplace(self, handle):
        for wk in handle[2]:
            wk.wait()
        staged = handle[5] if len(handle) > 5 else []
        if staged:
            delay = int(float(os.environ.get("XGGM_GATHER_DELAY_US", "0")) * 2000)  # the rehearsal's delay hook
            if delay > 0:
                torch.cuda._sleep(delay)
            for dst, st in staged:
                dst.copy_(st, non_blocking=True)
        if len(handle) > 4 and handle[4] is not None:
            self._finish_table(handle[4])

    def sync(self, ranges):
        """average ``flat_grads[s:e]`` over ranks for every (s, e) in ``ranges``."""
        if self.world == 1 and not self.force:
            if self.arena is not None:
                self.cast_vectors(ranges)  # the update reads the wire
            return
        if self.arena is not None:
            return self._finish_inplace(self._begin_inplace(ranges))
        buckets = ranges_to_buckets(ranges, self.bucket_elems)
        use_avg = self.backend == "nccl"
        op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
        if self.wire_dtype is not None and self.wire_dtype != self.g.dtype:
            # two wire buffers: the cast of bucket i+1 and the copy-back of bucket i-1 run on the compute
            # stream while bucket i is on the links (RCCL's stream); a buffer is reused only after the
            # copy-back of its previous occupant has been enqueued behind that bucket's completion
            if self._wire is None or self._wire[0].numel() < self.bucket_elems:
                self._wire = [torch.empty(self.bucket_elems, device=self.g.device, dtype=self.wire_dtype)
                              for _ in range(2)]

            def finish(i):
                s, e = buckets[i]
                works[i].wait()
                self.g[s:e].copy_(self._wire[i % 2][:e - s])
                if not use_avg:
                    self.g[s:e].div_(self.world)

            works = []
            for i, (s, e) in enumerate(buckets):
                if i >= 2:
                    finish(i - 2)  # frees wire[i % 2]
                w = self._wire[i % 2][:e - s]
                w.copy_(self.g[s:e])
                works.append(dist.all_reduce(w, op=op, group=self.group, async_op=True))
            for i in range(max(0, len(buckets) - 2), len(buckets)):
                finish(i)
            return
        works = [dist.all_reduce(self.g[s:e], op=op, group=self.group, async_op=True) for s, e in buckets]
        for w in works:
            w.wait()
        if not use_avg:
            for s, e in buckets:
                self.g[s:e].div_(self.world)


    # ---- split form: ``begin`` puts the ranges on the wire and returns at once, ``finish`` waits and writes back
    def begin(self, ranges, slot=0):
        if self.arena is not None and ranges:
            if self.world == 1 and not self.force:
                self.cast_vectors(ranges)
                return None
            return self._begin_inplace(ranges)
        if (self.world == 1 and not self.force) or not ranges:
            return None
        buckets = ranges_to_buckets(ranges, self.bucket_elems)
        use_avg = self.backend == "nccl"
        op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
        wire = None
        if self.wire_dtype is not None and self.wire_dtype != self.g.dtype:
            total = sum(e - s for s, e in buckets)
            # a dedicated buffer per exchange in flight (nothing may be reused before its ``finish``)
            if self._flight is None:
                self._flight = {}
            buf = self._flight.get(slot)
            if buf is None or buf.numel() < total:
                buf = self._flight[slot] = torch.empty(total, device=self.g.device, dtype=self.wire_dtype)
            wire, o = [], 0
            for s, e in buckets:
                w = buf[o:o + e - s]
                w.copy_(self.g[s:e])
                wire.append(w)
                o += e - s
            # the ranges of a stage are scattered over the gradient buffer (17 pieces per pass in the full model) but
            # contiguous on the wire: ONE collective per stage buffer (split only above 4 x bucket_elems) instead
            # of one per piece -- fewer, larger transfers for the point-to-point links
            step = 4 * self.bucket_elems
            works = [dist.all_reduce(buf[a:min(a + step, total)], op=op, group=self.group, async_op=True)
                     for a in range(0, total, step)]
        else:
            works = [dist.all_reduce(self.g[s:e], op=op, group=self.group, async_op=True) for s, e in buckets]
        return (buckets, wire, works, use_avg)

    def finish(self, handle):
        if handle is None:
            return
        if handle[0] == "inplace":
            return self._finish_inplace(handle)
        buckets, wire, works, use_avg = handle
        if wire is not None:  # collectives cover the whole stage buffer, not single buckets
            for w in works:
                w.wait()
        for i, (s, e) in enumerate(buckets):
            if wire is None:
                works[i].wait()
            if wire is not None:
                self.g[s:e].copy_(wire[i])
            if not use_avg:
                self.g[s:e].div_(self.world)


class ShardedUpdate(GradSync):
    """ZeRO-1 on the wire arena: the update of the MATRIX ranges is sharded over the data-parallel ranks.

        backward (per stage)   reduce-scatter of the stage's matrix runs: rank r keeps the averaged r-th slice of
                               every run (in place: the slice of the wire it already holds); all-reduce of the vector
                               runs (biases, LayerNorm, embedding tables: 12 % of the parameters, updated everywhere)
        norm                   sum of squares over the own slices -> all-reduce of ONE scalar -> + the vector ranges
        update                 BertAdam on the own slices (1 / world of p, m, v) and on the vector ranges
        after the update       all-gather of the bf16 shadow weights of the matrix runs (what the GEMMs read)

    Same bytes on the links as the all-reduce it replaces (reduce-scatter + all-gather), 1 / world of the 30 B per
    parameter the update streams through HBM -- the step is update-bound at 32 samples per GPU.
    The fp32 masters and moments of the other ranks' slices go stale: ``gather_state`` refreshes them before a
    checkpoint.  Runs are contiguous pieces of the gradient buffer whose length divides by the world size (matrices
    start on 256-element chunks, arena.ALIGN_MAT); the slices of one pass are recorded at the exchange and reused by
    the norm, the update and the gather."""

    def __init__(self, arena, group=None):
        if getattr(arena, "wire", None) is None:
            raise RuntimeError("the sharded update runs on the bf16 wire arena (ParamArena.enable_wire)")
        super().__init__(arena.grads, group, torch.bfloat16, arena=arena)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if any(256 % w for w in (self.world,)) and self.world > 1:
            raise RuntimeError("sharded update: world size %d must divide 256" % self.world)
        self.runs = []  # matrix runs (start, end) exchanged in this pass, in exchange order
        self.batches = []  # the same runs per exchange call (= per backward stage, stage 0 first): gather_begin
        self.pending = []  # events of all-gathers that are still running beside the next forward (gather_begin)
        self.all_runs = set()  # every run any pass has exchanged (gather_state)
        self.stale = False  # a sharded update has run since the last gather_state: the other ranks' slices of the
        # fp32 masters / moments on THIS rank are out of date (only the bf16 shadow is gathered after an update)

    def reset(self):
        self.runs = []
        self.batches = []

    def split(self, ranges):
        """ranges -> (matrix runs, vector runs): cut at the groups' matrix / vector borders"""
        mats, vecs = [], []
        for s, e in ranges:
            for G in self.arena.groups.values():
                a, b = max(s, G.start), min(e, G.vec_start)
                if b > a:
                    mats.append((a, b))
                a, b = max(s, G.vec_start), min(e, G.end)
                if b > a:
                    vecs.append((a, b))
        return mats, vecs

    def own(self, run):
        a, b = run
        c = (b - a) // self.world
        if c * self.world != b - a:
            raise RuntimeError("sharded update: run [%d, %d) does not divide by %d ranks" % (a, b, self.world))
        return a + self.rank * c, a + (self.rank + 1) * c

    def _begin_inplace(self, ranges):
        dense, sparse = self._table_path(ranges)
        self.cast_vectors(ranges, skip_table=sparse)
        w = self.arena.wire
        nccl = self.backend == "nccl"
        op = dist.ReduceOp.SUM  # the average is taken by the norm pass and the update (arena.grad_scale)
        tab = self._begin_table() if sparse else None
        mats, vecs = self.split(dense)
        self.batches.append(list(mats))
        works, staged = [], []
        for run in mats:
            self.runs.append(run)
            self.all_runs.add(run)
            a, b = run
            if nccl:  # in place: the output is this rank's slice of the input
                o0, o1 = self.own(run)
                works.append(dist.reduce_scatter_tensor(w[o0:o1], w[a:b], op=op, group=self.group, async_op=True))
            else:     # gloo has no reduce-scatter: all-reduce, every rank then uses its slice only
                wk, st = self._all_reduce_in_place(w[a:b])
                works.append(wk)
                if st is not None:
                    staged.append(st)
        for s, e in vecs:
            wk, st = self._all_reduce_in_place(w[s:e])
            works.append(wk)
            if st is not None:
                staged.append(st)
        return ("inplace", mats + vecs, works, nccl, tab, staged)

    def begin(self, ranges, slot=0):
        if not ranges:
            self.batches.append([])  # a stage without gradients still counts: batches[k] belongs to backward stage k
            return None
        if self.world == 1 and not self.force:
            self.cast_vectors(ranges)
            mats = self.split(ranges)[0]
            self.runs += mats
            self.batches.append(list(mats))
            self.all_runs.update(mats)
            return None
        return self._begin_inplace(ranges)

    def sync(self, ranges):
        self.finish(self.begin(ranges))

    # ---- after the exchange
    def norm_spans(self, ranges):
        """(spans summed locally and all-reduced, spans every rank sums itself)"""
        return [self.own(r) for r in self.runs], self.split(ranges)[1]

    def exchange_norm(self, sq):
        if self.world > 1 or self.force:
            all_reduce_sum(sq, self.group, self.backend, self.world)
            f8 = getattr(self.arena, "fp8", None)
            if f8 is not None:
                # fp8 forward: every rank quantises only its slices of the weights and records their maxima; the scale of
                # an operand needs the maximum over ALL its slices -- one small MAX all-reduce of the weight entries of the
                # scale table, once per pass, in front of the update that derives the new scales from it
                dist.all_reduce(f8.weight_amax(), op=dist.ReduceOp.MAX, group=self.group)

    def _gather_runs(self, runs, defer=None):
        """queue the all-gathers of ``runs`` (bf16 shadow and, with the fp8 forward, the e4m3 copies the owners wrote in
        the same update) on the current stream's side of the communicator; returns the work handles.  gloo on DEVICE
        tensors (tools/dp_rehearsal.py: two ranks on one GPU) gathers into staging buffers: the host blocks inside a gloo
        collective, so gathered straight into the shadow the bytes would always be in place before the next graph is
        launched and a consumer that forgot to wait for its batch could never be caught; the staged slices reach the shadow
        by device copies -- right away, or, with ``defer`` (a list that receives (destination, staging) pairs), when
        ``gather_begin`` queues them behind the test hook's delay: asynchronous as the nccl path is."""
        f8 = getattr(self.arena, "fp8", None)
        bufs = [self.arena.shadow] + ([f8.shadow8] if f8 is not None else [])
        works = []
        for sh in bufs:
            for run in runs:
                a, b = run
                o0, o1 = self.own(run)
                c = (b - a) // self.world
                if self.backend == "nccl":
                    works.append(dist.all_gather_into_tensor(sh[a:b], sh[o0:o1], group=self.group, async_op=True))
                elif not sh.is_cuda:
                    works.append(dist.all_gather([sh[a + i * c:a + (i + 1) * c] for i in range(self.world)], sh[o0:o1].clone(),
                                                 group=self.group, async_op=True))
                else:
                    stage = torch.empty(b - a, dtype=sh.dtype, device=sh.device)
                    dist.all_gather([stage[i * c:(i + 1) * c] for i in range(self.world)], sh[o0:o1].clone(), group=self.group)
                    if defer is not None:
                        defer.append((sh[a:b], stage))
                    else:
                        sh[a:b].copy_(stage, non_blocking=True)
        return works

    def gather(self):
        """all-gather the bf16 shadow weights of this pass's matrix runs from their owners"""
        self.stale = True
        if self.world == 1 and not self.force:
            return
        for wk in self._gather_runs(self.runs):
            wk.wait()

    def gather_begin(self, comm):
        """The same gather, stage by stage in the order the NEXT forward reads the weights, on the stream ``comm``: the
        runs of the LAST backward stage (the lowest layers) first.  Leaves one event per batch in ``pending`` (forward
        order: pending[i] covers what forward stage i reads); the consumer makes its stream wait for pending[i] right
        before stage i (engine.CapturedTrainer replays one forward graph per stage), so the gather of stage i + 1 runs
        under the forward of stage i.  Anything else that reads the shadow weights calls ``wait_pending`` first
        (runtime.runtime_of does, for every eager forward)."""
        self.stale = True
        self.pending = []
        if self.world == 1 and not self.force:
            return
        comm.wait_stream(torch.cuda.current_stream())  # the update that wrote the own slices
        # test hook (tools/dp_rehearsal.py): hold every batch back by so many microseconds -- a consumer that does not
        # wait for its batch then reads the weights of the previous step and the rehearsal's bit-exactness checks fail
        delay = int(float(os.environ.get("XGGM_GATHER_DELAY_US", "0")) * 2000)  # ~2 GHz shader clock
        staged = self.backend != "nccl" and self.arena.shadow.is_cuda
        with torch.cuda.stream(comm):
            if staged:  # every (host-blocking) gloo collective first, then the device side batch by batch
                parts = []
                for runs in reversed(self.batches):
                    d = []
                    self._gather_runs(runs, defer=d)
                    parts.append(d)
            for i, runs in enumerate(reversed(self.batches)):
                if delay > 0:
                    torch.cuda._sleep(delay)
                if staged:
                    for dst, st in parts[i]:
                        dst.copy_(st, non_blocking=True)
                else:
                    for wk in self._gather_runs(runs):
                        wk.wait()  # (stream-side: `comm` waits for the communicator's stream, the host does not block on nccl)
                ev = torch.cuda.Event()
                ev.record(comm)
                self.pending.append(ev)

    def take_pending(self):
        ev, self.pending = self.pending, []
        return ev

    def wait_pending(self):
        """the current stream waits for every all-gather ``gather_begin`` left running"""
        for ev in self.take_pending():
            torch.cuda.current_stream().wait_event(ev)

    def _rendezvous(self, what, timeout=None):
        """every rank of the group has reached the same collective call, or RuntimeError after ``timeout`` seconds
        (XGGM_COLLECTIVE_TIMEOUT, default 120): host side, through the default store, whatever the backend"""
        import time
        timeout = float(os.environ.get("XGGM_COLLECTIVE_TIMEOUT", "120")) if timeout is None else float(timeout)
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:
            return  # no store to ask (not the default initialisation): the collective is entered as it always was
        ranks = dist.get_process_group_ranks(self.group) if self.group is not None else list(range(dist.get_world_size()))
        self._calls = getattr(self, "_calls", 0) + 1
        key = "xggm/%s/%s/%d" % (what, "-".join(str(r) for r in ranks), self._calls)
        n = store.add(key, 1)
        t0 = time.time()
        while n < self.world:
            if time.time() - t0 > timeout:
                raise RuntimeError(
                    "sharded update: only %d of %d ranks reached %s (call %d) within %.0f s.  Under the sharded update the "
                    "fp32 masters and moments are spread over the ranks, so model.state_dict(), BertAdam.state_dict(), "
                    "save_training_state(), VQA.save() and sync_weights() are COLLECTIVE calls: make them on every rank "
                    "(and keep rank 0's result), not inside `if rank == 0:`" % (n, self.world, what, self._calls, timeout))
            time.sleep(0.005)
            n = store.add(key, 0)

    @torch.no_grad()
    def gather_state(self):
        """fp32 masters and BertAdam moments of EVERY matrix range from their owners (before state_dict / a
        checkpoint), run by run as the passes exchanged them.  A COLLECTIVE: every rank of the group calls it (through
        ``ParamArena.gather_sharded_state``: model.state_dict(), BertAdam.state_dict(), save_training_state, VQA.save
        and sync_shadow all do, so under the sharded update those are collective calls too -- as a sharded framework's
        state_dict is)."""
        if self.world == 1:
            self.stale = False
            return
        # the usual ``if rank == 0: torch.save(model.state_dict())`` enters this collective on ONE rank and used
        # to hang the group without a word.  A host-side rendezvous through the process group's store first: every rank
        # counts itself in under the same (call number) key and waits for the others with a timeout.
        self._rendezvous("gather_state")
        self.stale = False
        for run in sorted(self.all_runs):
            a, b = run
            o0, o1 = self.own(run)
            c = (b - a) // self.world
            for buf in (self.arena.params, self.arena.m, self.arena.v):
                if self.backend == "nccl":
                    dist.all_gather_into_tensor(buf[a:b], buf[o0:o1], group=self.group)
                else:
                    dist.all_gather([buf[a + i * c:a + (i + 1) * c] for i in range(self.world)], buf[o0:o1].clone(),
                                    group=self.group)


def stage_ranges(arena, ranges, layout, n_stages):
    """the active ``ranges`` of the flat gradient buffer split by backward stage: element k of the result holds
    the ranges whose gradients are FINAL once stage k of Runtime.backward has run (stage 0 = above the last cut).
    ``layout`` = Runtime.cut_layout (where LXRTEncoder.forward cuts), ``n_stages`` = cuts recorded + 1.
    Forward regions of ``enc_main``: 0 = embeddings and visn_fc; 1 = layer pairs below ``pair_cut``; 2 = the other
    single-modality layers; 3 = cross layers below ``x_mid``; 4 = the remaining cross layers and the pooler.  A cut
    sits between regions 0|1 (``emb_cut``), 1|2 (``pair_cut``), 2|3 (always), 3|4 (``x_mid``); regions without a cut
    between them are final together.  Every other arena group (enc_tail, heads, generator) is final after stage 0;
    the vector region of ``enc_main`` (biases, LayerNorm parameters, embedding tables of ALL layers) only after
    the last stage."""
    import re
    last = n_stages - 1
    main = arena.groups.get("enc_main")
    pair_cut, x_mid, emb_cut = layout.get("pair_cut"), layout.get("x_mid"), layout.get("emb_cut", False)
    # which cuts exist (in forward order) decides how regions map to stages: borders[i] = first region above cut i
    borders = []
    if emb_cut:
        borders.append(1)
    if pair_cut is not None:
        borders.append(2)
    borders.append(3)
    if x_mid is not None:
        borders.append(4)
    if len(borders) + 1 != n_stages:  # the forward recorded other cuts than the layout predicts: no split
        return [[] for _ in range(last)] + [list(ranges)]

    from .arena import region_of

    def region(name):
        return region_of(name, pair_cut, x_mid)

    def stage_of_region(r):
        return sum(1 for b in borders if b > r)  # cuts after the region = stages that run before it is final

    out = [[] for _ in range(n_stages)]
    for s, e in ranges:
        if main is None or s >= main.end or e <= main.start:
            out[0].append((s, e))
            continue
        # inside enc_main: matrices by region (runs of equal stage, alignment gaps included), vectors last
        items = sorted((o, k, n) for n, (o, k, g, atomic) in arena.info.items() if g == "enc_main" and not atomic)
        run_s, run_stage = None, None
        for idx, (o, k, n) in enumerate(items):
            st = min(stage_of_region(region(n)), last)
            if run_stage is None:
                run_s, run_stage = max(o, s), st
            elif st != run_stage:
                out[run_stage].append((run_s, o))
                run_s, run_stage = o, st
        if run_stage is not None:
            out[run_stage].append((run_s, min(main.vec_start, e)))
        if e > main.vec_start:
            out[last].append((max(main.vec_start, s), e))
    return [[r for r in rs if r[1] > r[0]] for rs in out]


def sync_branch(branch_is_rel, device, group=None):
    """make every rank take rank 0's host-side branch decision."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(branch_is_rel)
    t = torch.tensor([1 if branch_is_rel else 0], device=device, dtype=torch.int32)
    dist.broadcast(t, src=0, group=group)
    return bool(t.item())


def shard_batch(batch, rank, world):
    """equal contiguous shards of every per-sample tensor of a global batch."""
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            n = v.shape[0]
            if n % world:
                raise ValueError("global batch %d is not divisible by world size %d" % (n, world))
            per = n // world
            out[k] = v[rank * per:(rank + 1) * per]
        else:
            out[k] = v
    return out


def active_ranges(arena):
    """(start, end) of the arena groups that received gradients in this pass."""
    return [(arena.groups[g].start, arena.groups[g].end) for g in arena.active_groups()]


def broadcast_params(arena, group=None):
    """rank 0's parameters to everyone (identical replicas at step 0)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(arena.params, src=0, group=group)
        arena.sync_shadow()
