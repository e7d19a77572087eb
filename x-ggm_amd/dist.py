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
import torch
import torch.distributed as dist

DEFAULT_BUCKET = 32 * 1024 * 1024  # elements (128 MB fp32 / 64 MB bf16 on the wire)


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

    def __init__(self, flat_grads, group=None, wire_dtype=None, bucket_elems=DEFAULT_BUCKET):
        self.g = flat_grads
        self.group = group
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

    def sync(self, ranges):
        """average ``flat_grads[s:e]`` over ranks for every (s, e) in ``ranges``."""
        if self.world == 1 and not self.force:
            return
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

    def region(name):
        m = re.search(r"\.x_layers\.(\d+)\.", name)
        if m:
            return 4 if (x_mid is not None and int(m.group(1)) >= x_mid) else 3
        m = re.search(r"\.(?:layer|r_layers)\.(\d+)\.", name)
        if m:
            return 2 if (pair_cut is None or int(m.group(1)) >= pair_cut) else 1
        if ".pooler." in name:
            return 4
        return 0  # embeddings, visn_fc

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
