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
        self.wire_dtype = wire_dtype
        self.bucket_elems = bucket_elems
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self._wire = None
        self._flight = None

    def sync(self, ranges):
        """average ``flat_grads[s:e]`` over ranks for every (s, e) in ``ranges``."""
        if self.world == 1:
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
    def begin(self, ranges):
        if self.world == 1 or not ranges:
            return None
        buckets = ranges_to_buckets(ranges, self.bucket_elems)
        use_avg = self.backend == "nccl"
        op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
        wire = None
        if self.wire_dtype is not None and self.wire_dtype != self.g.dtype:
            total = sum(e - s for s, e in buckets)
            # a dedicated buffer for everything in flight (nothing may be reused before ``finish``)
            if self._flight is None or self._flight.numel() < total:
                self._flight = torch.empty(total, device=self.g.device, dtype=self.wire_dtype)
            wire, o = [], 0
            for s, e in buckets:
                w = self._flight[o:o + e - s]
                w.copy_(self.g[s:e])
                wire.append(w)
                o += e - s
            works = [dist.all_reduce(w, op=op, group=self.group, async_op=True) for w in wire]
        else:
            works = [dist.all_reduce(self.g[s:e], op=op, group=self.group, async_op=True) for s, e in buckets]
        return (buckets, wire, works, use_avg)

    def finish(self, handle):
        if handle is None:
            return
        buckets, wire, works, use_avg = handle
        for i, (s, e) in enumerate(buckets):
            works[i].wait()
            if wire is not None:
                self.g[s:e].copy_(wire[i])
            if not use_avg:
                self.g[s:e].div_(self.world)


def split_ranges(arena, ranges):
    """(upper, lower) parts of the active ``ranges``: upper = everything whose gradient is final once the
    backward has passed the cut below the cross-modality layers -- the x-layer weight matrices inside
    ``enc_main`` and all groups after it (enc_tail, heads, generator); lower = the rest of ``enc_main``
    (embeddings, single-modality layers, pooler, and its vector region)."""
    main = arena.groups.get("enc_main")
    if main is None:
        return [], list(ranges)
    xs = [(o, o + k) for n, (o, k, g, atomic) in arena.info.items()
          if g == "enc_main" and not atomic and ".x_layers." in n]
    if not xs:
        return [], list(ranges)
    x0, x1 = min(a for a, _ in xs), max(b for _, b in xs)
    inside = [(o, o + k) for n, (o, k, g, atomic) in arena.info.items()
              if g == "enc_main" and not atomic and x0 <= o < x1 and ".x_layers." not in n]
    if inside:  # the x-layer matrices are not one contiguous run: do not split
        return [], list(ranges)
    upper, lower = [], []
    for s, e in ranges:
        if s >= main.end:
            upper.append((s, e))
        elif s == main.start and e == main.end:
            lower.append((s, x0))
            upper.append((x0, x1))
            lower.append((x1, e))
        else:
            lower.append((s, e))
    return [r for r in upper if r[1] > r[0]], [r for r in lower if r[1] > r[0]]


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
