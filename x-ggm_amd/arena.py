"""Flat parameter arena: every parameter of the model is a view into ONE fp32 buffer, with
parallel flat buffers for gradients, BertAdam moments and (bf16 mode) shadow weights.

Why (MI355X-first): at 32 samples per GPU the step is bound by streaming 220.8 M parameters
through the optimiser.  Flat buffers let grad-norm + BertAdam + shadow
cast run as a handful of full-bandwidth launches instead of 468 x 5 tiny ones, let the
data-parallel all-reduce work on large contiguous buckets without packing copies, and let
Q/K/V weights sit adjacently so one GEMM computes the fused projection.

Layout: parameters are grouped by (optimiser group, usage) -- ``enc_main``, ``enc_tail`` (the
last cross layer's visual self-attention/FFN, which the plain-VQA pass never reaches),
``logit_fc``, ``generator``, ``encoder_adj``, ``node_fc``, ``fusion_fc`` -- because the
reference's BertAdam skips parameters whose grad is None (src/lxrt/optimization.py:118-119)
and which parameters those are depends on the pass.  Inside a group, matrices come first
(their wgrad GEMM overwrites), then the "atomic" tensors (vectors, embedding tables,
box_fc.weight, GIN eps) whose gradients are accumulated with atomics and must be zeroed
once per pass.
"""
import torch

from . import ops

ALIGN = 8        # elements: 32 B fp32 / 16 B bf16
ALIGN_MAT = 256  # matrices start on 256-element chunks: the unit of the e4m3 scale-id table (fp8 forward) and of the
                 # shard boundaries of a sharded update (any run of whole matrices divides by 2, 4, 8 ranks)


def _align(n, a=ALIGN):
    return (n + a - 1) // a * a


def region_of(name, pair_cut, x_mid):
    """forward region of an encoder parameter, as the data-parallel backward stages see it (LXRTEncoder.forward cuts
    the autograd graph between them): 0 = embeddings and visn_fc; 1 = layer pairs below ``pair_cut``; 2 = the other
    single-modality layers; 3 = cross layers below ``x_mid``; 4 = the remaining cross layers and the pooler."""
    import re
    m = re.search(r"\.x_layers\.(\d+)\.", name)
    if m:
        return 4 if (x_mid is not None and int(m.group(1)) >= x_mid) else 3
    m = re.search(r"\.(?:layer|r_layers)\.(\d+)\.", name)
    if m:
        return 2 if (pair_cut is None or int(m.group(1)) >= pair_cut) else 1
    if ".pooler." in name:
        return 4
    return 0


def encoder_cuts(names):
    """(pair_cut, x_mid) of the encoder whose parameter names are given: the rule of LXRTEncoder.forward"""
    import re

    def count(tag):
        idx = [int(m.group(1)) for n in names for m in [re.search(r"\.%s\.(\d+)\." % tag, n)] if m]
        return max(idx) + 1 if idx else 0

    n_pair = min(count("layer"), count("r_layers"))
    nx = count("x_layers")
    return (2 if n_pair >= 4 else None), (nx - 2 if nx >= 4 else None)


def layout(named, group_of, model=None):
    """offsets of every parameter in the flat buffers.  ``named``: [(name, p)] with p.dim() / p.numel() / p.shape.
    -> (group order, {group: Group}, {name: (offset, numel, group, atomic)}, total elements).
    Inside a group the matrices are laid out by DESCENDING backward-stage region (stable within a region), the
    vectors behind them: the gradients that become final together in a staged data-parallel backward are then ONE
    contiguous run of the buffer (one collective per stage instead of one per scattered piece), and the region that
    is final last sits next to the vector region, which is final last too."""
    pair_cut, x_mid = encoder_cuts([n for n, _ in named])
    order = []
    for n, p in named:
        g = group_of(n, model)
        if g not in order:
            order.append(g)
    groups = {g: Group(g) for g in order}
    info, off = {}, 0
    for gname in order:
        G = groups[gname]
        G.start = off = _align(off, ALIGN_MAT)
        members = [(n, p) for n, p in named if group_of(n, model) == gname]
        mats = [(n, p) for n, p in members if not is_atomic(n, p)]
        mats.sort(key=lambda t: -region_of(t[0], pair_cut, x_mid))  # stable: modules keep their parameter order
        for n, p in mats:
            info[n] = (off, p.numel(), gname, False)
            off = _align(off + p.numel(), ALIGN_MAT)
        G.vec_start = off
        for n, p in members:
            if is_atomic(n, p):
                info[n] = (off, p.numel(), gname, True)
                off = _align(off + p.numel())
        G.end = off
        G.params = [p for _, p in members]
    return order, groups, info, _align(off, ALIGN_MAT)


def default_group_of(name, model=None):
    if name.startswith("lxrt_encoder."):
        tail = getattr(model, "_enc_tail_prefixes", ())
        return "enc_tail" if name.startswith(tail) and tail else "enc_main"
    return name.split(".")[0]


def is_atomic(name, p):
    return p.dim() < 2 or "embeddings." in name or name.endswith("box_fc.weight") or p.numel() < 64


class Group:
    def __init__(self, name):
        self.name = name
        self.start = self.vec_start = self.end = 0
        self.params = []


class ParamArena:
    def __init__(self, model, compute_dtype, group_of=default_group_of):
        named = [(n, p) for n, p in model.named_parameters()]
        if not named:
            raise ValueError("model has no parameters")
        dev = named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("xggm_amd: parameters must be on the GPU before the first forward "
                               "(no CPU fallback); call model.cuda()")
        self.device = dev
        self.compute_dtype = compute_dtype
        order, self.groups, self.info, self.total = layout(named, group_of, model)
        self.params = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.shadow = (torch.zeros(self.total, device=dev, dtype=torch.bfloat16)
                       if compute_dtype == torch.bfloat16 else None)
        self.fp8 = None  # xggm_amd.fp8.Fp8State when the forward QKV / FFN products run on e4m3 operands
        # Data parallelism with bf16 on the wire: the weight-gradient GEMMs write MATRIX gradients straight into this
        # bf16 mirror of the gradient buffer (same offsets), RCCL reduces it in place and the update reads it -- no
        # fp32 copy of those gradients, no cast before and no copy after the exchange.  Vector gradients (biases,
        # LayerNorm, embedding tables: accumulated with fp32 atomics) stay in ``grads`` and are cast into their
        # ranges of the wire when the exchange starts.  ``p.grad`` of a matrix is then only a presence marker.
        self.wire = None
        self.grad_scale = 1.0  # 1 / world when the wire holds SUMS over the data-parallel ranks
        # device scalars: per-group step counter + schedule value, global sum of squares
        self.steps = torch.zeros(len(order), device=dev, dtype=torch.int64)
        self.lr_scale = torch.ones(len(order), device=dev, dtype=torch.float32)
        self.sqnorm = torch.zeros(1, device=dev, dtype=torch.float32)
        # learning rate of each group as the update kernels read it (BertAdam.sync_hyper keeps it equal to
        # param_groups[i]['lr']: an edit between replays of a captured pass takes effect)
        self.lr_table = torch.zeros(len(order), device=dev, dtype=torch.float32)
        self.lr_host = [None] * len(order)
        self.group_index = {g: i for i, g in enumerate(order)}
        self.pending_clip = None  # max_norm registered by clip_grad_norm_, consumed by BertAdam.step
        self.emb_uses = 0
        self.zero1 = None         # dist.ShardedUpdate when the update is sharded over data-parallel ranks
        self.touched = set()
        self.vec_zeroed = False
        self.all_dirty = True
        self._ptrs = {}
        with torch.no_grad():
            for n, p in named:
                o, k, gname, atomic = self.info[n]
                view = self.params[o:o + k].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = None
                p._xg = (self, o, k, gname, atomic, n)
                self._ptrs[n] = view.data_ptr()
        # Gradient-norm slots: the weight-gradient GEMMs of the big encoder groups leave the sum of squares of every
        # 64 x 64 block they store in ``sq_slots`` (ops.p_wgrad), so clip_grad_norm_ does not have to read those
        # gradients again (207 M of the 221 M parameters, 0.83 GB per pass).  bf16 mode only (the fp32 kernels
        # have no such epilogue); switched off when gradients are exchanged between ranks first.
        self.sq_enabled = compute_dtype == torch.bfloat16
        self.sq_base, self.sq_range, n_slots = {}, {}, 0
        for gname in order:
            if gname not in ("enc_main", "enc_tail"):
                continue
            n_slots = (n_slots + 3) // 4 * 4  # a group's slots start float4-aligned (xggm_sqnorm_multi_f32 ranges)
            first = n_slots
            for n, p in named:
                o, k, g_, atomic = self.info[n]
                if g_ == gname and not atomic and p.dim() == 2 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0:
                    self.sq_base[n] = n_slots
                    n_slots += k // 4096
            self.sq_range[gname] = (first, n_slots)
        # ... and behind them ROW_LIST_CAP slots for the word table's gradient rows (RowList below)
        self.row_sq0 = _align(max(n_slots, 1))
        self.sq_slots = torch.zeros(self.row_sq0 + ROW_LIST_CAP, device=dev, dtype=torch.float32)
        self.row_list = None       # RowList of the word-embedding table, made at its first backward
        self.row_list_enabled = True  # off when gradients are exchanged between ranks (other ranks' rows arrive)
        self.sq_covered = set()  # names whose gradient of this pass is accounted for in the slots
        self.sq_clean = False    # slots zeroed since the last zero_grad()
        self.named = dict(named)
        self._probe = [named[0][0], named[len(named) // 2][0], named[-1][0]]
        self._atomic_params = [p for _, p in named if p._xg[4]]
        self.sync_shadow()

    # ------------------------------------------------------------------ validity / views
    def valid(self):
        """False when someone re-allocated the parameters (e.g. model.cuda() afterwards)."""
        return all(self.named[n].data_ptr() == self._ptrs[n] for n in self._probe)

    def gather_sharded_state(self):
        """ZeRO-1 (dist.ShardedUpdate): after a sharded update every rank holds fresh fp32 masters and moments only for
        the slices it owns.  Anything that READS the masters as a whole -- a checkpoint, a shadow refresh -- first
        brings the other ranks' slices over.  Collective when it has something to do (every rank must get here); a
        no-op without the sharded update or when nothing has been updated since the last gather."""
        z = self.zero1
        if z is not None:
            z.wait_pending()  # all-gathers of the bf16 weights still running beside a forward (gather_begin)
        if z is not None and z.stale:
            z.gather_state()

    def sync_shadow(self):
        """refresh the bf16 shadow weights from the fp32 masters (after init / load_state_dict).  Under the sharded
        update the masters are made whole first (after a load_state_dict that every rank ran, the owners' slices hold
        the loaded values, so the gather hands out exactly those; tensors the load did not cover get their owners'
        fresh values instead of this rank's stale ones)."""
        self.gather_sharded_state()  # never build the shadow of all weights from stale slices
        if self.shadow is not None:
            ops.cast_bf16(self.params, self.shadow)
        if self.fp8 is not None:
            self.fp8.requantize_weights()

    def w(self, p):
        """the tensor the GEMMs consume for parameter ``p``: bf16 shadow or the fp32 master."""
        if self.shadow is None:
            return p.data
        _, o, k, _, _, _ = p._xg
        return self.shadow[o:o + k].view(p.shape)

    def after(self, p, n, use8=False, skip=0):
        """the (up to) ``n`` weight elements that FOLLOW parameter ``p`` in the buffer the products read (bf16 shadow, or
        the e4m3 copies with ``use8``), inside its group's matrix region: matrices are laid out in forward order within a
        backward-stage region, so this is what the next products of the forward will ask for (``ops.prefetch_next``).
        None without a shadow (fp32 mode) or at the end of the region."""
        buf = (self.fp8.shadow8 if (use8 and self.fp8 is not None) else self.shadow)
        if buf is None:
            return None
        _, o, k, g, _, _ = p._xg
        a = o + k + int(skip)
        b = min(a + int(n), self.groups[g].vec_start)
        return buf[a:b] if b > a else None

    def before(self, p, n):
        """the (up to) ``n`` bf16 weight elements in FRONT of parameter ``p`` inside its group's matrix region: what the
        dgrads of the backward ask for next"""
        if self.shadow is None:
            return None
        _, o, k, g, _, _ = p._xg
        a = max(o - int(n), self.groups[g].start)
        return self.shadow[a:o] if o > a else None

    def fused(self, ps, compute=True):
        """one [sum(rows), cols] view over parameters that are adjacent in the arena
        (query/key/value weights or biases)."""
        o0 = ps[0]._xg[1]
        k = 0
        for p in ps:
            if p._xg[1] != o0 + k:
                raise RuntimeError("parameters are not adjacent in the arena")
            k += p.numel()
            if p.numel() % ALIGN:
                raise RuntimeError("cannot fuse parameters whose size is not a multiple of %d" % ALIGN)
        buf = self.shadow if (compute and self.shadow is not None and ps[0].dim() >= 2) else self.params
        if ps[0].dim() >= 2:
            return buf[o0:o0 + k].view(-1, ps[0].shape[1])
        return buf[o0:o0 + k]

    def enable_wire(self):
        if self.shadow is None:
            raise RuntimeError("the bf16 wire arena needs bf16 storage")
        if self.wire is None:
            self.wire = torch.zeros(self.total, device=self.device, dtype=torch.bfloat16)
        self.sq_enabled = False
        self.row_list_enabled = False
        return self.wire

    def grad_view(self, ps, wire=False):
        if not isinstance(ps, (list, tuple)):
            ps = [ps]
        o0 = ps[0]._xg[1]
        k = 0
        for p in ps:
            if p._xg[1] != o0 + k:
                raise RuntimeError("parameters are not adjacent in the arena")
            k += p.numel()
        v = (self.wire if wire else self.grads)[o0:o0 + k]
        return v.view(-1, ps[0].shape[1]) if ps[0].dim() >= 2 else v

    # ------------------------------------------------------------------ gradient bookkeeping
    # torch semantics are kept: a parameter whose ``.grad`` is None gets its gradient WRITTEN
    # (and ``.grad`` published as a view of the flat buffer); one whose ``.grad`` exists gets
    # the new contribution ADDED (shared weights, or no zero_grad() between two backwards).
    def begin_pass(self):
        """called by zero_grad(): every ``.grad`` is None again, so the atomically accumulated
        ranges can be cleared with one fill per group at the next backward."""
        self.vec_zeroed = False
        self.emb_uses = 0  # look-ups of the word table since zero_grad() (dist.GradSync: the sparse exchange needs exactly one)
        if self.row_list is not None:
            self.row_list.listed = 0
        self.pending_clip = None
        self.sq_covered.clear()
        self.sq_clean = False
        if self.zero1 is not None:
            self.zero1.reset()

    def _publish(self, p):
        o, k = p._xg[1], p._xg[2]
        p.grad = self.grads[o:o + k].view(p.shape)

    def target(self, ps):
        """(fp32 grad view, accumulate?) for a weight-gradient GEMM into parameter(s) ``ps``."""
        if not isinstance(ps, (list, tuple)):
            ps = [ps]
        acc = ps[0].grad is not None
        for p in ps:
            if (p.grad is not None) != acc:
                raise RuntimeError("fused parameters disagree on having a gradient")
            if not acc:
                self._publish(p)
        # matrices (never "atomic" tensors) go to the wire arena when there is one
        return self.grad_view(ps, wire=self.wire is not None and not ps[0]._xg[4]), acc

    def sq_target(self, ps, like):
        """the norm slots of the weight-gradient GEMM into ``ps`` (adjacent, equally wide parameters), or None when
        this gradient has to be read by the norm pass (not an encoder matrix, fp32 mode, data-parallel)."""
        if not self.sq_enabled or like.dtype != torch.bfloat16:
            return None
        if not isinstance(ps, (list, tuple)):
            ps = [ps]
        names = [p._xg[5] for p in ps]
        if any(n not in self.sq_base for n in names):
            return None
        base = self.sq_base[names[0]]
        k = 0
        for p, n in zip(ps, names):  # fused parameters: adjacent in the arena, hence in the slot table
            if self.sq_base[n] != base + k or p.shape[1] != ps[0].shape[1]:
                return None
            k += p.numel() // 4096
        if not self.sq_clean:
            ops.zero_ranges(self.sq_slots, [(0, self.sq_slots.numel())])
            self.sq_clean = True
        self.sq_covered.update(names)
        return self.sq_slots[base:base + k]

    def _clear_for_first_touch(self, p):
        name = p._xg[5]
        if not self.vec_zeroed:
            # first atomic touch after zero_grad(): one fill per group, valid only while no
            # atomically accumulated gradient is live
            self.all_dirty = not all(q.grad is None for q in self._atomic_params)
            rl = self.row_list
            if not self.all_dirty:
                ranges = [(G.vec_start, G.end) for G in self.groups.values()]
                if rl is not None and rl.clean and self.row_list_enabled:
                    # every non-zero row of the word table's gradient is on the device-side list the last pass left:
                    # those rows are cleared instead of the table's 94 MB (same launch)
                    ranges = _cut(ranges, rl.o, rl.o + rl.R * rl.H)
                    ops.zero_ranges(self.grads, ranges, rows=(self.grads[rl.o:rl.o + rl.R * rl.H].view(rl.R, rl.H), rl.ids, rl.n))
                else:
                    ops.zero_ranges(self.grads, ranges)  # one launch
                    if rl is not None:
                        rl.clean = True  # the table is all zero: any list covers its non-zero rows
                self.touched.clear()
            elif rl is not None:
                rl.clean = False  # gradients are being accumulated over several backwards: the list would miss rows
            self.vec_zeroed = True
        if self.all_dirty or not p._xg[4] or name in self.touched:
            o, k = p._xg[1], p._xg[2]
            ops.zero_ranges(self.grads, [(o, o + (k + 3) // 4 * 4)])  # offsets are 8-aligned: the padding is ours
        self.touched.add(name)

    def row_list_for(self, p, M):
        """the (ids, sq, n) buffers the embedding backward fills for table ``p`` when the row-sparse bookkeeping of its
        gradient holds for this pass (RowList), else None (the table is then cleared and read as a whole)"""
        if not self.row_list_enabled:
            return None
        rl = self.row_list
        if rl is None or rl.name != p._xg[5]:
            rl = self.row_list = RowList(self, p)
            # first backward through this table: its gradient is all zero if this pass began with the groups' fill
            rl.clean = bool(p._xg[4]) and self.vec_zeroed and not self.all_dirty
        if not rl.clean or self.emb_uses != 1 or M > ROW_LIST_CAP:
            rl.clean, rl.listed = False, 0
            return None
        rl.listed = int(M)
        return rl.ids, self.sq_slots[self.row_sq0:self.row_sq0 + ROW_LIST_CAP], rl.n

    def atomic_target(self, ps):
        """fp32 grad view for gradients accumulated with atomics: cleared on first touch."""
        if not isinstance(ps, (list, tuple)):
            ps = [ps]
        for p in ps:
            if p.grad is None:
                self._clear_for_first_touch(p)
                self._publish(p)
        return self.grad_view(ps)

    def active_groups(self):
        """groups whose parameters received a gradient in this pass.  A group is used as a
        whole by construction of the passes; a partially touched group is reported too (its
        untouched members then hold zero gradients)."""
        act = []
        for g, G in self.groups.items():
            if any(p.grad is not None for p in G.params):
                act.append(g)
        return act


ROW_LIST_CAP = 8192  # look-ups of the word table per pass the row list holds (B * T: 5120 at 256 samples of 20 tokens)


class RowList:
    """Row-sparse bookkeeping of ONE embedding table's gradient (the word table: 30522 x 768 fp32 = 94 MB, of which a
    pass touches at most B * T rows; src/lxrt/modeling.py:298-313).  Invariant while ``clean``: every non-zero row of the
    table's gradient is among ``ids[:n]`` (device buffers the embedding backward rewrites, xggm_embed_bwd_listed_*).
    Then the start of the next backward clears those rows instead of the table (ParamArena._clear_for_first_touch) and
    clip_grad_norm_ adds the ``listed`` per-row sums of squares instead of reading it.  ``p.grad`` stays the dense view
    and the update stays dense (moments and weight decay move every row, src/lxrt/optimization.py:159-193).  Anything the
    list cannot vouch for -- two look-ups in one pass, gradients accumulated over several backwards, more look-ups than
    the buffers hold, gradients exchanged between ranks -- drops ``clean``, and the next pass falls back to the dense
    clear and read."""

    def __init__(self, arena, p):
        self.name, self.o = p._xg[5], p._xg[1]
        self.R, self.H = p.shape
        self.ids = torch.zeros(ROW_LIST_CAP, device=arena.device, dtype=torch.int64)
        self.n = torch.zeros(1, device=arena.device, dtype=torch.int32)
        self.clean = False
        self.listed = 0  # rows the embedding backward of THIS pass has listed (0: the norm reads the table)


def _cut(ranges, a, b):
    """(start, end) ranges minus [a, b)"""
    out = []
    for s, e in ranges:
        if e <= a or s >= b:
            out.append((s, e))
            continue
        if s < a:
            out.append((s, a))
        if e > b:
            out.append((b, e))
    return out


def arena_of(model, compute_dtype=None):
    """the model's arena, (re)built lazily when the parameters moved."""
    a = getattr(model, "_xg_arena", None)
    if a is None or not a.valid() or (compute_dtype is not None and a.compute_dtype != compute_dtype):
        a = ParamArena(model, compute_dtype or getattr(model, "compute_dtype", torch.bfloat16))
        object.__setattr__(model, "_xg_arena", a)
    return a
