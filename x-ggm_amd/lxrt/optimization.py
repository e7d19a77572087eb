"""BertAdam with the reference's constructor and semantics (src/lxrt/optimization.py:58-203)
running as ONE fused HIP pass per contiguous parameter range: clip scale, moment update,
weight decay, scheduled step and the bf16 shadow-weight write.

State lives in the model's flat arena (``next_m``/``next_v`` = arena.m / arena.v); the
per-parameter ``state['step']`` of the reference becomes one device-resident counter per
arena group (all parameters of a group always step together), from which
``warmup_linear`` is evaluated on the device -- no host synchronisation, graph replayable.
"""
import math

import torch
from torch.optim import Optimizer

from .. import ops


def warmup_cosine(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 1.0


def warmup_linear(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return max((x - 1.) / (warmup - 1.), 0)


SCHEDULES = {'warmup_cosine': warmup_cosine, 'warmup_constant': warmup_constant, 'warmup_linear': warmup_linear}


def _arena_of_params(params):
    for p in params:
        xg = getattr(p, "_xg", None)
        if xg is not None:
            return xg[0]
    return None


def clip_grad_norm_(parameters, max_norm, tail=None):
    """fused replacement of ``nn.utils.clip_grad_norm_(model.parameters(), 5.)``
    (src/vqa/vqacpv2.py:175): one sum-of-squares reduction per active arena range; the scale
    min(1, max_norm/(norm+1e-6)) is applied INSIDE the following BertAdam.step (the gradients
    in memory stay unscaled).  Returns the total norm as a device scalar.
    ``tail`` = (optimiser, runtime or None), given by ``vqa.vqacpv2.clip_and_step``: the launch that finishes the norm
    also takes the schedule step of the ``optimiser.step()`` that follows and -- with a runtime -- the RNG advance that
    ends the pass (``arena.sched_done`` / ``runtime.rng_advanced`` tell the two that their launch has been made)."""
    params = [p for p in parameters]
    arena = _arena_of_params(params)
    if arena is None:
        raise RuntimeError("clip_grad_norm_: parameters are not arena-managed; run a forward first")
    if arena.wire is not None:  # data parallel, bf16 wire arena: the (averaged) gradients live there
        if arena.zero1 is None:
            spans = [(a, b) for a, b in _active_ranges(arena) if b > a]
            if len(spans) <= ops.CLIP_NORM_MAX_SPANS and all(a % 8 == 0 for a, _ in spans):
                # replicated update: every rank sums the whole wire itself -- one pair of launches, the pass's scalar
                # bookkeeping on the finishing one (as on one GPU); |mean|^2 = |sum|^2 / world^2
                total = torch.empty(1, device=arena.grads.device, dtype=torch.float32)
                sched, rng = _tail_of(arena, tail)
                ops.clip_norm_bf16(arena.wire, spans, arena.sqnorm, total, mul=arena.grad_scale ** 2, sched=sched, rng=rng)
                arena.pending_clip = float(max_norm)
                return total.view(())
        clip_norm_local(arena)
        if arena.zero1 is not None:
            arena.zero1.exchange_norm(arena.sqnorm)
        return clip_norm_finish(arena, max_norm, tail)
    # ranges of the flat gradient buffer the norm pass has to READ: everything of the active groups except the
    # matrices whose weight-gradient GEMM already left its sum of squares in the slot table (arena.sq_target);
    # adjacent ranges are merged
    spans, slot_spans = [], []

    def add(a, b):
        if b > a:
            if spans and spans[-1][1] == a:
                spans[-1][1] = b
            else:
                spans.append([a, b])

    for g in sorted(arena.active_groups(), key=lambda n: arena.groups[n].start):
        G = arena.groups[g]
        covered = arena.sq_covered if (arena.sq_enabled and g in arena.sq_range) else ()
        if not covered:
            add(G.start, G.end)
            continue
        pos = G.start
        for o, k, n in sorted((o, k, n) for n, (o, k, g_, _) in arena.info.items() if g_ == g and n in covered):
            add(pos, o)
            pos = o + k  # alignment gaps hold zeros
        add(pos, G.end)
        slot_spans.append(arena.sq_range[g])
    # two launches for all ranges, two more for the slot table; sums in a fixed order (deterministic replicas); the
    # finish kernels seed the running sum and write the norm: no framework fill / add / sqrt kernels in the pass
    rl = arena.row_list
    if rl is not None and rl.clean and rl.listed and arena.row_list_enabled:
        # the word table's gradient: its non-zero rows are the ones the embedding backward of this pass listed, with
        # their sums of squares in the slots behind the products' -- 94 MB the norm does not read (arena.RowList)
        from ..arena import _cut
        cut = _cut([tuple(sp) for sp in spans], rl.o, rl.o + rl.R * rl.H)
        if cut != [tuple(sp) for sp in spans]:
            spans = [list(sp) for sp in cut]
            slot_spans.append((arena.row_sq0, arena.row_sq0 + rl.listed))
    total = torch.empty(1, device=arena.grads.device, dtype=torch.float32)
    if len(spans) + len(slot_spans) <= ops.CLIP_NORM_MAX_SPANS:
        # one pair of launches for the ranges AND the slot table; the finishing one carries the pass's scalar bookkeeping
        sched, rng = _tail_of(arena, tail)
        ops.clip_norm(arena.grads, spans, arena.sq_slots if slot_spans else None, slot_spans, arena.sqnorm, total, sched=sched,
                      rng=rng)
    else:
        ops.sqnorm_multi(arena.grads, spans, arena.sqnorm, None if slot_spans else total, overwrite=True)
        if slot_spans:
            ops.sqnorm_multi(arena.sq_slots, slot_spans, arena.sqnorm, total, overwrite=False, square=False)
    arena.pending_clip = float(max_norm)
    return total.view(())


def _tail_of(arena, tail):
    """(sched, rng) arguments of ops.clip_norm* for ``tail`` = (optimiser, runtime or None) of clip_grad_norm_, and the
    marks that tell ``optimiser.step()`` / ``runtime.advance()`` that their launch has been made"""
    sched = rng = None
    if tail is not None:
        optim, rt = tail
        entries = [(gi, pg['t_total'], pg['warmup']) for _, pg, gi in optim._todo(arena)]
        if entries and len(entries) <= ops.CLIP_NORM_MAX_SCHED:
            sched = (arena.steps, arena.lr_scale, entries)
            arena.sched_done = True
        if rt is not None:
            rng = (rt.rng, 1)
            rt.rng_advanced = True
    return sched, rng


def _active_ranges(arena):
    return [(arena.groups[g].start, arena.groups[g].end)
            for g in sorted(arena.active_groups(), key=lambda n: arena.groups[n].start)]


def clip_norm_local(arena):
    """wire-arena norm, first half: seed the running sum and add what THIS rank sums alone -- everything without a
    sharded update, the own slices of the matrix runs with one (their sum is then all-reduced: ShardedUpdate)"""
    z = arena.zero1
    spans = _active_ranges(arena) if z is None else z.norm_spans(_active_ranges(arena))[0]
    spans = [(a, b) for a, b in spans if b > a]
    if len(spans) <= ops.CLIP_NORM_MAX_SPANS and all(a % 8 == 0 for a, _ in spans):
        ops.clip_norm_bf16(arena.wire, spans, arena.sqnorm)  # seeds the sum: one pair of launches for all ranges
        return
    ops.sqnorm_multi(arena.grads, [], arena.sqnorm, None, overwrite=True)
    for a, b in spans:
        ops.sqnorm_bf16(arena.wire[a:b], arena.sqnorm)


def clip_norm_finish(arena, max_norm, tail=None):
    """second half: the vector ranges every rank holds in full (sharded update only), then the norm"""
    z = arena.zero1
    spans = [(a, b) for a, b in (z.norm_spans(_active_ranges(arena))[1] if z is not None else []) if b > a]
    total = torch.empty(1, device=arena.grads.device, dtype=torch.float32)
    # the wire holds sums over the ranks: |mean|^2 = |sum|^2 / world^2
    if len(spans) <= ops.CLIP_NORM_MAX_SPANS and all(a % 8 == 0 for a, _ in spans):
        sched, rng = _tail_of(arena, tail)
        ops.clip_norm_bf16(arena.wire, spans, arena.sqnorm, total, accumulate=True, mul=arena.grad_scale ** 2, sched=sched, rng=rng)
    else:
        for a, b in spans:
            ops.sqnorm_bf16(arena.wire[a:b], arena.sqnorm)
        ops.sqnorm_multi(arena.grads, [], arena.sqnorm, total, overwrite=False, mul=arena.grad_scale ** 2)
    arena.pending_clip = float(max_norm)
    return total.view(())


class BertAdam(Optimizer):
    """ref: src/lxrt/optimization.py:58-203 (no bias correction; decoupled weight decay on
    every parameter; gradient clipping is done outside, as LXMERT does)."""

    def __init__(self, params, lr, warmup=-1, t_total=-1, schedule='warmup_linear', b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if schedule != 'warmup_linear' and t_total != -1:
            raise NotImplementedError("only warmup_linear (the reference trainers' schedule) runs on the device")
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)

    # ---- checkpointing: the reference layout (src/lxrt/optimization.py:147-155 keeps per-parameter
    # state['step'], state['next_m'], state['next_v']), read from / written into the flat arena
    def _arena(self):
        arena = _arena_of_params(p for pg in self.param_groups for p in pg['params'])
        pending = getattr(self, "_pending_state", None)
        if arena is not None and pending is not None:
            self._pending_state = None
            self._state_into_arena(arena, pending)
        return arena

    def state_dict(self):
        sd = super().state_dict()
        arena = self._arena()
        if arena is None:
            if getattr(self, "_pending_state", None) is not None:
                sd['state'] = self._pending_state
            return sd
        arena.gather_sharded_state()  # ZeRO-1: the moments of the other ranks' slices (collective, see dist.ShardedUpdate)
        steps = arena.steps.tolist()
        state, idx = {}, 0
        for pg in self.param_groups:
            for p in pg['params']:
                xg = getattr(p, "_xg", None)
                if xg is not None and xg[0] is arena:
                    _, o, k, gname = xg[:4]
                    state[idx] = {'step': steps[arena.group_index[gname]],
                                  'next_m': arena.m[o:o + k].view(p.shape).clone(),
                                  'next_v': arena.v[o:o + k].view(p.shape).clone()}
                idx += 1
        sd['state'] = state
        return sd

    def load_state_dict(self, state_dict):
        """hyper-parameters through torch's loader, moments and step counters IN PLACE into the arena (captured
        graphs keep pointing at the same buffers).  Before the arena exists (no forward yet) the state is kept and
        applied at the first use."""
        super().load_state_dict({'state': {}, 'param_groups': state_dict['param_groups']})
        self._pending_state = dict(state_dict.get('state', {}))
        self._arena()

    @torch.no_grad()
    def _state_into_arena(self, arena, state):
        step_of, idx = {}, 0
        for pg in self.param_groups:
            for p in pg['params']:
                st = state.get(idx, state.get(str(idx)))
                idx += 1
                xg = getattr(p, "_xg", None)
                if st is None or xg is None or xg[0] is not arena:
                    continue
                _, o, k, gname = xg[:4]
                arena.m[o:o + k].view(p.shape).copy_(st['next_m'])
                arena.v[o:o + k].view(p.shape).copy_(st['next_v'])
                if step_of.setdefault(gname, int(st['step'])) != int(st['step']):
                    raise ValueError("BertAdam.load_state_dict: parameters of arena group '%s' carry different step "
                                     "counts (%d, %d); they always step together here" % (gname, step_of[gname],
                                                                                          int(st['step'])))
        for gname, s in step_of.items():
            arena.steps[arena.group_index[gname]] = s

    def _hyper_of_group(self, arena, gname):
        """the optimiser param_group that holds ALL parameters of arena group ``gname`` (one contiguous range, one
        launch, one set of hyper-parameters).  A param_groups split that cuts through an arena group -- the usual
        BERT "no decay for bias / LayerNorm" grouping would -- cannot be honoured by a flat update and is refused
        instead of silently applying the first parameter's settings to the whole range."""
        cache = getattr(self, "_group_pg", None)
        if cache is None:
            cache = self._group_pg = {}
        if gname not in cache:
            owner = {}
            for i, pg in enumerate(self.param_groups):
                for p in pg['params']:
                    owner[id(p)] = i
            idx = {owner.get(id(p)) for p in arena.groups[gname].params}
            if len(idx) > 1:
                raise ValueError("BertAdam: the parameters of arena group '%s' are spread over optimiser param_groups %s; "
                                 "a group is updated as one range with one lr / weight_decay / schedule -- keep its "
                                 "parameters in one param_group (vqa.vqacpv2.make_optimizer does)" % (gname, sorted(idx, key=str)))
            cache[gname] = idx.pop()
        i = cache[gname]
        return None if i is None else self.param_groups[i]

    def sync_hyper(self):
        """push ``param_groups[i]['lr']`` to the device table the update kernels read: call after editing a learning
        rate between replays of a captured pass (an eager ``step()`` does it by itself)"""
        arena = self._arena()
        if arena is None:
            return
        for g in arena.groups:
            pg = self._hyper_of_group(arena, g)
            gi = arena.group_index[g]
            if pg is not None and arena.lr_host[gi] != float(pg['lr']):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("BertAdam: a learning rate changed during graph capture; call sync_hyper() before")
                arena.lr_table[gi] = float(pg['lr'])
                arena.lr_host[gi] = float(pg['lr'])

    def get_lr(self):
        """ref :100-114: the scheduled learning rate of every parameter, in param_groups order; ``[0]`` while some
        parameter has never been stepped (the reference's empty ``state[p]``)."""
        arena = self._arena()
        if arena is None:
            return [0]
        steps = arena.steps.tolist()
        lr = []
        for pg in self.param_groups:
            for p in pg['params']:
                xg = getattr(p, "_xg", None)
                if xg is None or xg[0] is not arena:
                    return [0]
                s = steps[arena.group_index[xg[3]]]
                if s == 0:
                    return [0]
                if pg['t_total'] != -1:
                    lr.append(pg['lr'] * SCHEDULES[pg['schedule']](s / pg['t_total'], pg['warmup']))
                else:
                    lr.append(pg['lr'])
        return lr

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=True)
        arena = self._arena()
        if arena is not None:
            arena.begin_pass()

    def _todo(self, arena):
        """(group, its param_group, its index) for every arena group that received gradients in this pass"""
        todo = []
        for g in arena.active_groups():
            G = arena.groups[g]
            pg = self._hyper_of_group(arena, g)
            if pg is None:
                continue  # parameters not handed to this optimiser
            if any(p.grad is None for p in G.params):
                raise RuntimeError("arena group '%s' received gradients for only part of its parameters" % g)
            todo.append((G, pg, arena.group_index[g]))
        return todo

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        arena = self._arena()
        if arena is None:
            raise RuntimeError("BertAdam.step: parameters are not arena-managed; run a forward/backward first")
        sq = arena.sqnorm if arena.pending_clip is not None else None
        max_norm = arena.pending_clip if arena.pending_clip is not None else 0.0
        todo = self._todo(arena)
        if todo and not getattr(arena, "sched_done", False):  # schedule values and step counters of all groups: one launch
            ops.sched_step_multi(arena.steps, arena.lr_scale, [(gi, pg['t_total'], pg['warmup']) for _, pg, gi in todo])
        arena.sched_done = False  # (True: clip_grad_norm_'s finishing launch has taken the step along)
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()  # the kernels read lr from a device table: edits of param_groups survive graph replay
        elif any(arena.lr_host[gi] is None for _, _, gi in todo):
            raise RuntimeError("BertAdam.step is being captured before any eager step: run one pass eagerly first "
                               "(the learning rates are uploaded to the device outside of captures)")
        f8 = arena.fp8
        gbuf = arena.wire if arena.wire is not None else arena.grads  # bf16 wire arena: the update reads it directly
        gscale = arena.grad_scale if arena.wire is not None else 1.0     # ... and holds sums over the ranks
        z = arena.zero1
        jobs = []  # every span of this step: ONE launch (xggm_bertadam_multi)
        if f8 is not None:
            # new scales of the weight operands of every group, before the update rewrites their e4m3 copies
            f8.update_weight_scales_of([G.name for G, _, _ in todo])
        for G, pg, gi in todo:
            scale_t = arena.lr_scale[gi:gi + 1]
            if z is not None:
                # sharded update: this rank's slice of every matrix run of the group + the whole vector region.  With the
                # fp8 forward a slice also writes ITS part of the e4m3 weight copies (slices are whole 256-element chunks
                # of the scale-id table) under scales derived from maxima that ShardedUpdate.exchange_norm has just
                # MAX-reduced over the ranks -- identical tables on every rank; gather() brings the other slices over
                w8g = f8.adam_w8(G.name) if f8 is not None else None
                pieces = [z.own(r) + (True,) for r in z.runs if r[0] >= G.start and r[1] <= G.vec_start]
                pieces.append((G.vec_start, G.end, False))
                for a, b, is_mat in pieces:
                    if b > a:
                        sl = slice(a, b)
                        w8 = ((f8.shadow8[sl],) + w8g) if (w8g is not None and is_mat) else None
                        jobs.append(((arena.params[sl], gbuf[sl], arena.m[sl], arena.v[sl], arena.shadow[sl], sq, max_norm,
                                      pg['lr'], scale_t, pg['b1'], pg['b2'], pg['e'], pg['weight_decay']),
                                     dict(lr_dev=arena.lr_table[gi:gi + 1], w8=w8, elem0=a, g_scale=gscale)))
                continue
            sl = slice(G.start, G.end)
            w8 = f8.adam_w8(G.name) if f8 is not None else None
            if w8 is not None:
                # fp8 forward: this group's weight operands get their e4m3 copies from the same pass over p, with
                # the scales the delayed update derives from the maxima earlier steps recorded
                w8 = (f8.shadow8[sl],) + w8
            jobs.append(((arena.params[sl], gbuf[sl], arena.m[sl], arena.v[sl],
                          None if arena.shadow is None else arena.shadow[sl], sq, max_norm, pg['lr'], scale_t,
                          pg['b1'], pg['b2'], pg['e'], pg['weight_decay']),
                         dict(lr_dev=arena.lr_table[gi:gi + 1], w8=w8, elem0=G.start, g_scale=gscale)))
        ops.bertadam_multi(jobs)
        arena.pending_clip = None
        return loss
