"""Mixed fp8 forward (BASELINE.json configs[4]: "fp8 MFMA path for LXMERT QKV/FFN GEMMs + bf16 GNN").

What runs on OCP e4m3 operands: the FORWARD products of the encoder's Linear layers -- query / key / value
(src/lxrt/modeling.py:345-347), attention output (:385), intermediate (:429) and output (:442) -- i.e. four of
every six GEMMs of a BertLayer forward.  Everything else keeps bf16 storage: the backward (dgrad reads the weights
with the reduction index on the slow dimension, wgrad both operands), the graph generator, the heads, the visual
feature projection.  The reference itself is fp32 only (``--fp16`` is declared and never read, src/param.py:50).

No quantisation pass exists in the step:
  * activations leave their PRODUCERS as e4m3 next to the bf16 copy the backward needs: the residual LayerNorm
    (``xggm_ln_fwd_problem.out8``), the GELU epilogue of the intermediate product (``xggm_gemm_problem.c8``) and the
    attention core (``xggm_attn_problem.out8``); only the two encoder inputs (embedding output, visual-feature
    encoder output) go through ``xggm_quantize_fp8e4m3``;
  * weights get their e4m3 copy from the optimiser pass that updates them (``xggm_bertadam_ex.shadow8``).
Scaling is per tensor and delayed: one device-resident table (``xggm_fp8_scale_update``) holds, per weight operand
and per activation site, the recorded maximum, a short history, the quantisation scale and its reciprocal (which the
GEMM epilogue multiplies back in).  The first forward after ``enable_fp8`` is a calibration pass: products run in
bf16 while the producers record maxima.
"""
import torch

from . import ops

HIST = 4          # steps of amax history per entry
CAP = 1024        # entries of the scale table (weight operands + activation sites)
AMAX_SLOTS = 64   # floats every amax entry is spread over: workgroup w of a producer raises slot w % 64, the scale update
                  # takes the maximum -- hundreds of workgroups raising ONE address cost 0.25 ms per fp8 iteration
MARGIN_ACT = 1.25
MARGIN_W = 4.0 / 3.0


class Fp8State:
    def __init__(self, arena, model):
        if arena.shadow is None:
            raise RuntimeError("the fp8 forward needs bf16 storage (compute_dtype = torch.bfloat16)")
        dev = arena.device
        self.arena = arena
        self.amax = torch.zeros(CAP * AMAX_SLOTS, device=dev)
        self.hist = torch.zeros(CAP * HIST, device=dev)
        self.qscale = torch.full((CAP,), -1.0, device=dev)  # <= 0: not calibrated
        self.dscale = torch.ones(CAP, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int64, device=dev)
        self.shadow8 = torch.zeros(arena.total, dtype=torch.uint8, device=dev)
        self.w8_id = torch.zeros(arena.total // 256 + 1, dtype=torch.int16, device=dev)
        self.n = 1                # entry 0 = "no e4m3 copy"
        self.w_groups = []        # (entry, [params]) per weight operand
        self.w_range = {}         # arena group -> (first entry, count)
        self.w_entry = {}         # id(param) -> entry
        self.sites = {}           # key -> entry
        self.calibrated = False   # False until the first scale update after a forward has seen every site
        self.seen_forward = False
        self._q8 = {}             # data_ptr of a bf16 activation -> (e4m3 copy, entry)
        self.n_w = None
        self._register_weights(model)
        self.n_w = self.n
        self.requantize_weights()

    # ------------------------------------------------------------------ weights
    def _register_weights(self, model):
        from .lxrt.modeling import BertAttention, BertAttOutput, BertIntermediate, BertOutput
        enc = model.lxrt_encoder.model if hasattr(model, "lxrt_encoder") else model
        per_group = {}
        for mod in enc.modules():
            if isinstance(mod, BertAttention):
                ps = [mod.query.weight, mod.key.weight, mod.value.weight]  # one operand when fused, one scale always
            elif isinstance(mod, (BertAttOutput, BertIntermediate, BertOutput)):
                ps = [mod.dense.weight]
            else:
                continue
            per_group.setdefault(ps[0]._xg[3], []).append(ps)
        # entries of one arena group are contiguous: the optimiser updates the scales of exactly the groups it steps
        for gname, lst in per_group.items():
            first = self.n
            for ps in lst:
                e = self.n
                self.n += 1
                if self.n >= CAP:
                    raise RuntimeError("fp8 scale table full")
                self.w_groups.append((e, ps))
                for p in ps:
                    _, o, k, g_ = p._xg[:4]
                    assert g_ == gname and o % 256 == 0, "e4m3 weight operands start on a 256-element arena chunk"
                    self.w8_id[o >> 8:(o + k + 255) >> 8] = e
                    self.w_entry[id(p)] = e
            self.w_range[gname] = (first, self.n - first)

    @torch.no_grad()
    def requantize_weights(self):
        """scales and e4m3 copies from the current bf16 weights (enable, load_state_dict): the range of an operand is
        MARGIN_W x its largest magnitude, as the delayed update keeps it"""
        a = self.arena
        for e, ps in self.w_groups:
            m = max(float(a.shadow[p._xg[1]:p._xg[1] + p._xg[2]].abs().max()) for p in ps)
            q = 448.0 / (m * MARGIN_W) if m > 0 else 1.0
            self.qscale[e] = q
            self.dscale[e] = 1.0 / q
            for p in ps:
                o, k = p._xg[1], p._xg[2]
                ops.quantize_fp8(a.shadow[o:o + k], qscale=self.qscale[e:e + 1], out=self.shadow8[o:o + k])
        nw = self.n_w or self.n
        self.hist[:nw * HIST].zero_()
        self.amax[:nw * AMAX_SLOTS].zero_()

    def w8(self, ps):
        """e4m3 [rows, cols] view of (adjacent) weight parameters + the 1-element reciprocal-scale tensor"""
        if not isinstance(ps, (list, tuple)):
            ps = [ps]
        o0, k = ps[0]._xg[1], 0
        e = self.w_entry[id(ps[0])]
        for p in ps:
            assert p._xg[1] == o0 + k and self.w_entry[id(p)] == e
            k += p.numel()
        return self.shadow8[o0:o0 + k].view(-1, ps[0].shape[1]), self.dscale[e:e + 1]

    def weight_amax(self):
        """the maxima recorded for the weight operands (entries [0, n_w) of the table, AMAX_SLOTS floats each): what
        the sharded update all-reduces (MAX) over the data-parallel ranks"""
        return self.amax[:self.n_w * AMAX_SLOTS]

    def adam_w8(self, gname):
        """arguments of the e4m3 copy for the update of arena group ``gname`` (None: the group has none)"""
        if gname not in self.w_range:
            return None
        return self.w8_id, self.qscale, self.amax, AMAX_SLOTS

    def update_weight_scales(self, gname):
        """new scales for the operands of one arena group, BEFORE its BertAdam launch rewrites their e4m3 copies"""
        if gname in self.w_range:
            i0, n = self.w_range[gname]
            ops.fp8_scale_update(self.amax, self.hist, self.qscale, self.dscale, self.pos, i0, n, HIST, MARGIN_W, 0, 0,
                                 slots=AMAX_SLOTS)

    def update_weight_scales_of(self, gnames):
        """the same for several arena groups: their entries of the table as few launches as the ranges allow (adjacent
        ranges merge; the groups of a pass are laid out back to back: one launch)"""
        rs = sorted(self.w_range[g] for g in gnames if g in self.w_range)
        merged = []
        for i0, n in rs:
            if merged and merged[-1][0] + merged[-1][1] == i0:
                merged[-1][1] += n
            else:
                merged.append([i0, n])
        for i0, n in merged:
            for j in range(0, n, 1024):  # the kernel takes at most 1024 entries
                ops.fp8_scale_update(self.amax, self.hist, self.qscale, self.dscale, self.pos, i0 + j, min(1024, n - j), HIST,
                                     MARGIN_W, 0, 0, slots=AMAX_SLOTS)

    # ------------------------------------------------------------------ activations
    def site(self, key):
        e = self.sites.get(key)
        if e is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("fp8: a new activation site appeared during graph capture; run one eager pass first")
            e = self.sites[key] = self.n
            self.n += 1
            if self.n >= CAP:
                raise RuntimeError("fp8 scale table full")
            self.calibrated = False  # its scale is unknown: the next forward calibrates again
        return e

    def emit(self, key):
        """(qscale, amax) handed to a producer kernel, and the entry"""
        e = self.site(key)
        return (self.qscale[e:e + 1], self.amax[e * AMAX_SLOTS:(e + 1) * AMAX_SLOTS]), e

    def put(self, x, x8, e):
        self._q8[x.data_ptr()] = (x8, e, x)

    def get(self, x, key=None):
        """(e4m3 copy, reciprocal scale) of activation ``x``: from its producer when one registered it, else through
        the quantiser under site ``key``"""
        hit = self._q8.get(x.data_ptr())
        if hit is not None and hit[0].numel() == x.numel() and x.is_contiguous():
            # same bytes, possibly another view of them: the embedding kernels hand out [B, T, H], the first layer's
            # products ask with [B * T, H] (the shape test that stood here sent both streams of every pass through
            # the stand-alone quantiser: 4 launches of 13 us per iteration, tools/exp_fp8_quantize_sites.py)
            return hit[0].view(x.shape), self.dscale[hit[1]:hit[1] + 1]
        (q, amax), e = self.emit(key)
        x8 = ops.quantize_fp8(x if x.is_contiguous() else x.contiguous(), qscale=q, amax=amax).view(torch.uint8)
        self.put(x, x8, e)
        return x8, self.dscale[e:e + 1]

    def begin_forward(self):
        self._q8.clear()
        self.seen_forward = True

    def step(self):
        """once per pass (Runtime.advance): new scales for the activation sites from the maxima of this pass"""
        n_act = self.n - self.n_w
        if n_act > 0:
            ops.fp8_scale_update(self.amax, self.hist, self.qscale, self.dscale, self.pos, self.n_w, n_act, HIST, MARGIN_ACT, 1, 1,
                                 slots=AMAX_SLOTS)
        self._q8.clear()
        if self.seen_forward and not torch.cuda.is_current_stream_capturing():
            self.calibrated = True

    @property
    def active(self):
        """products run on e4m3 operands (False during the calibration forward)"""
        return self.calibrated


def enable_fp8(model):
    """switch the encoder's forward QKV / attention-output / FFN products of ``model`` to e4m3 operands"""
    from .runtime import runtime_of
    rt = runtime_of(model)
    if rt.arena.fp8 is None:
        rt.arena.fp8 = Fp8State(rt.arena, model)
    return model


def disable_fp8(model):
    from .runtime import runtime_of
    runtime_of(model).arena.fp8 = None
    return model
