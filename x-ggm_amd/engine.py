"""Graph-captured training iteration.

Eager execution of one pass issues ~700 kernel launches through Python; at 32 samples per GPU
each kernel runs for microseconds, so the host would be the bottleneck.  ``CapturedTrainer``
records each pass ONCE into a hipGraph (torch.cuda.CUDAGraph: every kernel here is launched on
torch's current stream, so stream capture sees them all, forward, backward, clip and BertAdam
alike) and replays it per step -- "HIP graphs instead of a tracing compiler".  Everything a
replay needs to differ from the previous one lives in device memory: the batch (static input
buffers), the Philox offset, the per-group step counters and schedule values.

With data parallelism each pass is captured as two graphs (forward+backward | clip+update)
and the gradient all-reduce runs between them on the live RCCL communicator; with the overlap
option (default) the backward is cut below the cross-modality layers into a third graph, and the
gradients above the cut are already on the wire while the lower backward graph replays.
With the sharded update (ZeRO-1) on top of the overlap the FORWARD is cut at the same places: the
all-gather of the bf16 weights that follows the update is queued stage by stage in forward order
on the communication stream, forward graph i waits only for batch i, and the gather of stage
i + 1 runs under the forward of stage i (dist.ShardedUpdate.gather_begin).
"""
import gc

import torch

from . import ops

from .runtime import runtime_of
from .vqa.vqacpv2 import (forward_backward_plain, forward_backward_ggm, clip_and_step, _sync_grads,
                          BCEWithLogitsLoss)


class _quiet_gc:
    """no garbage collection while a stream is being captured: a finalizer that runs in the middle of a capture
    (on whatever thread allocates next, the autograd thread included) may free pinned host memory or destroy
    events -- the caching host allocator then queries an event, which HIP refuses during a global-mode capture,
    and the process aborts.  Collect first, then keep the collector off until the capture has ended."""

    def __enter__(self):
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *a):
        if self.was:
            gc.enable()


class CapturedTrainer:
    def __init__(self, model, optim, batch, sigma=1.0, order="vqa", clip=5.0, use_graph=True, warmup_iters=2,
                 packed_spec=None):
        """``batch``: dict of DEVICE tensors feats, boxes, input_ids, input_mask, segment_ids,
        target, adj_true; they become the static input buffers (``load_batch`` copies into them).
        ``packed_spec`` = ``DataLoaderX.spec`` of a loader with ``handover="inline"``: the static buffers are then
        views of ONE flat device buffer laid out like the loader's pinned slots, and ``load_packed`` hands a batch over
        with a single host-to-device copy on the compute stream."""
        self.model, self.optim = model, optim
        self.rt = runtime_of(model)
        self.static_flat = None
        if packed_spec is not None:
            from .tools.data_loader import packed_like
            dev = next(model.parameters()).device
            self.static_flat, v = packed_like(packed_spec, dev)
            self.static = dict(feats=v["feats"], boxes=v["boxes"], input_ids=v["ids"][0], input_mask=v["ids"][1],
                               segment_ids=v["ids"][2], target=v["target"], adj_true=v["adj"])
            for k, t in self.static.items():
                if tuple(t.shape) != tuple(batch[k].shape) or t.dtype != batch[k].dtype:
                    raise ValueError("packed_spec field %s is %s %s, the batch has %s %s"
                                     % (k, tuple(t.shape), t.dtype, tuple(batch[k].shape), batch[k].dtype))
                t.copy_(batch[k])
        else:
            self.static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        self.sigma, self.clip = sigma, clip
        self.kl_weight = 8.0 if order == "vqa" else 12.0
        self.order = order
        self.bce = BCEWithLogitsLoss()
        self.use_graph = use_graph
        self.graphs = {}
        self.outputs = {}
        self.split = getattr(model, "_grad_sync", None) is not None
        self._comm = None
        # With a process group alive its watchdog thread polls events at any time; under the default "global"
        # capture mode HIP rejects that ("operation not permitted when stream is capturing") and the process
        # aborts.  "thread_local" only polices the capturing thread.
        self.capture_mode = "thread_local" if self.split else "global"
        model.train()
        if use_graph:
            self._capture(warmup_iters)

    # ------------------------------------------------------------------ the two halves of a pass
    def _fwd_bwd(self, kind, between=None):
        s = self.static
        sent = (s["input_ids"], s["input_mask"], s["segment_ids"])
        if kind == "plain":
            loss, logit = forward_backward_plain(self.model, self.bce, s["feats"], s["boxes"], sent, s["target"],
                                                 between=between)
        else:
            loss, logit, _ = forward_backward_ggm(self.model, self.bce, s["feats"], s["boxes"], sent, s["target"],
                                                  s["adj_true"], kind, self.sigma, self.kl_weight, between=between)
        return loss, logit

    def _update(self):
        return clip_and_step(self.model, self.optim, self.clip, advance=True)

    # the update as graphs: one, or -- sharded update -- two with the norm's scalar all-reduce between them and the
    # all-gather of the weights behind them (collectives run on the live communicator, never inside a capture)
    def _capture_update(self, pool):
        z = self.rt.arena.zero1
        if z is None:
            gu = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gu, pool=pool, capture_error_mode=self.capture_mode):
                total = self._update()
            return (gu,), total
        from .lxrt.optimization import clip_norm_local, clip_norm_finish
        arena = self.rt.arena
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, pool=pool, capture_error_mode=self.capture_mode):
            clip_norm_local(arena)
        z.exchange_norm(arena.sqnorm)
        with torch.cuda.graph(g2, pool=pool, capture_error_mode=self.capture_mode):
            total = clip_norm_finish(arena, self.clip, tail=(self.optim, None))  # (the schedule step rides on the finish)
            self.optim.step()
        z.gather()
        self.optim.zero_grad()
        # Runtime.advance (new dropout masks for the next pass) is NOT run here: a capture executes nothing, and the
        # replay issues it eagerly behind the gather (_replay_update) -- exactly once per trained pass
        return (g1, g2), total

    def _replay_update(self, gus, staged_gather=False):
        if len(gus) == 1:
            gus[0].replay()
            return
        z = self.rt.arena.zero1
        gus[0].replay()
        z.exchange_norm(self.rt.arena.sqnorm)
        gus[1].replay()
        if staged_gather:
            # the bf16 weights come back stage by stage beside the NEXT pass's forward graphs (run_pass waits per stage)
            z.gather_begin(self._comm_stream())
        else:
            z.gather()
        self.rt.advance()

    def _comm_stream(self):
        if self._comm is None:
            self._comm = torch.cuda.Stream()
        return self._comm

    def _stage_ranges(self):
        from .dist import active_ranges, stage_ranges
        return stage_ranges(self.rt.arena, active_ranges(self.rt.arena), self.rt.cut_layout, self.rt.n_stages)

    def _eager_pass(self, kind):
        if self.split and self.rt.cut_enabled:
            gs, handles = self.model._grad_sync, []

            def between(k):  # gradients above cut k are final: put them on the wire
                handles.append(gs.begin(self._stage_ranges()[k], slot=k))

            loss, logit = self._fwd_bwd(kind, between)
            gs.sync(self._stage_ranges()[-1])
            for h in handles:
                gs.finish(h)
        else:
            loss, logit = self._fwd_bwd(kind)
            _sync_grads(self.model)
        total = self._update()
        return loss, logit, total

    # ------------------------------------------------------------------ capture
    def _capture(self, warmup_iters):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup_iters):
                for kind in ("plain", "rel", "node"):
                    self._eager_pass(kind)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.split:
            # nothing of the eager warm-up exchanges may still be in flight (or polled) while graphs are captured
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()
        with _quiet_gc():
            self._capture_kinds()
        torch.cuda.synchronize()

    def _capture_kinds(self):
        for kind in ("plain", "rel", "node"):
            # one memory pool per pass kind (shared by the graphs of that kind only): with a pool shared across kinds
            # the outputs of one kind (loss, logits, norm) can land where an earlier-captured kind keeps its
            # intermediates, and replaying that kind then overwrites them before the caller has read them
            pool = None
            if not self.split:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode=self.capture_mode):
                    out = self._eager_pass(kind)
                pool = g.pool()
                self.graphs[kind] = (g,)
                self.outputs[kind] = out
            elif not self.rt.cut_enabled:
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1, pool=pool, capture_error_mode=self.capture_mode):
                    loss, logit = self._fwd_bwd(kind)
                pool = g1.pool()
                ranges = None
                from .dist import active_ranges
                ranges = active_ranges(self.rt.arena)
                if self.rt.arena.zero1 is not None:
                    # the sharded update records its runs at the exchange: run the one of this pass kind (on whatever the
                    # buffers hold -- the captured kernels have not run; every rank does the same)
                    self.rt.arena.zero1.reset()
                    self.model._grad_sync.sync(ranges)
                gus, total = self._capture_update(pool)
                self.graphs[kind] = (g1, gus, ranges)
                self.outputs[kind] = (loss, logit, total)
            else:
                # one graph per backward stage: forward + stage 0 | stage 1 | ... | clip + update.  The capture is
                # switched from one graph to the next inside Runtime.backward (the ``between`` callback).
                graphs, early = [torch.cuda.CUDAGraph()], []
                n_fwd = [0]  # forward cuts passed = forward graphs in front of the one that also holds backward stage 0

                def switch(k):
                    graphs[-1].capture_end()
                    early.append(self._stage_ranges()[k])
                    graphs.append(torch.cuda.CUDAGraph())
                    graphs[-1].capture_begin(pool=graphs[0].pool(), capture_error_mode=self.capture_mode)

                def fwd_switch(i):  # sharded update: one graph per forward stage (see the module docstring)
                    graphs[-1].capture_end()
                    graphs.append(torch.cuda.CUDAGraph())
                    graphs[-1].capture_begin(pool=graphs[0].pool(), capture_error_mode=self.capture_mode)
                    n_fwd[0] += 1

                torch.cuda.synchronize()
                gc.collect()
                torch.cuda.empty_cache()
                cap = torch.cuda.Stream()
                cap.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cap):
                    if pool is None:
                        graphs[0].capture_begin(capture_error_mode=self.capture_mode)
                    else:
                        graphs[0].capture_begin(pool=pool, capture_error_mode=self.capture_mode)
                    self.rt.fwd_hook = fwd_switch if self.rt.arena.zero1 is not None else None
                    try:
                        loss, logit = self._fwd_bwd(kind, switch)
                    finally:
                        self.rt.fwd_hook = None
                    graphs[-1].capture_end()
                torch.cuda.current_stream().wait_stream(cap)
                pool = graphs[0].pool()
                final = self._stage_ranges()[-1] if early else None
                if final is None:
                    from .dist import active_ranges
                    final = active_ranges(self.rt.arena)
                if self.rt.arena.zero1 is not None:
                    # the runs of this pass kind, recorded as the real exchange would record them
                    self.rt.arena.zero1.reset()
                    for rs in early + [final]:
                        self.model._grad_sync.sync(rs)
                gus, total = self._capture_update(pool)
                self.graphs[kind] = ("staged", graphs, early, final, gus, n_fwd[0])
                self.outputs[kind] = (loss, logit, total)
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ run
    def load_batch(self, batch):
        for k, v in batch.items():
            if k in self.static:
                self.static[k].copy_(v, non_blocking=True)

    def load_packed(self, it):
        """hand over the batch the loader iterator ``it`` (DataLoaderX(handover="inline")) has just yielded: ONE copy of
        the slot's pinned flat buffer into the static inputs, in stream order on the compute stream (it runs between
        the previous iteration's graphs and the next one's: no second stream, no event the graphs wait for, no
        device-to-device copies), then the slot is marked so that the producer does not rewrite it under the copy"""
        if self.static_flat is None:
            raise RuntimeError("load_packed needs CapturedTrainer(packed_spec=loader.spec)")
        if it.flat is None or it.flat.numel() != self.static_flat.numel():
            raise ValueError("the iterator's slot does not have this trainer's layout (handover='inline', same batch size)")
        rows = getattr(it, "rows", None)
        if rows is not None and rows != self.static["feats"].shape[0]:
            # a short (last) batch fills only the head of the slot: the captured step would train on its rows PLUS the
            # stale tail of an older batch (the ring path fails on the shape in copy_; this one has to say so itself)
            raise ValueError("load_packed: the batch holds %d samples, the captured step %d -- iterate the loader with "
                             "drop_last=True (or run the short batch through an eager pass)" % (rows, self.static["feats"].shape[0]))
        self.static_flat.copy_(it.flat, non_blocking=True)
        it.mark_copied()

    def run_pass(self, kind):
        if not self.use_graph:
            return self._eager_pass(kind)
        gs = self.graphs[kind]
        arena = self.rt.arena
        # host-side bookkeeping a replay relies on but never runs (the Python of a pass executes at CAPTURE time only):
        # the sparse exchange of the word table gathers the rows of THIS trainer's ids, looked up once per pass -- an eval
        # forward or a second trainer's capture in between would otherwise leave another batch's ids or a count != 1 behind
        self.rt.emb_ids = self.static["input_ids"]
        arena.emb_uses = 1
        rl = arena.row_list
        if rl is not None and not rl.clean and arena.row_list_enabled:
            # an eager pass in between left rows of the word table's gradient the device-side list does not name (two
            # look-ups, accumulation): the captured clear only knows the list -- clear the whole table once
            ops.zero_ranges(arena.grads, [(rl.o, rl.o + rl.R * rl.H)])
            rl.clean = True
        if arena.zero1 is not None:
            arena.zero1.reset()  # the runs of this pass are recorded again by its exchanges
        if len(gs) == 1:
            gs[0].replay()
        elif gs[0] != "staged":
            gs[0].replay()
            self.model._grad_sync.sync(gs[2])
            self._replay_update(gs[1])
        else:
            # the exchange of stage k (cast to the wire type, all-reduce, copy back) is queued on a side stream
            # right behind graph k and runs under graph k + 1; the update waits for all of it
            _, graphs, early, final, gus, n_fwd = gs
            sync = self.model._grad_sync
            main = torch.cuda.current_stream()
            comm = self._comm_stream()
            z = self.rt.arena.zero1
            # all-gathers of the previous pass's update that are still running: pend[i] covers what forward graph i reads
            pend = z.take_pending() if z is not None else []

            def exchange(k, after):
                comm.wait_event(after)
                with torch.cuda.stream(comm):
                    if k < len(early):
                        sync.finish(sync.begin(early[k], slot=k))
                    else:
                        sync.sync(final)

            # graph k + 1 is handed to the GPU BEFORE the host enqueues the (eager) exchange of stage k: those dozens
            # of small launches take the host longer than the GPU needs to get to the end of graph k
            prev = None
            for i, g in enumerate(graphs):
                if i < len(pend):
                    main.wait_event(pend[i])
                if i == n_fwd:  # the last forward graph (it also holds backward stage 0): nothing may be left running
                    for ev in pend[i + 1:]:
                        main.wait_event(ev)
                g.replay()
                k = i - n_fwd  # backward stage this graph ends with (negative: a forward-only graph)
                if k < 0:
                    continue
                ev = torch.cuda.Event()
                ev.record(main)
                if prev is not None:
                    exchange(k - 1, prev)
                prev = ev
            exchange(len(graphs) - 1 - n_fwd, prev)
            main.wait_stream(comm)
            self._replay_update(gus, staged_gather=z is not None and n_fwd > 0)
        return self.outputs[kind]

    def iteration(self, branch):
        """one training iteration: VQA order = plain then GGM; GQA order = GGM then plain."""
        if self.order == "vqa":
            a = self.run_pass("plain")
            b = self.run_pass(branch)
        else:
            b = self.run_pass(branch)
            a = self.run_pass("plain")
        return a, b


class CapturedPredictor:
    """The validation sweep of the reference (``VQA.predict``, src/vqa/vqacpv2.py:315-339; GQA twin
    src/gqa/gqa_ood.py:379-403): eval mode, no autograd, encoder -> ``logit_fc`` -> arg-max; the generator is
    not on this path.  The forward of one full batch is captured once and replayed; a short last batch is
    padded (its padding rows are computed and dropped: samples are independent, so the kept rows do not change).
    The logits are fp32 and the arg-max is torch's (first maximal index), as in ``logit.max(1)``."""

    def __init__(self, model, batch_size, n_objects=36, feat_dim=None, max_seq_length=None, use_graph=True):
        self.model = model
        dev = next(model.parameters()).device
        enc = model.lxrt_encoder
        T = max_seq_length or enc.max_seq_length
        F = feat_dim or enc.model.bert.encoder.visn_fc.visn_fc.in_features
        self.B = batch_size
        self.static = dict(feats=torch.zeros(batch_size, n_objects, F, device=dev),
                           boxes=torch.zeros(batch_size, n_objects, 4, device=dev),
                           ids=torch.zeros(3, batch_size, T, dtype=torch.long, device=dev))
        self.static["ids"][1, :, 0] = 1  # a valid mask for the rows nobody has filled yet
        self.graph, self.logit, self.label = None, None, None
        self.use_graph = use_graph
        if use_graph:
            was_training = model.training
            model.eval()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._forward()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with _quiet_gc(), torch.cuda.graph(self.graph):
                self.logit, self.label = self._forward()
            model.train(was_training)

    def _forward(self):
        s = self.static
        with torch.no_grad():
            _, _, x = self.model(s["feats"], s["boxes"], (s["ids"][0], s["ids"][1], s["ids"][2]))
            logit = self.model.logit_fc(x)
            return logit, logit.max(1)[1]

    def __call__(self, feats, boxes, sent):
        """-> (labels [b] int64, logits [b, A] fp32) for b <= batch_size samples; ``sent``: list of strings or the
        (input_ids, input_mask, segment_ids) tuple.  The returned tensors are views of static buffers: consume
        them (``.cpu()``) before the next call."""
        b = feats.shape[0]
        if b > self.B:
            raise ValueError("batch of %d exceeds the captured batch size %d" % (b, self.B))
        s = self.static
        if isinstance(sent, (tuple, list)) and len(sent) == 3 and torch.is_tensor(sent[0]):
            for i in range(3):
                s["ids"][i, :b].copy_(sent[i], non_blocking=True)
        else:
            batcher = self.model.lxrt_encoder.batcher
            s["ids"][:, :b].copy_(batcher.host_batch(list(sent)), non_blocking=True)
            batcher.record_copy()  # the pinned buffer is not rewritten before this copy has read it
        s["feats"][:b].copy_(feats, non_blocking=True)
        s["boxes"][:b].copy_(boxes, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            was_training = self.model.training
            self.model.eval()
            self.logit, self.label = self._forward()
            self.model.train(was_training)
        return self.label[:b], self.logit[:b]
