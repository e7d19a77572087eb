"""Prefetching batch loader: the role of the reference's ``DataLoaderX`` (src/tools/data_loader.py:8-10: a
``torch.utils.data.DataLoader`` whose iterator runs in a ``BackgroundGenerator`` thread) for a loop whose step
takes milliseconds.

What the reference does per batch on the critical path of ``VQA.train`` (src/vqa/vqacpv2.py:164-171): collate
B python items (each a fresh fp32 array out of an h5 group), ``.cuda()`` four tensors one after the other on the
compute stream, then tokenise B strings inside ``model.forward``.  Here a producer thread
  1. assembles batch k + 1 .. k + depth straight from the memory-mapped shard into PINNED host buffers
     (``dataset.collate``: row copies, no per-item objects; features are already bf16),
  2. tokenises its questions through the cached ``SentenceBatcher`` (ids / mask / segment ids, one pinned tensor),
  3. queues ALL host-to-device copies of the batch on a copy stream into a ring of device buffers and records an
     event,
while the consumer trains on batch k.  Iterating yields the reference's tuple
``(ques_id, feats, boxes, sent, target, adj)`` -- with ``device`` set the tensors are device tensors whose copies
the current stream has been made to wait for (the trainer's own ``.cuda()`` calls become no-ops), and ``sent`` is
the ``(input_ids, input_mask, segment_ids)`` triple the encoder accepts in place of strings when a batcher was
given.  A slot of the ring is rewritten only after the consumer has asked for the next batch AND, on the device side,
the work it queued on the slot's device buffer has passed an event (the next copy into it waits for that event on the
copy stream), AND, on the host side, the slot's previous host-to-device copy has actually RUN (the producer blocks on
that copy's event before it writes the pinned buffer again) -- however far the host runs ahead of the GPU.
"""
import queue
import threading

import numpy as np
import torch


def _packed(spec, device, pin):
    """one flat byte buffer + typed views {name: tensor} at 256-byte aligned offsets (``spec``: name -> (shape, dtype))"""
    offs, total = {}, 0
    for k, (shape, dt) in spec.items():
        n = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
        offs[k] = (total, n)
        total += (n + 255) // 256 * 256
    flat = torch.zeros(total, dtype=torch.uint8, device=device) if device is not None else (
        torch.zeros(total, dtype=torch.uint8).pin_memory() if pin else torch.zeros(total, dtype=torch.uint8))
    return flat, {k: flat[o:o + n].view(spec[k][1]).view(spec[k][0]) for k, (o, n) in offs.items()}


def packed_like(spec, device):
    """a DEVICE buffer in the layout of the loader's pinned slots (``DataLoaderX.spec``): ``engine.CapturedTrainer``
    keeps its static input buffers as views of one, so that a batch is handed over by ONE host-to-device copy"""
    return _packed(spec, device, False)


class DataLoaderX:
    def __init__(self, dataset, batch_size, shuffle=False, drop_last=False, device=None, batcher=None, depth=3, seed=0,
                 epochs=1, handover="ring"):
        """``dataset``: object with ``__len__``, ``alloc(batch_size, pin)`` and ``collate(items, out)``
        (vqa.vqacpv2_data.VQATorchDataset); ``batcher``: lxrt.entry.SentenceBatcher or None (strings are passed
        through); ``depth``: batches in flight (>= 2); ``epochs``: passes over the data per ``iter`` (None: endless).
        ``handover`` (with a CUDA ``device``):
          "ring"    the producer copies every batch to a ring of DEVICE buffers on its own copy stream; iterating yields
                    device tensors (the reference loop's ``.cuda()`` calls become no-ops);
          "inline"  the producer only fills the PINNED ring; iterating yields pinned host tensors and the consumer copies
                    the slot's flat buffer itself, stream-ordered on the compute stream (``_Iter.flat`` ->
                    ``CapturedTrainer.load_packed``): ONE host-to-device copy per batch, no second stream, no events in
                    the compute stream, no device-to-device hand-over.  Measured on MI355X (tools/exp_loader_variants.py,
                    32 samples, 5.2 MB per batch): a copy stream beside the replayed graphs costs the step +0.36 ms
                    (+0.52 with the hand-over copies), the same bytes copied in stream order +0.2 ms."""
        if depth < 2:
            raise ValueError("depth >= 2: one batch is consumed while the next is produced")
        if handover not in ("ring", "inline"):
            raise ValueError("handover: 'ring' or 'inline'")
        self.handover = handover
        self.ds, self.B, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last
        self.device = torch.device(device) if device is not None else None
        if self.device is not None and self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.batcher, self.depth, self.seed, self.epochs = batcher, depth, seed, epochs
        self._epoch = 0
        pin = self.device is not None and self.device.type == "cuda"
        self.dev = None
        if not pin:
            self.host = [dataset.alloc(batch_size, False) for _ in range(depth)]
            return
        # ONE pinned buffer and ONE device buffer per slot, the fields of a batch are typed views at 256-byte offsets:
        # a batch crosses PCIe as one copy (one DMA command, one event) instead of one per field
        spec = {k: (tuple(v.shape), v.dtype) for k, v in dataset.alloc(batch_size, False).items()}
        if batcher is not None:
            spec["ids"] = ((3, batch_size, batcher.T), torch.long)
        self.spec = spec
        self.host_flat, self.dev_flat, self.host, self.dev = [], [], [], []
        for _ in range(depth):
            hf, hv = _packed(spec, None, True)
            self.host_flat.append(hf)
            self.host.append(hv)
            if handover == "ring":
                df, dv = _packed(spec, self.device, False)
                self.dev_flat.append(df)
                self.dev.append(dv)
        self._copied = [None] * depth  # per slot: event of the last H2D copy out of its pinned buffer (all iterators)
        if handover == "inline":
            self.dev = None
            return
        self.copy_stream = torch.cuda.Stream(device=self.device)
        # the device buffers were zero-filled on the CURRENT stream just now: the first copies into them must not
        # overtake that fill (seen: a busy compute stream ran the fill after the first two batches had landed)
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))

    def __len__(self):
        n = len(self.ds)
        return n // self.B if self.drop_last else (n + self.B - 1) // self.B

    def _batches(self):
        e = 0
        while self.epochs is None or e < self.epochs:
            n = len(self.ds)
            order = np.random.default_rng([self.seed, self._epoch]).permutation(n) if self.shuffle else np.arange(n)
            self._epoch += 1
            for s in range(0, n, self.B):
                items = order[s:s + self.B]
                if len(items) < self.B and self.drop_last:
                    break
                yield items
            e += 1

    def __iter__(self):
        return _Iter(self)


class _Iter:
    def __init__(self, L):
        self.L = L
        self.q = queue.Queue(maxsize=L.depth - 1)
        self.free = [threading.Event() for _ in range(L.depth)]   # slot may be rewritten (host side)
        for f in self.free:
            f.set()
        self.released = [None] * L.depth                         # CUDA event: consumer's work on the slot is queued
        self.copied = getattr(L, "_copied", [None] * L.depth)    # CUDA event: the slot's last host-to-device copy
        self.stop = False
        self.err = None
        self.held = None
        self.flat = None
        self.rows = 0
        self._marked = False
        self.t = threading.Thread(target=self._produce, daemon=True)
        self.t.start()

    def _produce(self):
        L = self.L
        try:
            if L.device is not None and L.device.type == "cuda":
                torch.cuda.set_device(L.device)
            slot = 0
            for items in L._batches():
                self.free[slot].wait()
                if self.stop:
                    return
                self.free[slot].clear()
                host = L.host[slot]
                if self.copied[slot] is not None:
                    # The previous copy OUT of this pinned buffer may still be queued: it waits on the copy stream for
                    # the consumer's event, and with graph replays and no host synchronisation the host runs many
                    # batches ahead of the GPU.  Writing the buffer now would hand the GPU a batch mixed from two
                    # sample sets.  Block THIS (producer) thread until that copy has run.
                    self.copied[slot].synchronize()
                ids, sents, B = L.ds.collate(items, host)
                sent = sents
                ev = None
                if L.dev is not None:
                    rel = self.released[slot]
                    dev = L.dev[slot]
                    if L.batcher is not None:  # token ids into this slot's pinned buffer (host copy of a few KB)
                        host["ids"].numpy()[:, :B] = L.batcher.host_batch(sents).numpy()
                        sent = (dev["ids"][0, :B], dev["ids"][1, :B], dev["ids"][2, :B])
                    with torch.cuda.stream(L.copy_stream):
                        if rel is not None:
                            L.copy_stream.wait_event(rel)  # the consumer's kernels on this slot's buffers are done
                        L.dev_flat[slot].copy_(L.host_flat[slot], non_blocking=True)  # the whole batch: one copy
                        ev = torch.cuda.Event()
                        ev.record(L.copy_stream)
                    self.copied[slot] = ev
                    out = {k: v[:B] for k, v in dev.items() if k != "ids"}
                elif getattr(L, "handover", "ring") == "inline" and "ids" in host:
                    # pinned slot, token ids included: the consumer ships the whole slot with one copy (``flat``)
                    host["ids"].numpy()[:, :B] = L.batcher.host_batch(sents).numpy()
                    sent = (host["ids"][0, :B], host["ids"][1, :B], host["ids"][2, :B])
                    out = {k: v[:B] for k, v in host.items() if k != "ids"}
                else:
                    out = {k: v[:B] for k, v in host.items()}
                    if L.batcher is not None:
                        hb = L.batcher.host_batch(sents)
                        sent = (hb[0].clone(), hb[1].clone(), hb[2].clone())
                self.q.put((slot, ids, out, sent, ev))
                slot = (slot + 1) % L.depth
        except BaseException as ex:  # surfaced in the consumer
            self.err = ex
        finally:
            self.q.put(None)

    def _release(self):
        if self.held is not None:
            slot = self.held
            if self.L.dev is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.L.device))
                self.released[slot] = ev
            elif self.flat is not None and not self._marked:
                # "inline" hand-over and the consumer did not say when its copy was queued: whatever it queued on the
                # current stream out of this pinned slot lies in front of an event recorded now
                self.mark_copied()
            self.free[slot].set()
            self.held = None

    def __iter__(self):
        return self

    def __next__(self):
        self._release()  # the previous batch has been consumed (its work is queued on the current stream)
        item = self.q.get()
        if item is None:
            if self.err is not None:
                raise self.err
            raise StopIteration
        slot, ids, out, sent, ev = item
        if ev is not None:
            torch.cuda.current_stream(self.L.device).wait_event(ev)
        self.held = slot
        # "inline" hand-over: the slot's flat pinned buffer, for the consumer's own single copy (then ``mark_copied``)
        self.flat = self.L.host_flat[slot] if getattr(self.L, "handover", "ring") == "inline" else None
        self.rows = len(ids)  # samples of THIS batch (a short last batch fills only the head of the slot)
        self._marked = False
        if "adj" in out:
            return ids, out["feats"], out["boxes"], sent, out["target"], out["adj"]
        return ids, out["feats"], out["boxes"], sent

    def mark_copied(self, event=None):
        """"inline" hand-over: the consumer has queued its host-to-device copy of the current batch's pinned slot;
        ``event`` (default: one recorded now on the current stream) tells the producer when the slot may be rewritten"""
        if self.held is not None and self.L.device is not None and self.L.device.type == "cuda":
            if event is None:
                event = torch.cuda.Event()
                event.record(torch.cuda.current_stream(self.L.device))
            self.copied[self.held] = event
            self._marked = True

    def close(self):
        self.stop = True
        for f in self.free:
            f.set()
        try:
            while self.q.get_nowait() is not None:
                pass
        except queue.Empty:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
