"""Contiguous, memory-mapped shard format for the per-image inputs of the training step.

The reference keeps one h5 group per image (``features [36, 2048] f32``, ``boxes [36, 4] f32``,
src/vqa/vqacpv2_data.py:98-110) plus a second h5 file of ``[36, 36] f32`` adjacencies (:76-79, :126):
three random reads and 295 KB of fp32 features per sample, normalised and copied in Python per item.  At
millisecond step times that loader is the critical path.  A shard stores the
same information the way the GPU consumes it:

    <name>.xgs      4096-byte header (magic, JSON) followed by 4096-byte-aligned arrays
        feats   [n, N, F]  bf16   region features, already in the storage type of the encoder's first GEMM
                                  (``ShardWriter(feat_dtype="float32")`` keeps the reference's exact fp32 values instead,
                                  for parity runs in fp32 execution: twice the bytes, rounded to bf16 only when a batch
                                  is gathered into a bf16 buffer)
        boxes   [n, N, 4]  f32    ALREADY normalised to 0..1 by (img_w, img_h), range-checked once at write time
                                  (vqacpv2_data.py:113-117)
        adj     [n, N, N]  f32    attribute-class cosine adjacency (absent for splits that have none)
        img_id  [n]        int64 (numeric ids) -- string ids live in the header's ``ids`` list

Rows are fixed-size and contiguous, so a batch is ``n`` row copies out of one mapping (or one slice copy for
a sequential batch) straight into a pinned buffer: no per-item Python objects, no fp32 -> bf16 cast kernel, half
the PCIe bytes.  ``ShardReader`` only maps the file; nothing is read until rows are touched.
"""
import json
import os

import numpy as np
import torch

MAGIC = b"XGGMSHARD1\n"
HEADER = 4096
ALIGN = 4096


def _bf16_bits(x):
    """fp32 ndarray -> uint16 bit patterns of the round-to-nearest-even bf16 values (torch's rounding)"""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def normalize_boxes(boxes, img_w, img_h):
    """the reference's box normalisation and its range check (src/vqa/vqacpv2_data.py:108-117)"""
    b = np.array(boxes, dtype=np.float32, copy=True)
    b[:, (0, 2)] /= img_w
    b[:, (1, 3)] /= img_h
    np.testing.assert_array_less(b, 1 + 1e-5)
    np.testing.assert_array_less(-b, 0 + 1e-5)
    return b


class ShardWriter:
    """collects per-image records and writes one shard.  ``add`` takes what the reference's h5 group and info
    json hold for an image; ``adj`` may be None for every record (splits without adjacency)."""

    def __init__(self, path, n_objects=36, feat_dim=2048, feat_dtype="bf16"):
        if feat_dtype not in ("bf16", "float32"):
            raise ValueError("feat_dtype: 'bf16' or 'float32', got %r" % (feat_dtype,))
        self.path, self.N, self.F, self.fdt = path, n_objects, feat_dim, feat_dtype
        self.ids, self.feats, self.boxes, self.adj = [], [], [], []

    def add(self, img_id, feats, boxes, img_w, img_h, adj=None):
        feats = np.asarray(feats, dtype=np.float32)
        if feats.shape != (self.N, self.F) or np.shape(boxes) != (self.N, 4):
            raise ValueError("record %r: features %s / boxes %s, expected (%d, %d) / (%d, 4)"
                             % (img_id, feats.shape, np.shape(boxes), self.N, self.F, self.N))
        if adj is not None and np.shape(adj) != (self.N, self.N):
            raise ValueError("record %r: adjacency %s" % (img_id, np.shape(adj)))
        if self.ids and (adj is None) != (not self.adj):
            raise ValueError("either every record of a shard has an adjacency or none has")
        nb = normalize_boxes(boxes, img_w, img_h)  # may raise: nothing of a bad record is kept
        fb = _bf16_bits(feats) if self.fdt == "bf16" else feats.copy()
        self.ids.append(img_id)
        self.feats.append(fb)
        self.boxes.append(nb)
        if adj is not None:
            self.adj.append(np.asarray(adj, dtype=np.float32))

    def close(self):
        n = len(self.ids)
        if n == 0:
            raise ValueError("empty shard")
        arrays = [("feats", np.stack(self.feats)), ("boxes", np.stack(self.boxes))]
        if self.adj:
            if len(self.adj) != n:
                raise ValueError("either every record of a shard has an adjacency or none has")
            arrays.append(("adj", np.stack(self.adj)))
        numeric = all(isinstance(i, (int, np.integer)) for i in self.ids)
        if numeric:
            arrays.append(("img_id", np.asarray(self.ids, dtype=np.int64)))
        off, table = HEADER, {}
        for name, a in arrays:
            table[name] = {"offset": off, "shape": list(a.shape),
                           "dtype": "bf16" if (name == "feats" and self.fdt == "bf16") else str(a.dtype)}
            off = (off + a.nbytes + ALIGN - 1) // ALIGN * ALIGN
        head = {"n": n, "n_objects": self.N, "feat_dim": self.F, "arrays": table,
                "ids": None if numeric else [str(i) for i in self.ids]}
        blob = MAGIC + json.dumps(head).encode()
        if len(blob) > HEADER and not numeric:
            # long string-id lists do not fit the fixed header: they go behind the arrays
            head["ids"], head["ids_offset"], head["ids_bytes"] = None, off, 0
            ids_blob = json.dumps([str(i) for i in self.ids]).encode()
            head["ids_bytes"] = len(ids_blob)
            blob = MAGIC + json.dumps(head).encode()
        else:
            ids_blob = None
        if len(blob) > HEADER:
            raise ValueError("shard header too large")
        tmp = self.path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(blob.ljust(HEADER, b"\0"))
            for name, a in arrays:
                f.seek(table[name]["offset"])
                f.write(a.tobytes())
            if ids_blob is not None:
                f.seek(off)
                f.write(ids_blob)
            f.truncate(max(f.tell(), off))
        os.replace(tmp, self.path)
        return self.path


class ShardReader:
    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            head = f.read(HEADER)
            if not head.startswith(MAGIC):
                raise ValueError("%s is not an xggm shard" % path)
            self.meta = json.loads(head[len(MAGIC):].rstrip(b"\0").decode())
            ids = self.meta.get("ids")
            if ids is None and "ids_offset" in self.meta:
                f.seek(self.meta["ids_offset"])
                ids = json.loads(f.read(self.meta["ids_bytes"]).decode())
        self.n, self.N, self.F = self.meta["n"], self.meta["n_objects"], self.meta["feat_dim"]
        t = self.meta["arrays"]

        def mm(name, dtype):
            return np.memmap(path, mode="r", dtype=dtype, offset=t[name]["offset"], shape=tuple(t[name]["shape"]))

        self.feat_dtype = t["feats"]["dtype"]
        self.feats_bits = mm("feats", np.uint16) if self.feat_dtype == "bf16" else None  # bf16 bit patterns
        self.feats_f = mm("feats", np.float32) if self.feat_dtype != "bf16" else None     # or the exact fp32 values
        self.boxes = mm("boxes", np.float32)
        self.adj = mm("adj", np.float32) if "adj" in t else None
        if ids is None:
            ids = mm("img_id", np.int64).tolist()
        self.ids = ids
        self.row_of = {i: r for r, i in enumerate(ids)}

    def __len__(self):
        return self.n

    def feats_f32(self, row):
        """features of one image as fp32 (exactly the stored values: bf16-rounded, or the reference's own fp32 ones in
        a float32 shard): what ``Dataset.__getitem__`` hands out"""
        if self.feats_f is not None:
            return np.array(self.feats_f[row])
        return (self.feats_bits[row].astype(np.uint32) << 16).view(np.float32)

    def gather(self, rows, out):
        """copy rows into preallocated (pinned) host tensors: out['feats'] bf16 [B,N,F], out['boxes'] f32,
        out['adj'] f32 (when the shard has one).  Sequential row runs are one slice copy each."""
        rows = np.asarray(rows, dtype=np.int64)
        B = len(rows)
        if self.feats_f is not None:  # float32 shard (parity runs): exact rows into an fp32 buffer, rounded into a bf16 one
            fo = out["feats"][:B]
            for b, r in enumerate(rows):
                if fo.dtype == torch.float32:
                    fo[b].numpy()[:] = self.feats_f[r]
                else:
                    fo[b].view(torch.int16).numpy().view(np.uint16)[:] = _bf16_bits(np.array(self.feats_f[r]))
                out["boxes"][b].numpy()[:] = self.boxes[r]
                if self.adj is not None and "adj" in out:
                    out["adj"][b].numpy()[:] = self.adj[r]
            return B
        fb = out["feats"][:B].view(torch.int16).numpy().view(np.uint16)
        if B and np.all(np.diff(rows) == 1):
            s = slice(int(rows[0]), int(rows[0]) + B)
            fb[:] = self.feats_bits[s]
            out["boxes"][:B].numpy()[:] = self.boxes[s]
            if self.adj is not None and "adj" in out:
                out["adj"][:B].numpy()[:] = self.adj[s]
        else:
            # one contiguous row copy per sample (147 KB of features each): numpy releases the GIL inside these
            # copies, which a fancy-indexed np.take over the mapping does not (measured 3.7 ms vs 0.7 ms per 32 rows,
            # all of it with the GIL held -- the training thread stalls behind it)
            bx, ad = out["boxes"].numpy(), (out["adj"].numpy() if self.adj is not None and "adj" in out else None)
            for b, r in enumerate(rows):
                fb[b] = self.feats_bits[r]
                bx[b] = self.boxes[r]
                if ad is not None:
                    ad[b] = self.adj[r]
        return B


def from_h5(obj_h5_path, info_json_path, out_path, adj_h5_path=None):
    """convert the reference's files (``*_obj36.h5`` + ``*_obj36_info.json`` [+ ``*_obj36_adj_v2.h5``]) to a shard;
    needs h5py, which the offline image lacks -- the conversion runs wherever the datasets live."""
    import h5py
    info = {d["img_id"]: d for d in json.load(open(info_json_path))}
    adj = h5py.File(adj_h5_path, "r") if adj_h5_path else None
    with h5py.File(obj_h5_path, "r") as h:
        first = h[next(iter(h.keys()))]
        w = ShardWriter(out_path, n_objects=first["features"].shape[0], feat_dim=first["features"].shape[1])
        for key in h.keys():
            d = info.get(key, info.get(int(key) if key.isdigit() else key))
            w.add(d["img_id"], h[key]["features"][:], h[key]["boxes"][:], d["img_w"], d["img_h"],
                  adj[key][:] if adj is not None else None)
    return w.close()
