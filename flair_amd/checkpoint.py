"""Checkpoint ingest: reference-format ``.pt`` state dicts -> the HIP model.

The reference ships fp32 ``state_dict`` pickles (``scripts/video_sample.py:165-171,330``:
``model.load_state_dict(torch.load(CKPT_PATH[task], map_location="cpu"))``).  The models here keep the
reference's parameter names and shapes, so such a file loads unchanged; the kernel-native copies
(bf16 ``[Cout][taps][Cin]`` conv weights, tap-major offset convolutions, the batched embedding
matrix, ...) are rebuilt from the fp32 master on the next forward (``_ensure_packed``).

Files are read with ``weights_only=True``: nothing in the pickle is executed.
"""
import torch


def load_reference_checkpoint(model, path, strict=True):
    """Load a reference checkpoint (a plain state dict, or a dict holding one under ``params_ema`` / ``params`` /
    ``state_dict`` / ``model``) into ``model``; returns torch's missing / unexpected key report."""
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict):
        for k in ("params_ema", "params", "state_dict", "model"):
            if k in obj and isinstance(obj[k], dict):
                obj = obj[k]
                break
    if not isinstance(obj, dict) or not all(isinstance(v, torch.Tensor) for v in obj.values()):
        raise ValueError(f"{path}: not a tensor state dict")
    report = model.load_state_dict(obj, strict=strict)
    if hasattr(model, "_packed_key"):
        model._packed_key = None           # kernel-native weight copies are stale now
    return report


# ------------------------------------------------------------------ kernel-native weight blob
# What the kernels read is not the fp32 state dict but its one-time repack (``_ensure_packed``): conv weights
# as [Cout][taps][Cin] in the compute dtype (offset convolutions tap-major), the batched embedding matrix,
# folded temporal-attention codes, f32 biases / norm parameters.  ``export_packed`` lays all of it out in ONE
# flat device buffer (256-byte aligned pieces) plus a small description; ``import_packed`` rebuilds the models'
# packed attributes as views into such a buffer.  This is the unit the multi-GPU start-up ships
# (flair_amd.parallel.broadcast_packed_weights): 0.83 GB for unet_new.UNetModel in
# bf16 instead of the 1.65 GB fp32 master copy plus a repack on every rank.
PACKED_ATTRS = ("_pk", "_pk_w", "_pk_b", "_emb_w", "_emb_b", "_te", "_head_w", "_head_b", "_head_g", "_head_be",
                "_plain", "_silu", "_mlp", "_fin")
INT_ATTRS = ("film_off", "plain_off", "silu_off")
_ALIGN = 256


def _flatten(obj, tensors):
    """Nested dict / list / tuple of tensors -> the same structure with ('T', index) leaves."""
    if isinstance(obj, torch.Tensor):
        tensors.append(obj)
        return ("T", len(tensors) - 1)
    if isinstance(obj, dict):
        return ("D", {k: _flatten(v, tensors) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return ("L" if isinstance(obj, list) else "U", [_flatten(v, tensors) for v in obj])
    return ("V", obj)                                   # plain python value (ints, None)


def _rebuild(node, tensors):
    kind, val = node
    if kind == "T":
        return tensors[val]
    if kind == "D":
        return {k: _rebuild(v, tensors) for k, v in val.items()}
    if kind in ("L", "U"):
        seq = [_rebuild(v, tensors) for v in val]
        return seq if kind == "L" else tuple(seq)
    return val


def export_packed(model, device=None):
    """Pack ``model`` (if needed) and return (meta, blob): blob is one flat uint8 tensor on the model's device
    holding every kernel-native tensor, meta a picklable description (a few hundred KB).  Tensors that several
    attributes share (views of one storage) are stored once per attribute; the total is dominated by the conv packs."""
    if device is None:
        device = next(model.parameters()).device
    model._ensure_packed(device)
    tensors, entries = [], []
    for name, mod in model.named_modules():
        for attr in PACKED_ATTRS:
            if attr in mod.__dict__:
                entries.append((name, attr, _flatten(mod.__dict__[attr], tensors)))
        ints = {a: int(mod.__dict__[a]) for a in INT_ATTRS if a in mod.__dict__}
        if ints:
            entries.append((name, "__ints__", ("V", ints)))
    layout, off = [], 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        layout.append((off, tuple(t.shape), str(t.dtype).replace("torch.", "")))
        off = (off + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    blob = torch.zeros(off, dtype=torch.uint8, device=device)
    for t, (o, shape, _) in zip(tensors, layout):
        n = t.numel() * t.element_size()
        blob[o:o + n].copy_(t.detach().contiguous().reshape(-1).view(torch.uint8))
    meta = dict(entries=entries, layout=layout, nbytes=off, dtype=str(model.dtype), cls=type(model).__name__)
    return meta, blob


def import_packed(model, meta, blob):
    """Give ``model`` the kernel-native weights of ``blob`` (views, no copy).  Its fp32 parameters are left as
    they are: they are not read again unless the dtype changes or a state dict is loaded (which repacks)."""
    if meta["cls"] != type(model).__name__:
        raise ValueError(f"packed blob is for {meta['cls']}, not {type(model).__name__}")
    tensors = []
    for off, shape, dt in meta["layout"]:
        dtype = getattr(torch, dt)
        n = 1
        for d in shape:
            n *= d
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        tensors.append(blob[off:off + nbytes].view(dtype).view(shape))
    mods = dict(model.named_modules())
    for name, attr, node in meta["entries"]:
        mod = mods[name]
        if attr == "__ints__":
            for k, v in node[1].items():
                setattr(mod, k, v)
        else:
            mod.__dict__[attr] = _rebuild(node, tensors)
    model.dtype = {"torch.bfloat16": torch.bfloat16, "torch.float32": torch.float32}[meta["dtype"]]
    model._packed_key = (model.dtype, blob.device)
    model._packed_blob = blob                       # keeps the storage alive
    if hasattr(model, "_flow_cache"):
        model._flow_cache = {}
    if hasattr(model, "_graphs"):
        model._graphs = {}
