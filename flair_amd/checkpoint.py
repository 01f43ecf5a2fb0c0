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
    """Load a reference checkpoint (a plain state dict, or a dict holding one under ``params_ema`` /
    ``state_dict`` / ``model``) into ``model``; returns torch's missing / unexpected key report."""
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict):
        for k in ("params_ema", "state_dict", "model"):
            if k in obj and isinstance(obj[k], dict):
                obj = obj[k]
                break
    if not isinstance(obj, dict) or not all(isinstance(v, torch.Tensor) for v in obj.values()):
        raise ValueError(f"{path}: not a tensor state dict")
    report = model.load_state_dict(obj, strict=strict)
    if hasattr(model, "_packed_key"):
        model._packed_key = None           # kernel-native weight copies are stale now
    return report
