"""Clip-parallel multi-GPU support: one process per MI355X, no communication inside a clip.

Within a clip everything is coupled (GroupNorm statistics span all frames, BasicVSR++ is a
recurrence over frames, temporal attention, ``prev_recon`` between windows of one video;
SURVEY.md section 8e), so the unit of parallelism is the clip: rank r restores clips
r, r+W, r+2W, ...  The only collective is the start-up distribution of the weights, which
replaces the reference's ``dist_util.load_state_dict`` / ``sync_params`` (pickled chunks
through ``broadcast_object_list`` plus one ``dist.broadcast`` per parameter,
guided_diffusion/dist_util.py:40-79: 1638 small messages): the parameters are flattened into
a few large buffers and broadcast from the source rank -- RCCL over xGMI with
``backend="nccl"``, gloo on CPU in the tests.
"""
import time

import torch
import torch.distributed as dist


def clips_for_rank(num_clips, rank, world_size):
    """Round-robin partition of independent clips (no clip is split across GPUs)."""
    return list(range(rank, num_clips, world_size))


def broadcast_weights(model, src=0, bucket_bytes=256 << 20):
    """Make every rank's parameters and buffers equal to ``src``'s.  Tensors are packed per
    dtype into flat buckets of <= ``bucket_bytes`` so that each collective moves a large
    message over the (per-link bound) xGMI fabric.  Returns the wall time in seconds."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0.0
    t0 = time.perf_counter()
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, group in by_dtype.items():
        bucket, size = [], 0
        for t in group + [None]:
            if t is None or (bucket and size + t.numel() * t.element_size() > bucket_bytes):
                flat = torch.cat([b.reshape(-1) for b in bucket])
                dist.broadcast(flat, src=src)
                off = 0
                for b in bucket:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()
                bucket, size = [], 0
            if t is not None:
                bucket.append(t)
                size += t.numel() * t.element_size()
    if tensors and tensors[0].is_cuda:
        torch.cuda.synchronize()
    if hasattr(model, "_packed_key"):
        model._packed_key = None      # kernel-native weight copies must be rebuilt
    return time.perf_counter() - t0


def gather_results(local_items, dst=0):
    """Collect per-rank python results (timings, file names) on ``dst`` (host-side gather)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_items]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local_items, out, dst=dst)
    return out
