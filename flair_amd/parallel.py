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


def broadcast_packed_weights(model, src=0, chunk_bytes=512 << 20):
    """Start-up weight distribution in the kernels' own format: ``src`` packs once (bf16 [Cout][taps][Cin] conv
    weights, batched embedding matrix, ...: flair_amd.checkpoint.export_packed) and broadcasts ONE flat blob
    (0.83 GB for unet_new.UNetModel in bf16) in ``chunk_bytes`` messages -- RCCL over xGMI on the GPUs, gloo in
    the CPU tests; the other ranks attach views of it to their model and never repack.  Replaces the reference's
    pickled-chunk ``load_state_dict`` + per-parameter ``sync_params`` (dist_util.py:40-79) and halves the bytes of
    ``broadcast_weights`` (which ships the fp32 masters).  Returns (seconds, blob bytes)."""
    from . import checkpoint
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0.0, 0
    t0 = time.perf_counter()
    device = next(model.parameters()).device
    rank = dist.get_rank()
    if rank == src:
        meta, blob = checkpoint.export_packed(model, device)
        box = [meta]
    else:
        box = [None]
    dist.broadcast_object_list(box, src=src)            # the description: a few hundred KB of python
    meta = box[0]
    if rank != src:
        blob = torch.empty(meta["nbytes"], dtype=torch.uint8, device=device)
    for off in range(0, meta["nbytes"], chunk_bytes):
        dist.broadcast(blob[off:off + chunk_bytes], src=src)
    if rank != src:
        checkpoint.import_packed(model, meta, blob)
    if blob.is_cuda:
        torch.cuda.synchronize()
    return time.perf_counter() - t0, int(meta["nbytes"])


def gather_results(local_items, dst=0):
    """Collect per-rank python results (timings, file names) on ``dst`` (host-side gather)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_items]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local_items, out, dst=dst)
    return out
