"""Frame file I/O either side of the sampling path, and the streaming form of the window loop.

Mirror of the reference's ``scripts/video_sample.py:334-345`` (frame discovery with the glob
``*.[jJpP][pPnN][gG]`` in natural order, ``cv2.imread`` -> RGB -> float / 255, stacked to
``(1, N, 3, h, w)``) and ``:487-492`` (``(x * 255).byte()`` -> ``{i:04d}.png``), SURVEY.md section 8f
"next" row 2.  The reference decodes every frame up front, samples window after window, keeps all
results on the host and writes them at the end; here the three stages overlap:

  * a reader thread decodes the frames of window ``w + 1`` (PIL; ``cv2`` is not a dependency) into
    pinned host memory and uploads them on a side HIP stream while window ``w`` is being sampled;
  * finished frames leave the GPU with an asynchronous device-to-host copy and are PNG-encoded by a
    writer thread while the next window samples.

Windows of one video stay sequential (``prev_recon`` couples them); independent videos are the
clip-parallel unit (``flair_amd.parallel``).  Host-side glue only: no arithmetic of the hot path
happens here.
"""
import os
import queue
import re
import threading

import numpy as np
import torch

from . import video

_FRAME_RE = re.compile(r".*\.[jJpP][pPnN][gG]$")          # the reference's glob: jpg / png / jng / pnG ...


def natural_key(name):
    """natsort's default ordering of the reference (``natsorted(video_path.glob(...))``): digit runs
    compare as integers, the rest as text."""
    return [int(tok) if tok.isdigit() else tok for tok in re.split(r"(\d+)", os.path.basename(str(name)))]


def list_frames(video_path):
    """Frame files of a directory in the reference's order (video_sample.py:334)."""
    names = [n for n in os.listdir(video_path) if _FRAME_RE.match(n)]
    return [os.path.join(str(video_path), n) for n in sorted(names, key=natural_key)]


def decode_frame(path):
    """One frame file -> (3, h, w) uint8 RGB (what ``cv2.cvtColor(cv2.imread(p), COLOR_BGR2RGB)`` yields)."""
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(arr.transpose(2, 0, 1))


def read_frames(paths, pin=False):
    """Frame files -> (1, N, 3, h, w) float32 in [0, 1] on the host (video_sample.py:337-345)."""
    frames = np.stack([decode_frame(p) for p in paths])
    out = torch.from_numpy(frames).float().div_(255.0).unsqueeze(0)
    return out.pin_memory() if pin and torch.cuda.is_available() else out


def to_bytes(frames01):
    """(N, 3, H, W) float in [0, 1] -> (N, H, W, 3) uint8 exactly as ``(x * 255).byte()`` (truncation,
    video_sample.py:489) followed by the ``t c h w -> t h w c`` rearrange."""
    return (frames01.float() * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def write_frame(path, hwc_u8):
    from PIL import Image
    Image.fromarray(np.asarray(hwc_u8), mode="RGB").save(path, format="PNG")


class _Writer(threading.Thread):
    """Encodes finished frames in the background (PNG compression is host work that would otherwise sit
    between two windows)."""

    def __init__(self, output_path):
        super().__init__(daemon=True)
        self.q = queue.Queue()
        self.output_path = str(output_path)
        self.error = None
        os.makedirs(self.output_path, exist_ok=True)
        self.start()

    def run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            first, host, event = item
            try:
                if event is not None:
                    event.synchronize()                    # the asynchronous D2H copy has landed
                for i, frame in enumerate(host.numpy()):
                    write_frame(os.path.join(self.output_path, f"{first + i:04d}.png"), frame)
            except Exception as exc:                       # surfaced by close()
                self.error = exc

    def submit(self, first, frames01_dev):
        """frames01_dev: (n, 3, H, W) float on the GPU (or host)."""
        u8 = to_bytes(frames01_dev)
        if u8.is_cuda:
            host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(u8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.q.put((first, host, ev))
        else:
            self.q.put((first, u8, None))

    def close(self):
        self.q.put(None)
        self.join()
        if self.error is not None:
            raise self.error


def iter_windows(paths, device, length=video.FRAME_SLICE_LEN, overlap=video.OVERLAP):
    """Yield ``(indices, degraded01)`` per window with ``degraded01`` = (1, T, 3, h, w) float in [0, 1] on
    ``device``: the frames of the NEXT window are decoded and uploaded (pinned memory, side stream) while the
    caller works on the current one.  Frames shared by two windows are decoded once."""
    wins = video.window_indices(len(paths), length, overlap)
    cache, q = {}, queue.Queue(maxsize=2)
    on_gpu = torch.device(device).type == "cuda"
    side = torch.cuda.Stream(device=device) if on_gpu else None

    def produce():
        try:
            for idx in wins:
                for i in idx:
                    if i not in cache:
                        cache[i] = decode_frame(paths[i])
                host = torch.from_numpy(np.stack([cache[i] for i in idx]))
                for i in [k for k in cache if k < idx[-1] - overlap]:   # older frames are never needed again
                    del cache[i]
                if on_gpu:
                    host = host.pin_memory()
                    with torch.cuda.stream(side):
                        dev_u8 = host.to(device, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    q.put((idx, dev_u8, ev, host))
                else:
                    q.put((idx, host, None, host))
            q.put(None)
        except Exception as exc:
            q.put(exc)

    threading.Thread(target=produce, daemon=True).start()
    while True:
        item = q.get()
        if item is None:
            return
        if isinstance(item, Exception):
            raise item
        idx, u8, ev, _keep = item
        if ev is not None:
            cur = torch.cuda.current_stream(device)
            cur.wait_event(ev)
            u8.record_stream(cur)        # allocated on the side stream, read here: keep the block until this read is done
        yield idx, (u8.float() / 255.0).unsqueeze(0)


def restore_video_files(task, video_path, output_path, model, diffusion, restore_fn_for, *, size, device,
                        length=video.FRAME_SLICE_LEN, overlap=video.OVERLAP, **kw):
    """``scripts/video_sample.py:334-492`` end to end: frame files in, restored ``{i:04d}.png`` out, with
    decode / upload / sampling / download / encode overlapped.  ``kw`` goes to ``video.restore_window``
    (aux_model, vsrpp_weights_fn, hp, tau, t_start, noise_fn, q_noise_fn).  Returns the number of frames written."""
    paths = list_frames(video_path)
    writer = _Writer(output_path)
    prev_recon, written = None, 0
    try:
        for wi, (idx, degraded01) in enumerate(iter_windows(paths, device, length, overlap)):
            keep, prev_recon = video.restore_window(task, degraded01, model, diffusion, restore_fn_for, size=size,
                                                    prev_recon=prev_recon, overlap=overlap, window_index=wi, **kw)
            writer.submit(written, keep)
            written += keep.shape[0]
    finally:
        writer.close()
    return written
