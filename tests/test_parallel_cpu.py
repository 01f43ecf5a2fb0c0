"""CPU (gloo, world_size 2): the N>1 path -- weight broadcast + clip partition + gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flair_amd import parallel
    torch.manual_seed(100 + rank)                       # ranks start with DIFFERENT weights
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.GroupNorm(2, 8), torch.nn.Linear(8, 5))
    model.register_buffer("mean", torch.full((3,), float(rank)))
    model.half_param = torch.nn.Parameter(torch.randn(7).to(torch.bfloat16))
    model._packed_key = "stale"
    parallel.broadcast_weights(model, src=0, bucket_bytes=256)      # tiny buckets: several collectives
    flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()] + [model.mean])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    clips = parallel.clips_for_rank(5, rank, world)
    res = parallel.gather_results({"rank": rank, "clips": clips}, dst=0)
    if rank == 0:
        q.put((all(torch.equal(gathered[0], g) for g in gathered), model._packed_key, res))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_partition_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, packed_key, res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same, "ranks disagree after broadcast_weights"
    assert packed_key is None, "kernel-native weight packs must be invalidated after a broadcast"
    assert sorted(c for r in res for c in r["clips"]) == [0, 1, 2, 3, 4]
    assert res[0]["clips"] == [0, 2, 4] and res[1]["clips"] == [1, 3]


def test_single_process_is_noop():
    from flair_amd import parallel
    m = torch.nn.Linear(2, 2)
    assert parallel.broadcast_weights(m) == 0.0
    assert parallel.clips_for_rank(3, 0, 1) == [0, 1, 2]
    assert parallel.gather_results([1]) == [[1]]


def _packed_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flair_amd import checkpoint, parallel
    from flair_amd.guided_diffusion.unet_new import UNetModel
    from tests.test_gpu_unet import SMALL
    torch.manual_seed(100 + rank)                       # ranks start with DIFFERENT weights
    model = UNetModel(**SMALL)
    model.convert_to_fp16()                             # the kernels' bf16 format is what travels
    secs, nbytes = parallel.broadcast_packed_weights(model, src=0, chunk_bytes=64 << 20)
    # every rank now holds the SAME kernel-native tensors (views into its copy of the blob) and does not repack
    assert model._packed_key == (torch.bfloat16, torch.device("cpu"))
    _, blob = checkpoint.export_packed(model)           # re-export from the attached views
    digest = torch.tensor([float(blob[::97].double().sum()), float(blob.numel())], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    res = model.input_blocks[1][0]._pk["w1"]
    first = [torch.zeros_like(res.float()) for _ in range(world)]
    dist.all_gather(first, res.float())
    masters = sum(p.numel() * 4 for p in model.parameters())
    if rank == 0:
        q.put((all(torch.equal(gathered[0], g) for g in gathered), torch.equal(first[0], first[1]),
               str(res.dtype), nbytes, masters))
    dist.barrier()
    dist.destroy_process_group()


def test_packed_weight_blob_broadcast_world2():
    """The multi-GPU start-up ships ONE kernel-native blob (bf16 packs) instead of the fp32 masters: after
    broadcast_packed_weights both ranks hold identical packed tensors although their fp32 parameters differ,
    and the blob is about half the masters' size."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same_blob, same_w, dt, nbytes, masters = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert same_blob and same_w and dt == "torch.bfloat16"
    assert 0 < nbytes < 0.6 * masters, (nbytes, masters)
