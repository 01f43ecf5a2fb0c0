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
