"""RCCL on the hardware that is available to the tests (ONE MI355X): the path's only collective executed with a
world of size 1 -- the library is loaded, a communicator is built, ncclBroadcast runs on the launch stream and leaves the blob intact.
(The world-size-2 semantics are covered on CPU by tests/test_parallel_cpu.py with gloo; the 8-GPU run is the driver's.)"""
import ctypes
import socket

import pytest
import torch


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


@pytest.mark.gpu
def test_flair_bcast_weights_runs_rccl_on_one_rank(dev):
    from flair_amd import _lib
    try:
        rccl = ctypes.CDLL("librccl.so", mode=ctypes.RTLD_GLOBAL)
    except OSError:
        rccl = ctypes.CDLL("librccl.so.1", mode=ctypes.RTLD_GLOBAL)
    uid = _NcclUniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        g = torch.Generator(device=dev).manual_seed(3)
        blob = torch.randint(0, 255, (3 * (1 << 20) + 17,), dtype=torch.uint8, device=dev, generator=g)
        keep = blob.clone()
        lib = _lib.lib()
        lib.flair_bcast_weights.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        rc = lib.flair_bcast_weights(ctypes.c_void_p(blob.data_ptr()), ctypes.c_size_t(blob.numel()), 0, comm, _lib.stream())
        assert rc == 0, lib.flair_last_error()
        torch.cuda.synchronize()
        assert torch.equal(blob, keep)
        # a root that is not a rank of the communicator is RCCL's error, reported through flair_last_error
        rc = lib.flair_bcast_weights(ctypes.c_void_p(blob.data_ptr()), ctypes.c_size_t(1024), 5, comm, _lib.stream())
        assert rc != 0 and b"ncclBroadcast" in lib.flair_last_error()
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.gpu
def test_torch_distributed_nccl_backend_world_of_one(dev):
    """backend "nccl" IS RCCL on ROCm: the process-group path bench.py / flair_amd.parallel use, on a world of one rank."""
    import torch.distributed as dist
    from flair_amd import parallel
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        x = torch.arange(1 << 16, device=dev, dtype=torch.float32)
        dist.broadcast(x, src=0)
        dist.all_reduce(x)
        torch.cuda.synchronize()
        assert torch.equal(x, torch.arange(1 << 16, device=dev, dtype=torch.float32))
        assert parallel.clips_for_rank(5, dist.get_rank(), dist.get_world_size()) == [0, 1, 2, 3, 4]
        assert parallel.gather_results({"rank": 0}) == [{"rank": 0}]
    finally:
        dist.destroy_process_group()
