"""Host side of the C-ABI collectives (include/npp_hip.h: npp_comm_init / npp_allreduce_bucket / npp_syncbn_exchange).

The default transport of GradReducer and of the SyncBatchNorm exchange is torch.distributed (backend "nccl" = RCCL); this module
is the other one: the library's own RCCL communicator, enqueued directly on the stream the caller is on -- no ProcessGroup work
objects, no internal stream, nothing but the collective in a captured graph.  It is what a non-torch host of libnpp_hip.so
would use, and `NPP_COMM=npp` (or `comm.enable()`) switches the training path to it.

torch.distributed is still the side channel for the 128-byte unique id (replaces ProcessGroupNCCL's store exchange behind
augment_lip_sync.py:68 init_process_group).
"""
import os

import torch
import torch.distributed as dist

from . import _lib

_state = {"world": 0, "group": None}


def active():
    return _state["world"] > 0


def enable(group=None):
    """Collective: every rank of `group` (default: WORLD) joins one RCCL communicator owned by libnpp_hip.so on its current
    CUDA device.  Idempotent per process."""
    if active():
        return
    if not torch.cuda.is_available():
        raise RuntimeError("npp_amd.comm: the library's RCCL transport needs a GPU")
    lib = _lib.lib()
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    blob = [None]
    if rank == 0:
        import ctypes
        buf = ctypes.create_string_buffer(128)
        _lib.check(lib.npp_comm_unique_id(buf), "npp_comm_unique_id")
        blob[0] = bytes(buf.raw)
    if world > 1:
        dist.broadcast_object_list(blob, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    torch.cuda.current_stream().synchronize()
    _lib.check(lib.npp_comm_init(blob[0], rank, world), "npp_comm_init")
    _state["world"], _state["group"] = world, group


def disable():
    if active():
        torch.cuda.synchronize()
        _lib.check(_lib.lib().npp_comm_destroy(), "npp_comm_destroy")
        _state["world"], _state["group"] = 0, None


def wanted():
    return os.environ.get("NPP_COMM", "torch") == "npp"


def all_reduce_bucket(flat, average=True):
    """In-place all-reduce of one flat gradient bucket on the current stream."""
    assert flat.is_cuda and flat.is_contiguous() and flat.dtype in (torch.float32, torch.bfloat16)
    dt = _lib.NPP_F32 if flat.dtype == torch.float32 else _lib.NPP_BF16
    _lib.check(_lib.lib().npp_allreduce_bucket(flat.data_ptr(), flat.numel(), dt, 1 if average else 0,
                                               torch.cuda.current_stream().cuda_stream), "npp_allreduce_bucket")


def syncbn_exchange(stats):
    """In-place SUM of f64 BatchNorm partial sums on the current stream."""
    assert stats.is_cuda and stats.is_contiguous() and stats.dtype == torch.float64
    _lib.check(_lib.lib().npp_syncbn_exchange(stats.data_ptr(), stats.numel(), torch.cuda.current_stream().cuda_stream),
               "npp_syncbn_exchange")
