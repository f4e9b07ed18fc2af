"""Multi-GPU: independent video clips sharded over ranks, one process per GPU.

A clip's only state is its own K/Q/V FIFO (td4_psp18.py:118-134), so ranks never exchange per-frame data.  The only
collectives are: ONE broadcast of the flat fp32 weight blob from rank 0 (RCCL over xGMI; 219 MB for td4) at start, and
small all-reduces of the 19x19 confusion matrix / timing at the end.  Backend "nccl" is RCCL on ROCm; the CPU tests run
the same code over gloo with world_size 2.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import arch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def clips_of_rank(n_clips, rank, world):
    """clip c is served by rank c mod world; frames of a clip stay ordered on one GPU."""
    return list(range(rank, n_clips, world))


def broadcast_state_dict(spec, h, w, sd, device, src=0):
    """Rank `src` holds `sd` ({name: ndarray}); every rank returns the identical dict.  One flat fp32 broadcast."""
    shapes = arch.state_dict_shapes(spec, h, w)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return sd
    sizes = {k: int(np.prod(s)) if len(s) else 1 for k, s in shapes.items()}
    total = sum(sizes.values())
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        off = 0
        host = np.empty(total, np.float32)
        for k in shapes:
            host[off:off + sizes[k]] = np.asarray(sd[k], dtype=np.float32).reshape(-1)
            off += sizes[k]
        flat.copy_(torch.from_numpy(host))
    dist.broadcast(flat, src=src)
    host = flat.cpu().numpy()
    out, off = {}, 0
    for k, s in shapes.items():
        out[k] = host[off:off + sizes[k]].reshape(s).copy()
        off += sizes[k]
    return out


def allreduce_sum(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_max(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
