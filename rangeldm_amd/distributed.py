"""Sample-sharded multi-GPU inference.  One process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).  The reference shards by file-index arithmetic with no collective
(ldm/inference.py:159,174-183: image index = (rank + nproc*i)*B + j); here the same index arithmetic picks each
rank's global sample indices, x_T is a function of the GLOBAL index (so 1/2/4/8-GPU runs produce identical images),
and the finished range images are all-gathered once per batch."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local %= torch.cuda.device_count()              # (more ranks than visible devices: the one-GPU rehearsal of an N-rank run)
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get("RLDM_DIST_BACKEND", backend)      # e.g. gloo: RCCL refuses two ranks on one device
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def global_sample_indices(iteration, batch_size, rank, world):
    """Indices of the images rank `rank` produces in outer iteration `iteration` -- ldm/inference.py:174-183."""
    base = (rank + world * iteration) * batch_size
    return list(range(base, base + batch_size))


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of `total` samples for strong-scaling a fixed global batch."""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def all_gather_images(local_images, world=None):
    """All-gather finished (B_local, C, W, H) tensors along dim 0; every rank gets the full batch in rank order.
    One collective per batch: 8 x 1 MiB for BASELINE config 2/3 -- latency-bound, negligible vs the sampling time."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_images
    world = dist.get_world_size()
    local_images = local_images.contiguous()
    out = torch.empty((world * local_images.shape[0], *local_images.shape[1:]), dtype=local_images.dtype,
                      device=local_images.device)
    dist.all_gather_into_tensor(out, local_images)
    return out


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
