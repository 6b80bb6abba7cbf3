"""Multi-GPU partitioning of the env batch: one process per GPU, contiguous shards, and NO
collective on the step path (every env is independent — SURVEY.md §8e).  The only optional
exchange is a learner-side all-gather of per-shard rows (returns / advantages) over
RCCL (`backend='nccl'` on ROCm) or gloo in CPU tests.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch


TILE = 64  # envs per wavefront: the unit the in-launch reset stream is keyed by (quadrotor_hip.h: reset_count)


def shard_range(num_envs_global: int, rank: int, world_size: int, align: int = TILE) -> Tuple[int, int]:
    """Contiguous [start, stop) of the global env index range owned by `rank`.

    Shards are cut at multiples of `align` (= 64, one wavefront): the in-launch reset draws of a
    64-env tile are keyed by the global id of its first env, so with tile-aligned shards every
    result is independent of the number of GPUs.  The 64-env tiles are dealt as evenly as possible
    (the first `tiles % world_size` ranks get one more); the last shard ends at num_envs_global.
    Batches smaller than `align * world_size` envs fall back to align = 1."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    if num_envs_global < 0:
        raise ValueError("num_envs_global must be >= 0")
    if align < 1 or num_envs_global < align * world_size:
        align = 1
    tiles = -(-num_envs_global // align)
    base, rem = divmod(tiles, world_size)
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    return min(start * align, num_envs_global), min(stop * align, num_envs_global)


def rank_world_from_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as exported by torch.distributed.run."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def make_sharded_env(kind: str, num_envs_global: int, rank: int = None, world_size: int = None,
                     local_rank: int = None, **kwargs):
    """Build this rank's QuadVecEnv shard on cuda:<local_rank>.  `env_offset` is set to the
    shard start so RNG draws depend only on (seed, global env id, episode) — in-launch resets on
    (seed, global 64-env tile, tile counter, slot): results are the same whatever the number of GPUs."""
    from .vec_env import QuadVecEnv
    r, lr, w = rank_world_from_env()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    local_rank = lr if local_rank is None else local_rank
    start, stop = shard_range(num_envs_global, rank, world_size)
    if stop <= start:
        raise ValueError(f"rank {rank} owns no envs ({num_envs_global} envs over {world_size} ranks)")
    return QuadVecEnv(kind=kind, num_envs=stop - start, device=torch.device("cuda", local_rank),
                      env_offset=start, **kwargs)


def all_gather_rows(local: torch.Tensor, num_envs_global: int, group=None) -> torch.Tensor:
    """Gather per-shard tensors whose LAST-BUT-TRAILING env axis is dim `-1` ... simply:
    `local` is [..., n_local]; returns [..., num_envs_global] on every rank, in global env
    order.  Shards may differ in size (see shard_range), so ranks pad to the max shard."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    n_max = max(e - s for s, e in (shard_range(num_envs_global, r, world) for r in range(world)))
    pad = n_max - local.shape[-1]
    buf = torch.nn.functional.pad(local, (0, pad)) if pad else local
    buf = buf.contiguous()
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = []
    for r, p in enumerate(parts):
        s, e = shard_range(num_envs_global, r, world)
        out.append(p[..., : e - s])
    return torch.cat(out, dim=-1)
