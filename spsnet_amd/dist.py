"""Scene sharding over the GPUs of one node.

Every kernel of the SA path indexes scenes independently (FPS: blockIdx.x = scene,
reference sampling_gpu.cu:105; the others blockIdx.y/z), so a batch shards over ranks along
the batch dimension with replicated weights -- the reference's DDP + DistributedSampler layout
(tools/train.py:147-149, pcdet/datasets/__init__.py:57-66).  There is no collective on the data
path; the only exchange is an all-gather of each layer's `sampled_idx_list` when a consumer wants
the global-batch view (int32 (B/world, M) per rank -> (B, M)); with backend "nccl" that is RCCL
over xGMI, and the per-layer gathers are fused into one call because they are latency bound.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(total_scenes: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of scenes owned by `rank`; the first total % world ranks get one extra."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(total_scenes, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def all_gather_sampled_idx(per_layer_idx: Sequence[torch.Tensor], group=None, always_collective: bool = False) -> List[torch.Tensor]:
    """All-gather every layer's (b_local, M_l) int32 sampled indices into (B, M_l) tensors with ONE
    collective: the layers are packed side by side into a (b_local, sum M_l) buffer first.
    Ranks must hold the same number of scenes (pad the batch otherwise).
    always_collective: issue the collective in a group of one rank too (bench.py --force-exchange: the only way to put the
    RCCL call itself on a one-GPU box); by default a lone rank just copies."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        return [t.clone() for t in per_layer_idx]
    world = dist.get_world_size(group)
    widths = [int(t.shape[1]) for t in per_layer_idx]
    packed = torch.cat([t.to(torch.int32) for t in per_layer_idx], dim=1).contiguous()
    gathered = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=packed.dtype, device=packed.device)
    if hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(gathered, packed, group=group)
    else:  # pragma: no cover
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed, group=group)
        gathered = torch.cat(parts, dim=0)
    return [g.contiguous() for g in torch.split(gathered, widths, dim=1)]
