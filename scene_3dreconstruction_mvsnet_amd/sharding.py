"""Reference-view sharding across ranks and the final result gather.

One unit = one dataset meta `(scan, ref_view, src_views)` = one depth map
(datasets/dataloader_eval.py:41-49,101-110); units are independent during `save_depth`
(eval.py:326-440).  Rank r of R owns `metas[r::R]`; the only collective is one all-gather of the
padded `[ceil(U/R), 2, h, w]` (depth, confidence) buffers at the end -- RCCL over xGMI on the GPU
box (`backend="nccl"`), gloo in the CPU tests.  The reference itself has no multi-process
support (`nn.DataParallel` at batch 1 uses one device, eval.py:309,326).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_units(n_units: int, rank: int, world: int) -> list:
    """Unit indices owned by `rank`: metas[rank::world]."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_units, world))


def padded_len(n_units: int, world: int) -> int:
    return (n_units + world - 1) // world


def gather_maps(local: torch.Tensor, n_units: int, rank: int, world: int, group=None) -> torch.Tensor:
    """All-gather per-rank results into unit order.

    local: [padded_len, 2, h, w] float32 (rows beyond this rank's share are padding).
    Returns [n_units, 2, h, w] on every rank, row u = result of unit u.
    """
    kp = padded_len(n_units, world)
    if local.shape[0] != kp:
        raise ValueError(f"local has {local.shape[0]} rows, expected {kp}")
    if world == 1:
        return local[:n_units]
    # RCCL (backend "nccl") gathers device tensors directly; gloo has no CUDA all-gather, so a
    # gloo rehearsal with device tensors stages through the host.
    stage_host = local.is_cuda and dist.get_backend(group) == "gloo"
    src = local.contiguous().cpu() if stage_host else local.contiguous()
    gathered = torch.empty((world * kp,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(gathered, src, group=group)
    if stage_host:
        gathered = gathered.to(local.device)
    # gathered[r*kp + i] is unit r + i*world
    out = torch.empty((n_units,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_units(n_units, r, world)
        out[idx] = gathered[r * kp: r * kp + len(idx)]
    return out
