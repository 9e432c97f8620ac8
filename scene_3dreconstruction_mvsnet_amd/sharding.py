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


# ---------------------------------------------------------------------------- host-core affinity
# One process per GPU: each rank's host threads (launch thread, decoder pool, PFM writers) should sit on
# the cores of the NUMA node its GPU hangs off -- the counterpart of nothing in the reference
# (nn.DataParallel, eval.py:309, is one process), needed because 8 ranks otherwise wander over both sockets.

def _parse_cpulist(text: str) -> list:
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(sysfs_root: str = "/sys", with_ids: bool = False):
    """`local_cpulist` of every AMD display/accelerator PCI function, in PCI bus order (the order HIP
    usually enumerates devices in; `check_gpu_order` verifies it once HIP is up).  Reads sysfs only -- no HIP
    call, safe before a fork/exec.  with_ids: also return the PCI addresses ("0000:05:00.0") in the same order."""
    import glob
    import os
    out, ids = [], []
    for dev in sorted(glob.glob(os.path.join(sysfs_root, "bus/pci/devices/*"))):
        try:
            with open(os.path.join(dev, "vendor")) as f:
                vendor = f.read().strip()
            with open(os.path.join(dev, "class")) as f:
                cls = int(f.read().strip(), 16) >> 16
            if vendor != "0x1002" or cls not in (0x03, 0x12):   # display controller / processing accelerator
                continue
            with open(os.path.join(dev, "local_cpulist")) as f:
                out.append(_parse_cpulist(f.read()))
            ids.append(os.path.basename(dev))
        except (OSError, ValueError):
            continue
    return (out, ids) if with_ids else out


def rank_cpus(local_rank: int, local_world: int, allowed: list, gpu_cpus: list | None = None) -> list:
    """Host cores for `local_rank` of `local_world` ranks on this node: the allowed cores local to its GPU,
    split evenly among the ranks whose GPUs share those cores; a contiguous 1/local_world slice of the
    allowed cores when the topology is unknown (or does not list that many GPUs)."""
    allowed = sorted(allowed)
    if not (0 <= local_rank < local_world):
        raise ValueError(f"local rank {local_rank} outside {local_world}")
    if gpu_cpus and len(gpu_cpus) >= local_world:
        mine = sorted(set(gpu_cpus[local_rank]) & set(allowed))
        peers = [r for r in range(local_world) if sorted(set(gpu_cpus[r]) & set(allowed)) == mine]
        if mine and len(mine) >= len(peers):
            i, n = peers.index(local_rank), len(peers)
            return mine[i * len(mine) // n:(i + 1) * len(mine) // n]
    n = len(allowed)
    if n < local_world:
        return allowed
    return allowed[local_rank * n // local_world:(local_rank + 1) * n // local_world]


def visible_gpu_order(n_physical: int, environ=None) -> list:
    """Physical indices of the GPUs this process sees, in the order HIP numbers them.  The launcher may have
    narrowed or permuted them: ROCR_VISIBLE_DEVICES acts on the runtime below HIP; on top of that the HIP runtime
    honours HIP_VISIBLE_DEVICES, and CUDA_VISIBLE_DEVICES only as its ALIAS when HIP_VISIBLE_DEVICES is unset (both
    set to the same permutation is one re-indexing, not two).  Integer lists only; anything else -- UUIDs, an
    empty value, an index out of range -- leaves the order as it was."""
    import os
    env = os.environ if environ is None else environ

    def apply(order, val):
        try:
            idx = [int(t) for t in val.split(",") if t.strip() != ""]
        except ValueError:
            return order
        if idx and all(0 <= i < len(order) for i in idx):
            return [order[i] for i in idx]
        return order

    order = list(range(n_physical))
    if env.get("ROCR_VISIBLE_DEVICES"):
        order = apply(order, env["ROCR_VISIBLE_DEVICES"])
    hip = env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES")
    if hip:
        order = apply(order, hip)
    return order


def check_gpu_order(gpu_cpus_sysfs_order: list, sysfs_bus_ids: list, hip_bus_ids: list) -> list:
    """After the first GPU call: HIP's own PCI bus ids against the sorted-sysfs order `gpu_local_cpus` assumed
    (ROCr does not promise to enumerate in PCI address order).  Returns the per-HIP-device cpu lists re-ordered by
    bus id, or the input when an id is unknown; callers log a mismatch (affinity only -- results are unaffected)."""
    pos = {b.lower(): i for i, b in enumerate(sysfs_bus_ids)}
    try:
        return [gpu_cpus_sysfs_order[pos[b.lower()]] for b in hip_bus_ids]
    except KeyError:
        return gpu_cpus_sysfs_order


def pin_rank(local_rank: int, local_world: int, sysfs_root: str = "/sys") -> list:
    """Pin the calling process to its rank's cores (call BEFORE the first GPU call so that the runtime's
    helper threads inherit the mask).  Returns the cores; [] when the platform has no affinity API."""
    import os
    if not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return []
    gpus = gpu_local_cpus(sysfs_root)
    gpus = [gpus[i] for i in visible_gpu_order(len(gpus))]
    cpus = rank_cpus(local_rank, local_world, sorted(os.sched_getaffinity(0)), gpus)
    if cpus:
        os.sched_setaffinity(0, cpus)
    return cpus


def verify_pinning(local_rank: int, hip_bus_id: str, sysfs_root: str = "/sys") -> dict:
    """After the first GPU call: is the GPU HIP gave this rank the one whose cores `pin_rank` chose?  Compares
    the device's PCI address as HIP reports it with the sorted-sysfs entry `pin_rank` assumed for `local_rank`.
    Affinity only -- a mismatch is reported (bench.py prints it), never fatal."""
    _, ids = gpu_local_cpus(sysfs_root, with_ids=True)
    order = visible_gpu_order(len(ids))
    assumed = ids[order[local_rank]] if local_rank < len(order) and order[local_rank] < len(ids) else None
    return {"hip_bus_id": hip_bus_id, "assumed_bus_id": assumed,
            "match": (assumed is not None and assumed.lower() == hip_bus_id.lower())}
