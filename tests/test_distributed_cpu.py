"""N>1 path on CPU: 2 gloo ranks shard the units (metas[r::R]) and all-gather the results.

The GPU box runs the same code with backend nccl (= RCCL over xGMI); bench.py --gpus N uses
the same sharding/gather helpers."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from scene_3dreconstruction_mvsnet_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_units, h, w, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = sharding.shard_units(n_units, rank, world)
        kp = sharding.padded_len(n_units, world)
        local = torch.full((kp, 2, h, w), -1.0)
        for i, u in enumerate(mine):  # stand-in for one depth-map inference of unit u
            local[i, 0] = float(u) + torch.arange(h * w, dtype=torch.float32).reshape(h, w) * 1e-3
            local[i, 1] = 1.0 / (1.0 + u)
        out = sharding.gather_maps(local, n_units, rank, world)
        ok = out.shape == (n_units, 2, h, w)
        for u in range(n_units):
            ok &= bool(torch.allclose(out[u, 0, 0, 0], torch.tensor(float(u))))
            ok &= bool(torch.allclose(out[u, 1], torch.full((h, w), 1.0 / (1.0 + u))))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [7, 8, 1])
def test_two_rank_shard_and_gather(n_units):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, 4, 6, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(ret) == {0: True, 1: True}


def test_shard_units_cover_everything_once():
    for n in (0, 1, 5, 1078):  # 1078 = 22 DTU scans x 49 ref views (lists/dtu/test.txt)
        for world in (1, 2, 4, 8):
            seen = sorted(u for r in range(world) for u in sharding.shard_units(n, r, world))
            assert seen == list(range(n))
            assert max(len(sharding.shard_units(n, r, world)) for r in range(world)) == sharding.padded_len(n, world)
    with pytest.raises(ValueError):
        sharding.shard_units(10, 3, 2)
