#!/usr/bin/env python3
"""Debug helper: dump / compare the variance volume of the warp kernel forms (env is read once per
process, so `dump` runs in child processes).  python tools/gpu/dbg_warp.py"""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
CASES = [(3, 16, 24, 40, 1.0), (5, 24, 32, 48, 0.0), (2, 8, 8, 8, 0.0)]


def dump(path):
    import torch
    from scene_3dreconstruction_mvsnet_amd import _lib, synthetic
    out = {}
    for ci, (N, D, h, w, yaw) in enumerate(CASES):
        feats = synthetic.random_features(N, 32, h, w, seed=4)
        proj = synthetic.cameras(N, h, w, yaw_deg=yaw)
        dv = synthetic.depth_values(D)
        dev = "cuda:0"
        ws = _lib.alloc_workspace(N, 32, D, h, w, dev)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
        var = _lib.warp_variance(t(feats), _lib.relative_proj(t(proj)), t(dv), ws)
        torch.cuda.synchronize()
        out[f"c{ci}"] = _lib.from_c8(var).cpu().numpy()
    np.savez(path, **out)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "dump":
        return dump(sys.argv[2])
    from oracle import oracle as orc
    from scene_3dreconstruction_mvsnet_amd import synthetic
    csrc = os.path.join(REPO, "scene_3dreconstruction_mvsnet_amd", "csrc")
    envs = {"form1": {"MVS_WARP_TC": "1"}, "form2_cpt4": {"MVS_WARP_TC": "2"}, "form3": {}}
    for v in (1, 2, 4, 8):
        lib = os.path.join(csrc, f"libmvs_hip_dbg{v}.so")
        if os.path.exists(lib):
            envs[f"dbg{v}"] = {"MVS_LIB_PATH": lib}
    res = {}
    for name, env in envs.items():
        path = f"/tmp/dbg_{name}.npz"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "dump", path], env=dict(os.environ, **env))
        res[name] = np.load(path)
    for ci, (N, D, h, w, yaw) in enumerate(CASES):
        feats = synthetic.random_features(N, 32, h, w, seed=4)
        want = orc.variance_volume(feats, synthetic.cameras(N, h, w, yaw_deg=yaw), synthetic.depth_values(D))
        for name in envs:
            got = res[name][f"c{ci}"]
            err = np.abs(got - want)
            bad = np.argwhere(err > 1e-3)
            print(f"case {CASES[ci]} {name}: max err {err.max():.3e}, {len(bad)} bad of {err.size}")
            if len(bad):
                c, d, y, x = bad.T
                print("   bad channels", np.unique(c)[:40], "depths", np.unique(d), "ys", np.unique(y)[:30], "xs", np.unique(x)[:50])
                print("   first bad", bad[:6].tolist(), "got", got[tuple(bad[0])], "want", want[tuple(bad[0])])
                print("   equal to form1 where bad:", float(np.abs(got - res['form1'][f'c{ci}']).max()))


if __name__ == "__main__":
    main()
