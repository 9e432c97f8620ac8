"""Synthetic multi-view depth maps of a tilted plane for the filter / fusion tests (no reference
data is available offline).  Cameras are DTU-like at feature scale; every view's depth map is the
analytic ray/plane intersection times a smooth view-dependent perturbation of a few 0.1 %, so that
the 1 % relative-depth and 1 px reprojection checks (eval.py:574-582) both pass and fail somewhere."""
import numpy as np


def make_scene(V=6, h=64, w=80, seed=0, noise=0.006, rot=0.04):
    rng = np.random.default_rng(seed)
    K = np.array([[361.5 * w / 160, 0, w / 2], [0, 360.0 * h / 128, h / 2], [0, 0, 1]], np.float64)
    Ks = np.tile(K.astype(np.float32), (V, 1, 1))
    Es = np.zeros((V, 4, 4), np.float32)
    n = np.array([0.2, 0.1, 1.0])
    c = 700.0
    depths = np.zeros((V, h, w), np.float32)
    confs = np.zeros((V, h, w), np.float32)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    for v in range(V):
        a, b = rot * rng.standard_normal(2)
        Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        R = Ry @ Rx
        t = np.array([-25.0 * v + 60, 6.0 * v - 15, 3.0 * rng.standard_normal()])
        E = np.eye(4)
        E[:3, :3], E[:3, 3] = R, t
        Es[v] = E.astype(np.float32)
        E64 = Es[v].astype(np.float64)
        Rf, tf = E64[:3, :3], E64[:3, 3]
        dirs = np.linalg.inv(Ks[v].astype(np.float64)) @ np.stack([xs.ravel(), ys.ravel(), np.ones(h * w)])
        Rt = np.linalg.inv(Rf)
        d = (c + n @ (Rt @ tf)) / (n @ (Rt @ dirs))
        ph = rng.uniform(0, 6.28, 2)
        pert = 1 + noise * np.sin(xs.ravel() * 0.21 + ph[0]) * np.cos(ys.ravel() * 0.17 + ph[1])
        depths[v] = (d * pert).reshape(h, w).astype(np.float32)
        confs[v] = (0.5 + 0.5 * np.sin(xs * 0.13 + v) * np.cos(ys * 0.11 - v)).astype(np.float32)
    pairs = [(v, [s for s in np.roll(np.arange(V), -v)[1:]]) for v in range(V)]
    pairs = [(int(r), [int(s) for s in ss]) for r, ss in pairs]
    return depths, confs, Ks, Es, pairs
