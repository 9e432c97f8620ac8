"""Times mvs_filter_depth on a DTU-sized scan with HIP events and, beside it, the
numpy oracle on the host.  Usage: python tools/time_filter.py [--views 49] [--h 128] [--w 160]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from scene_3dreconstruction_mvsnet_amd import _lib, fusion  # noqa: E402
from synthetic_scene import make_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=49)
    ap.add_argument("--h", type=int, default=128)
    ap.add_argument("--w", type=int, default=160)
    ap.add_argument("--nsrc", type=int, default=10)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu-refs", type=int, default=4, help="reference views timed with the oracle")
    a = ap.parse_args()
    depths, confs, Ks, Es, pairs = make_scene(V=a.views, h=a.h, w=a.w, seed=1)
    pairs = [(r, ss[:a.nsrc]) for r, ss in pairs]
    dev = torch.device("cuda", 0)
    ref, src = fusion._pad_pairs(pairs, a.nsrc)
    rm, pm = _lib.filter_compose(Ks, Es, ref, src)
    t = [torch.from_numpy(x).to(dev) for x in (depths, confs, rm, pm, ref, src)]
    for _ in range(3):
        _lib.filter_depth(*t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        _lib.filter_depth(*t)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    R, S = src.shape
    hw = a.h * a.w
    alg = R * hw * ((2 + S) * 4 + 4 + 8 + 3 + 24)
    from oracle import filter_oracle as fo
    t0 = time.perf_counter()
    fo.filter_views(depths, confs, Ks, Es, pairs[:a.cpu_refs], n_view_filter=a.nsrc)
    cpu_s = (time.perf_counter() - t0) / a.cpu_refs
    print(json.dumps({"kernel": "filter_depth_kernel", "views": a.views, "h": a.h, "w": a.w, "nsrc": S,
                      "ms_per_scan": round(ms, 4), "ref_views_per_s": round(R / ms * 1e3, 1),
                      "algorithmic_GB": round(alg / 1e9, 4), "achieved_GBps": round(alg / ms / 1e6, 1),
                      "oracle_ref_views_per_s_numpy": round(1 / cpu_s, 2)}))


if __name__ == "__main__":
    main()
