"""Seeded on-disk DTU-like mini dataset used by the dataset tests and their golden generator."""
import os

import numpy as np
from PIL import Image


def write_synthetic_dataset(root: str) -> str:
    """Creates root/data/{pair.txt,Cameras/*_cam.txt,Rectified/scan*/rect_*.png} and root/list.txt;
    returns the list file path.  2 scans x 4 views of 150x200 RGB images (one grey-scale)."""
    rng = np.random.default_rng(7)
    data = os.path.join(root, "data")
    os.makedirs(os.path.join(data, "Cameras"), exist_ok=True)
    nviews = 4
    with open(os.path.join(data, "pair.txt"), "w") as f:
        f.write(f"{nviews}\n")
        for v in range(nviews):
            others = [o for o in range(nviews) if o != v]
            f.write(f"{v}\n{len(others)} " + " ".join(f"{o} {100.0 - o:.2f}" for o in others) + " \n")
    for v in range(nviews):
        E = np.eye(4)
        E[:3, 3] = [-30.0 * v, 5.0 * v, 0.5 * v]
        K = np.array([[452.0, 0.0, 100.3], [0.0, 450.5, 74.8], [0.0, 0.0, 1.0]])
        with open(os.path.join(data, "Cameras", f"{v:08d}_cam.txt"), "w") as f:
            f.write("extrinsic\n")
            for row in E:
                f.write(" ".join(f"{x:.6f}" for x in row) + " \n")
            f.write("\nintrinsic\n")
            for row in K:
                f.write(" ".join(f"{x:.6f}" for x in row) + " \n")
            f.write(f"\n{425.0 + v} 2.5 \n")
    scans = ["scan1", "scan9"]
    for scan in scans:
        d = os.path.join(data, "Rectified", scan)
        os.makedirs(d, exist_ok=True)
        for v in range(nviews):
            if scan == "scan9" and v == 2:
                arr = rng.integers(0, 256, size=(150, 200), dtype=np.uint8)  # grey-scale image
            else:
                arr = rng.integers(0, 256, size=(150, 200, 3), dtype=np.uint8)
            Image.fromarray(arr).save(os.path.join(d, f"rect_{v + 1:03d}_3_r5000.png"))
    listfile = os.path.join(root, "list.txt")
    with open(listfile, "w") as f:
        f.write("\n".join(scans) + "\n")
    return listfile
