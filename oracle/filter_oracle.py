"""CPU oracle for the depth-map filter / fusion step (reference eval.py:508-585 and
eval.py:590-760, helpers eval.py:253-275).

TEST INFRASTRUCTURE ONLY: imported by tests/ and tools/ timing scripts as the checker, never by
scene_3dreconstruction_mvsnet_amd (the product).

PARITY UNPINNED for the bilinear sampler: the reference samples the source depth map with
`cv2.remap(..., INTER_LINEAR)` (eval.py:541) and OpenCV is not importable in this image (no
`cv2`, nothing may be installed), and the reference holds no fixture for this step.  `remap_linear`
below restates OpenCV's published algorithm for float32 maps (modules/imgproc/src/imgwarp.cpp,
remap -> remapBilinear, OpenCV 4.x): coordinates are quantised to 1/32 pixel with round-half-even,
weights come from the 32x32 bilinear table, taps outside the image contribute the border value 0.
Everything else (numpy dtype promotion) follows eval.py line by line in behaviour and is exercised
against hand-computed cases in tests/test_filter_oracle.py.

Operation order: the reference forms its small matrix products with `np.matmul` and its inverses
with `np.linalg.inv`, whose rounding order belongs to whatever BLAS / LAPACK build numpy links
(fused or unfused multiply-adds, blocking).  This oracle fixes ONE order so that the GPU kernel
can be held to bit-equality with it: dot products are evaluated left to right with separately
rounded multiplies and adds (`_dot3`, `_mm4_f32`), inverses by LU with partial pivoting and plain
forward / back substitution in float32 (`_inv_f32`, the algorithm of LAPACK's sgesv).  Against a
BLAS-linked numpy the values differ in the last bit of some float32 matrix entries.
"""
from __future__ import annotations

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def _cv_round(v: np.ndarray) -> np.ndarray:
    """cvRound on float32: round-half-even; NaN / out-of-int32-range -> INT_MIN (cvtss2si)."""
    r = np.rint(v.astype(np.float32))
    ok = np.isfinite(r) & (r >= -2147483648.0) & (r < 2147483648.0)
    out = np.full(v.shape, -2147483648, np.int64)
    out[ok] = r[ok].astype(np.int64)
    return out


def remap_linear(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv2.remap(src, map_x, map_y, INTER_LINEAR) for float32 single-channel `src`, default
    BORDER_CONSTANT / borderValue 0 (the call at eval.py:541)."""
    assert src.dtype == np.float32 and map_x.dtype == np.float32 and map_y.dtype == np.float32
    H, W = src.shape
    sx = _cv_round(map_x * np.float32(INTER_TAB_SIZE))
    sy = _cv_round(map_y * np.float32(INTER_TAB_SIZE))
    fx = (sx & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE)
    fy = (sy & (INTER_TAB_SIZE - 1)).astype(np.float32) / np.float32(INTER_TAB_SIZE)
    ix = np.clip(sx >> INTER_BITS, -32768, 32767)
    iy = np.clip(sy >> INTER_BITS, -32768, 32767)
    one = np.float32(1)
    w = [(one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx]  # exact products

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(ok, v, np.float32(0))

    with np.errstate(invalid="ignore", over="ignore"):
        out = tap(iy, ix) * w[0]
        out = out + tap(iy, ix + 1) * w[1]
        out = out + tap(iy + 1, ix) * w[2]
        out = out + tap(iy + 1, ix + 1) * w[3]
    gone = (ix >= W) | (ix + 1 < 0) | (iy >= H) | (iy + 1 < 0)
    return np.where(gone, np.float32(0), out).astype(np.float32)


def _inv_f32(a: np.ndarray) -> np.ndarray:
    """float32 inverse: LU with partial pivoting, then forward / back substitution per unit column;
    every multiply and add rounded separately in float32 (no fused operations)."""
    n = a.shape[0]
    f = np.float32
    lu = [[f(a[i, j]) for j in range(n)] for i in range(n)]
    piv = []
    for k in range(n):
        p = k
        for i in range(k + 1, n):
            if abs(lu[i][k]) > abs(lu[p][k]):
                p = i
        piv.append(p)
        if lu[p][k] == 0:
            raise np.linalg.LinAlgError("singular matrix")
        if p != k:
            lu[k], lu[p] = lu[p], lu[k]
        rp = f(1) / lu[k][k]
        for i in range(k + 1, n):
            lu[i][k] = f(lu[i][k] * rp)
        for i in range(k + 1, n):
            for j in range(k + 1, n):
                lu[i][j] = f(lu[i][j] - f(lu[i][k] * lu[k][j]))
    out = np.empty((n, n), np.float32)
    for c in range(n):
        b = [f(1) if i == c else f(0) for i in range(n)]
        for k in range(n):
            b[k], b[piv[k]] = b[piv[k]], b[k]
        for i in range(n):
            for j in range(i):
                b[i] = f(b[i] - f(lu[i][j] * b[j]))
        for i in range(n - 1, -1, -1):
            for j in range(i + 1, n):
                b[i] = f(b[i] - f(lu[i][j] * b[j]))
            b[i] = f(b[i] / lu[i][i])
        for i in range(n):
            out[i, c] = b[i]
    return out


def _mm4_f32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """4x4 float32 product, s = ((a0*b0 + a1*b1) + a2*b2) + a3*b3 starting from 0, unfused."""
    f = np.float32
    out = np.empty((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            s_ = f(0)
            for k in range(4):
                s_ = f(s_ + f(f(a[i, k]) * f(b[k, j])))
            out[i, j] = s_
    return out


def _dot3(m, a, b, c):
    """(m0*a + m1*b) + m2*c in float64, every operation rounded separately (numpy never fuses)."""
    m = np.asarray(m, np.float64)
    return (m[0] * a + m[1] * b) + m[2] * c


def _dot4(m, a, b, c):
    return _dot3(m, a, b, c) + np.float64(m[3])


def _pix_rows(h, w):
    ys, xs = np.mgrid[0:h, 0:w]
    return xs.reshape(-1), ys.reshape(-1)  # int64, row-major (np.meshgrid + reshape, eval.py:518-519)


def reproject(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """eval.py:508-561.  Returns depth_reprojected, x_reprojected, y_reprojected, x_src, y_src."""
    h, w = depth_ref.shape
    xr, yr = _pix_rows(h, w)
    with np.errstate(all="ignore"):
        pix = np.stack([xr, yr, np.ones_like(xr)]) * depth_ref.reshape(-1)      # int64*f32 -> f64
        Kri = _inv_f32(K_ref)                                                    # f32 inverse, eval.py:522
        p_ref = [_dot3(Kri[i], pix[0], pix[1], pix[2]) for i in range(3)]
        T = _mm4_f32(E_src, _inv_f32(E_ref))                                     # f32 4x4, eval.py:525
        p_src = [_dot4(T[i], p_ref[0], p_ref[1], p_ref[2]) for i in range(3)]    # homogeneous 1 -> + T[i,3]
        q = [_dot3(K_src[i], p_src[0], p_src[1], p_src[2]) for i in range(3)]
        xy = np.stack([q[0] / q[2], q[1] / q[2]])                                # eval.py:529
        x_src = xy[0].reshape(h, w).astype(np.float32)
        y_src = xy[1].reshape(h, w).astype(np.float32)
        samp = remap_linear(depth_src, x_src, y_src)                             # eval.py:541
        sm = samp.reshape(-1).astype(np.float64)
        Ksi = _inv_f32(K_src)
        bx, by, bz = xy[0] * sm, xy[1] * sm, sm                                  # eval.py:546
        back = [_dot3(Ksi[i], bx, by, bz) for i in range(3)]
        T2 = _mm4_f32(E_ref, _inv_f32(E_src))
        p_rep = [_dot4(T2[i], back[0], back[1], back[2]) for i in range(3)]      # eval.py:549
        d_rep = p_rep[2].reshape(h, w).astype(np.float32)
        q2 = [_dot3(K_ref[i], p_rep[0], p_rep[1], p_rep[2]) for i in range(3)]
        x_rep = (q2[0] / q2[2]).reshape(h, w).astype(np.float32)
        y_rep = (q2[1] / q2[2]).reshape(h, w).astype(np.float32)
    return d_rep, x_rep, y_rep, x_src, y_src


def geometric_consistency(depth_ref, K_ref, E_ref, depth_src, K_src, E_src,
                          condmask_pixel=1.0, condmask_depth=0.01):
    """eval.py:566-585: mask, masked reprojected depth, and the margins to both thresholds
    (margins are oracle extras used by the tests to recognise borderline pixels)."""
    h, w = depth_ref.shape
    ys, xs = np.mgrid[0:h, 0:w]
    d_rep, x_rep, y_rep, x_src, y_src = reproject(depth_ref, K_ref, E_ref, depth_src, K_src, E_src)
    with np.errstate(all="ignore"):
        dx, dy = x_rep - xs, y_rep - ys                                          # f32 - int64 -> f64
        dist = np.sqrt(dx * dx + dy * dy)
        rel = np.abs(d_rep - depth_ref) / depth_ref                              # f32
        mask = np.logical_and(dist < condmask_pixel, rel < np.float32(condmask_depth))
    d_rep = d_rep.copy()
    d_rep[~mask] = 0
    return mask, d_rep, dist, rel


def depth2pts(depth_map, K, E):
    """eval.py:253-275 (pixel centres at +0.5; x,y of the world point scaled by 1.0531)."""
    h, w = depth_map.shape
    xs = np.linspace(0.5, w - 0.5, w)
    ys = np.linspace(0.5, h - 0.5, h)
    gx, gy = np.meshgrid(xs, ys)
    gx, gy, one = gx.reshape(-1), gy.reshape(-1), np.ones(h * w)
    Ki = _inv_f32(np.asarray(K, np.float32))
    Ri = _inv_f32(np.ascontiguousarray(np.asarray(E, np.float32)[:3, :3]))
    d = np.asarray(depth_map, np.float64).reshape(-1)
    cam = [_dot3(Ki[i], gx, gy, one) * d - np.float64(E[i, 3]) for i in range(3)]
    world = np.stack([_dot3(Ri[i], cam[0], cam[1], cam[2]) for i in range(3)], axis=1)
    world[:, :2] = world[:, :2] * 1.0531
    return world


def filter_views(depths, confs, Ks, Es, pairs, n_view_filter=10, photomask=0.8, geomask=3,
                 condmask_pixel=1.0, condmask_depth=0.01):
    """Per-reference-view part of filter_depth (eval.py:620-705, 744-752), arrays in / arrays out.

    depths, confs: [V,h,w] float32; Ks [V,3,3], Es [V,4,4] float32; pairs: [(ref, [src...])].
    Returns one dict per pair with geo_sum (int32), depth_avg (float64), photo/geo/final masks,
    xyz_world [h*w,3] float64, and the per-source margins."""
    out = []
    for ref, srcs in pairs:
        d_ref = depths[ref]
        geo_sum = np.zeros(d_ref.shape, np.int32)
        acc = 0
        margins = []
        for s in list(srcs)[:n_view_filter]:
            m, d_rep, dist, rel = geometric_consistency(d_ref, Ks[ref], Es[ref], depths[s], Ks[s], Es[s],
                                                        condmask_pixel, condmask_depth)
            geo_sum = geo_sum + m.astype(np.int32)
            acc = acc + d_rep                                                    # sum(list), eval.py:699
            margins.append((dist, rel))
        with np.errstate(all="ignore"):
            depth_avg = (acc + d_ref) / (geo_sum + 1)                            # f32 / int32 -> f64
        photo = confs[ref] > np.float32(photomask)
        geo = geo_sum >= geomask
        final = np.logical_and(photo, geo)
        with np.errstate(all="ignore"):
            xyz = depth2pts(depth_avg, Ks[ref], Es[ref])
        out.append(dict(geo_sum=geo_sum, depth_avg=depth_avg, photo=photo, geo=geo, final=final,
                        xyz_world=xyz, margins=margins))
    return out
