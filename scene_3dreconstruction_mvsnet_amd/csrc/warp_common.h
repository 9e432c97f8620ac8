// warp_common.h -- sampling helpers shared by the warp+variance kernels (device code).
#pragma once
#include <hip/hip_runtime.h>

namespace mvs {

struct Samp {
    int o00, o01, o10, o11;        // texel offsets (clamped into the window)
    float w00, w01, w10, w11;      // bilinear weights; 0 for taps outside the image; NaN if the
                                   // sampling coordinate is not finite (torch CPU grid_sample)
};

// Window = [x_lo, x_lo+bw) x [y_lo, y_lo+bh) inside the image; offsets are relative to it.
__device__ __forceinline__ Samp make_samp(float qx, float qy, float qz, float tx, float ty, float tz,
                                          float d, float sx, float sy, int h, int w, int x_lo,
                                          int y_lo, int bw, int bh) {
    const float X = fmaf(qx, d, tx), Y = fmaf(qy, d, ty), Z = fmaf(qz, d, tz);
    // one v_rcp_f32 (1 ulp) instead of two IEEE divisions: the sampling coordinate moves by
    // <= 2e-5 px, far inside the parity tolerance, and the expansion of X/Z, Y/Z was ~30 VALU
    // instructions of the ~160 per (pixel, depth, view)
    const float rz = __builtin_amdgcn_rcpf(Z);
    const float ix = (X * rz) * sx - 0.5f;   // px*W/(W-1) - 0.5      (module.py:129-136)
    const float iy = (Y * rz) * sy - 0.5f;
    const bool bad = !(fabsf(ix) <= 3.0e38f) || !(fabsf(iy) <= 3.0e38f);
    const float cx = fminf(fmaxf(ix, -2.0f), (float)w + 1.0f);
    const float cy = fminf(fmaxf(iy, -2.0f), (float)h + 1.0f);
    const float fx0 = floorf(cx), fy0 = floorf(cy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float ax = cx - fx0, ay = cy - fy0;
    const bool in = (cx == ix) && (cy == iy);
    const bool x0ok = in && x0 >= 0 && x0 < w, x1ok = in && x0 + 1 >= 0 && x0 + 1 < w;
    const bool y0ok = y0 >= 0 && y0 < h, y1ok = y0 + 1 >= 0 && y0 + 1 < h;
    Samp s;
    s.w00 = (x0ok && y0ok) ? (1.0f - ax) * (1.0f - ay) : 0.0f;
    s.w01 = (x1ok && y0ok) ? ax * (1.0f - ay) : 0.0f;
    s.w10 = (x0ok && y1ok) ? (1.0f - ax) * ay : 0.0f;
    s.w11 = (x1ok && y1ok) ? ax * ay : 0.0f;
    if (bad) { s.w00 = NAN; s.w01 = NAN; s.w10 = NAN; s.w11 = NAN; }
    const int xa = min(max(x0 - x_lo, 0), bw - 1), xb = min(max(x0 + 1 - x_lo, 0), bw - 1);
    const int ya = min(max(y0 - y_lo, 0), bh - 1), yb = min(max(y0 + 1 - y_lo, 0), bh - 1);
    s.o00 = ya * bw + xa; s.o01 = ya * bw + xb;
    s.o10 = yb * bw + xa; s.o11 = yb * bw + xb;
    return s;
}

// 8 channels of one plane: base points at [texel][8] floats
__device__ __forceinline__ void sample8(const float* __restrict__ base, const Samp& s, float4& lo,
                                        float4& hi) {
    const float4 a0 = *reinterpret_cast<const float4*>(base + s.o00 * 8);
    const float4 a1 = *reinterpret_cast<const float4*>(base + s.o00 * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(base + s.o01 * 8);
    const float4 b1 = *reinterpret_cast<const float4*>(base + s.o01 * 8 + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(base + s.o10 * 8);
    const float4 c1 = *reinterpret_cast<const float4*>(base + s.o10 * 8 + 4);
    const float4 e0 = *reinterpret_cast<const float4*>(base + s.o11 * 8);
    const float4 e1 = *reinterpret_cast<const float4*>(base + s.o11 * 8 + 4);
    lo.x = fmaf(a0.x, s.w00, fmaf(b0.x, s.w01, fmaf(c0.x, s.w10, e0.x * s.w11)));
    lo.y = fmaf(a0.y, s.w00, fmaf(b0.y, s.w01, fmaf(c0.y, s.w10, e0.y * s.w11)));
    lo.z = fmaf(a0.z, s.w00, fmaf(b0.z, s.w01, fmaf(c0.z, s.w10, e0.z * s.w11)));
    lo.w = fmaf(a0.w, s.w00, fmaf(b0.w, s.w01, fmaf(c0.w, s.w10, e0.w * s.w11)));
    hi.x = fmaf(a1.x, s.w00, fmaf(b1.x, s.w01, fmaf(c1.x, s.w10, e1.x * s.w11)));
    hi.y = fmaf(a1.y, s.w00, fmaf(b1.y, s.w01, fmaf(c1.y, s.w10, e1.y * s.w11)));
    hi.z = fmaf(a1.z, s.w00, fmaf(b1.z, s.w01, fmaf(c1.z, s.w10, e1.z * s.w11)));
    hi.w = fmaf(a1.w, s.w00, fmaf(b1.w, s.w01, fmaf(c1.w, s.w10, e1.w * s.w11)));
}

__device__ __forceinline__ void accum(float4& S, float4& Q, const float4& v) {
    S.x += v.x; S.y += v.y; S.z += v.z; S.w += v.w;
    Q.x = fmaf(v.x, v.x, Q.x); Q.y = fmaf(v.y, v.y, Q.y);
    Q.z = fmaf(v.z, v.z, Q.z); Q.w = fmaf(v.w, v.w, Q.w);
}

__device__ __forceinline__ float4 variance4(const float4& S, const float4& Q, float inv_n) {
    // var = Q/N - (S/N)^2        (models/mvsnet.py:177)
    float4 o;
    float m;
    m = S.x * inv_n; o.x = fmaf(-m, m, Q.x * inv_n);
    m = S.y * inv_n; o.y = fmaf(-m, m, Q.y * inv_n);
    m = S.z * inv_n; o.z = fmaf(-m, m, Q.z * inv_n);
    m = S.w * inv_n; o.w = fmaf(-m, m, Q.w * inv_n);
    return o;
}

}  // namespace mvs
