// filter_depth.hip -- geometric-consistency filter + fusion of the estimated depth maps
// (reference eval.py:508-585 reproject/check, 620-705 per-view loop, 253-275
// depth2pts_np).  One thread per (reference view, pixel); the loop over source views runs
// inside the thread so geo_sum / the depth sum never leave registers.  HBM-bound and tiny:
// per reference view it reads (1+S) depth maps + 1 confidence map and writes 40 B per pixel.
//
// Arithmetic types follow numpy's promotion in the reference exactly: float32 camera matrices
// (inverses and 4x4 products formed in float32 by mvs_filter_compose), float64 for everything
// that touches the int64 pixel grid, float32 casts where the reference casts (eval.py:539-540,
// 552-556).  The bilinear sampler restates cv2.remap(INTER_LINEAR) on float32 maps: positions
// quantised to 1/32 px (round-half-even), taps outside the image are 0.
#include <climits>
#include <cmath>

#include "mvs_internal.h"

// No fused multiply-adds anywhere in this file (the Makefile also builds it with -ffp-contract=off):
// HIP's __dmul_rn / __dadd_rn / __fmul_rn are plain `*` and `+` that hipcc contracts into fma under its
// default -ffp-contract=fast, while the oracle (numpy elementwise arithmetic) rounds every product and
// sum separately; the outputs are held to bit-equality.
#pragma clang fp contract(off)

namespace mvs {

__device__ __forceinline__ int cv_round(float v) {
    const float r = rintf(v);
    return (r >= -2147483648.f && r < 2147483648.f) ? (int)r : INT_MIN;  // NaN -> INT_MIN
}

__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

__device__ float remap_linear(const float* __restrict__ src, int h, int w, float mx, float my) {
    const int sx = cv_round(mx * 32.f), sy = cv_round(my * 32.f);
    const float fx = (float)(sx & 31) * (1.f / 32.f), fy = (float)(sy & 31) * (1.f / 32.f);
    const int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);
    if (ix >= w || ix + 1 < 0 || iy >= h || iy + 1 < 0) return 0.f;
    const bool x0 = ix >= 0, x1 = ix + 1 < w, y0 = iy >= 0, y1 = iy + 1 < h;
    const float v00 = (y0 && x0) ? src[(size_t)iy * w + ix] : 0.f;
    const float v01 = (y0 && x1) ? src[(size_t)iy * w + ix + 1] : 0.f;
    const float v10 = (y1 && x0) ? src[(size_t)(iy + 1) * w + ix] : 0.f;
    const float v11 = (y1 && x1) ? src[(size_t)(iy + 1) * w + ix + 1] : 0.f;
    // separate mul/add (no contraction): OpenCV's scalar float path is not FMA-fused
    float r = __fmul_rn(v00, __fmul_rn(1.f - fy, 1.f - fx));
    r = __fadd_rn(r, __fmul_rn(v01, __fmul_rn(1.f - fy, fx)));
    r = __fadd_rn(r, __fmul_rn(v10, __fmul_rn(fy, 1.f - fx)));
    r = __fadd_rn(r, __fmul_rn(v11, __fmul_rn(fy, fx)));
    return r;
}

__device__ __forceinline__ double dot3(const float* m, double a, double b, double c) {
    return __dadd_rn(__dadd_rn(__dmul_rn((double)m[0], a), __dmul_rn((double)m[1], b)),
                     __dmul_rn((double)m[2], c));
}
__device__ __forceinline__ double dot4(const float* m, double a, double b, double c) {
    return __dadd_rn(dot3(m, a, b, c), (double)m[3]);
}

struct FilterParams {
    const float* depth;
    const float* conf;
    const float* ref_mats;
    const float* pair_mats;
    const int* ref_idx;
    const int* src_idx;
    int V, R, S, h, w;
    float photomask, condmask_depth;
    double condmask_pixel;
    int geomask;
    int* geo_sum;
    double* depth_avg;
    unsigned char* masks;
    double* xyz_world;
};

__global__ __launch_bounds__(256) void filter_depth_kernel(FilterParams P) {
    const int r = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = P.h * P.w;
    if (p >= hw) return;
    const int y = p / P.w, x = p - y * P.w;
    const int ref = P.ref_idx[r];
    if (ref < 0 || ref >= P.V) return;
    const float* RM = P.ref_mats + (size_t)r * MVS_FILTER_REF_FLOATS;
    const float* Kri = RM;        // inverse(K_ref)                 eval.py:522
    const float* Kr = RM + 9;     // K_ref                          eval.py:554
    const float d_ref = P.depth[(size_t)ref * hw + p];
    // reference 3-D point: inv(K_ref) @ ([x,y,1] * d)              eval.py:522
    const double X = (double)x * (double)d_ref, Y = (double)y * (double)d_ref, Z = (double)d_ref;
    const double pr0 = dot3(Kri, X, Y, Z), pr1 = dot3(Kri + 3, X, Y, Z), pr2 = dot3(Kri + 6, X, Y, Z);

    int geo = 0;
    float acc = 0.f;
    for (int s = 0; s < P.S; ++s) {
        const int src = P.src_idx[r * P.S + s];
        if (src < 0 || src >= P.V) continue;
        const float* PM = P.pair_mats + ((size_t)r * P.S + s) * MVS_FILTER_PAIR_FLOATS;
        const float* T = PM;          // (E_src @ inv(E_ref))[:3]   eval.py:525
        const float* Ks = PM + 12;    // K_src
        const float* Ksi = PM + 21;   // inverse(K_src)             eval.py:546
        const float* T2 = PM + 30;    // (E_ref @ inv(E_src))[:3]   eval.py:549
        const double ps0 = dot4(T, pr0, pr1, pr2), ps1 = dot4(T + 4, pr0, pr1, pr2),
                     ps2 = dot4(T + 8, pr0, pr1, pr2);
        const double q0 = dot3(Ks, ps0, ps1, ps2), q1 = dot3(Ks + 3, ps0, ps1, ps2),
                     q2 = dot3(Ks + 6, ps0, ps1, ps2);
        const double xs = q0 / q2, ys = q1 / q2;                                    // eval.py:529
        const float smp = remap_linear(P.depth + (size_t)src * hw, P.h, P.w, (float)xs, (float)ys);
        const double bx = xs * (double)smp, by = ys * (double)smp, bz = (double)smp;  // eval.py:546
        const double b0 = dot3(Ksi, bx, by, bz), b1 = dot3(Ksi + 3, bx, by, bz),
                     b2 = dot3(Ksi + 6, bx, by, bz);
        const double w0 = dot4(T2, b0, b1, b2), w1 = dot4(T2 + 4, b0, b1, b2),
                     w2 = dot4(T2 + 8, b0, b1, b2);
        const float d_rep = (float)w2;                                              // eval.py:552
        const double k0 = dot3(Kr, w0, w1, w2), k1 = dot3(Kr + 3, w0, w1, w2),
                     k2 = dot3(Kr + 6, w0, w1, w2);
        const float xr = (float)(k0 / k2), yr = (float)(k1 / k2);                   // eval.py:554-556
        const double dx = (double)xr - (double)x, dy = (double)yr - (double)y;
        const double dist = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));  // eval.py:574
        const float rel = fabsf(d_rep - d_ref) / d_ref;                             // eval.py:577-578
        const bool ok = (dist < P.condmask_pixel) && (rel < P.condmask_depth);      // eval.py:582
        geo += ok ? 1 : 0;
        acc = __fadd_rn(acc, ok ? d_rep : 0.f);                                     // eval.py:585,699
    }
    const double avg = (double)__fadd_rn(acc, d_ref) / (double)(geo + 1);           // eval.py:699
    const size_t o = (size_t)r * hw + p;
    P.geo_sum[o] = geo;
    P.depth_avg[o] = avg;
    const bool photo = P.conf[(size_t)ref * hw + p] > P.photomask;                  // eval.py:660
    const bool gm = geo >= P.geomask;                                               // eval.py:702
    unsigned char* M = P.masks + (size_t)r * 3 * hw;
    M[p] = photo;
    M[hw + p] = gm;
    M[2 * hw + p] = photo && gm;                                                    // eval.py:706
    // depth2pts_np (eval.py:253-265): pixel centres at +0.5, x/y of the world point * 1.0531
    const float* Ri = RM + 18;
    const float* t = RM + 27;
    const double gx = (double)x + 0.5, gy = (double)y + 0.5;
    const double c0 = __dsub_rn(__dmul_rn(dot3(Kri, gx, gy, 1.0), avg), (double)t[0]);
    const double c1 = __dsub_rn(__dmul_rn(dot3(Kri + 3, gx, gy, 1.0), avg), (double)t[1]);
    const double c2 = __dsub_rn(__dmul_rn(dot3(Kri + 6, gx, gy, 1.0), avg), (double)t[2]);
    double* W = P.xyz_world + o * 3;
    W[0] = __dmul_rn(dot3(Ri, c0, c1, c2), 1.0531);
    W[1] = __dmul_rn(dot3(Ri + 3, c0, c1, c2), 1.0531);
    W[2] = dot3(Ri + 6, c0, c1, c2);
}

// ---- host: float32 inverse by LU with partial pivoting (what numpy.linalg.inv's sgesv does) ----
// every multiply / add rounded separately (no contraction into fma): oracle/filter_oracle.py fixes
// exactly this order, and the kernel's integer outputs are held to bit-equality with it
template <int N>
static bool inv_f32(const float* a, float* out) {
#pragma clang fp contract(off)
    float lu[N][N];
    int piv[N];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) lu[i][j] = a[i * N + j];
    for (int k = 0; k < N; ++k) {
        int p = k;
        for (int i = k + 1; i < N; ++i)
            if (std::fabs(lu[i][k]) > std::fabs(lu[p][k])) p = i;
        piv[k] = p;
        if (lu[p][k] == 0.f) return false;
        if (p != k)
            for (int j = 0; j < N; ++j) std::swap(lu[k][j], lu[p][j]);
        const float rp = 1.f / lu[k][k];
        for (int i = k + 1; i < N; ++i) lu[i][k] *= rp;
        for (int i = k + 1; i < N; ++i)
            for (int j = k + 1; j < N; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
    }
    for (int c = 0; c < N; ++c) {
        float b[N];
        for (int i = 0; i < N; ++i) b[i] = (i == c) ? 1.f : 0.f;
        for (int k = 0; k < N; ++k) std::swap(b[k], b[piv[k]]);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < i; ++j) b[i] -= lu[i][j] * b[j];
        for (int i = N - 1; i >= 0; --i) {
            for (int j = i + 1; j < N; ++j) b[i] -= lu[i][j] * b[j];
            b[i] /= lu[i][i];
        }
        for (int i = 0; i < N; ++i) out[i * N + c] = b[i];
    }
    return true;
}

static void matmul4_rows3(const float* a, const float* b, float* out12) {
#pragma clang fp contract(off)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
            out12[i * 4 + j] = s;
        }
}

}  // namespace mvs

using namespace mvs;

extern "C" {

int mvs_filter_compose(const float* intrinsics, const float* extrinsics, const int* ref_idx,
                       const int* src_idx, int V, int R, int S, float* ref_mats, float* pair_mats) {
    if (!intrinsics || !extrinsics || !ref_idx || !src_idx || !ref_mats || !pair_mats)
        return fail(MVS_ERR_NULL, "mvs_filter_compose: NULL argument");
    if (V < 1 || R < 1 || S < 1) return fail(MVS_ERR_BAD_SHAPE, "V,R,S = %d,%d,%d", V, R, S);
    for (int r = 0; r < R; ++r) {
        const int ref = ref_idx[r];
        if (ref < 0 || ref >= V) return fail(MVS_ERR_BAD_SHAPE, "ref_idx[%d] = %d outside [0,%d)", r, ref, V);
        const float* Kr = intrinsics + 9 * ref;
        const float* Er = extrinsics + 16 * ref;
        float* RM = ref_mats + (size_t)r * MVS_FILTER_REF_FLOATS;
        float Eri[16], R3[9];
        if (!inv_f32<3>(Kr, RM) || !inv_f32<4>(Er, Eri))
            return fail(MVS_ERR_BAD_SHAPE, "camera %d has a singular intrinsic/extrinsic matrix", ref);
        for (int i = 0; i < 9; ++i) RM[9 + i] = Kr[i];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R3[i * 3 + j] = Er[i * 4 + j];
        if (!inv_f32<3>(R3, RM + 18)) return fail(MVS_ERR_BAD_SHAPE, "camera %d rotation is singular", ref);
        for (int i = 0; i < 3; ++i) RM[27 + i] = Er[i * 4 + 3];
        for (int s = 0; s < S; ++s) {
            const int src = src_idx[r * S + s];
            float* PM = pair_mats + ((size_t)r * S + s) * MVS_FILTER_PAIR_FLOATS;
            if (src < 0) {
                for (int i = 0; i < MVS_FILTER_PAIR_FLOATS; ++i) PM[i] = 0.f;
                continue;
            }
            if (src >= V) return fail(MVS_ERR_BAD_SHAPE, "src_idx[%d][%d] = %d outside [0,%d)", r, s, src, V);
            const float* Ks = intrinsics + 9 * src;
            const float* Es = extrinsics + 16 * src;
            float Esi[16];
            if (!inv_f32<3>(Ks, PM + 21) || !inv_f32<4>(Es, Esi))
                return fail(MVS_ERR_BAD_SHAPE, "camera %d has a singular intrinsic/extrinsic matrix", src);
            matmul4_rows3(Es, Eri, PM);
            for (int i = 0; i < 9; ++i) PM[12 + i] = Ks[i];
            matmul4_rows3(Er, Esi, PM + 30);
        }
    }
    return MVS_OK;
}

int mvs_filter_depth(const float* depth, const float* conf, const float* ref_mats,
                     const float* pair_mats, const int* ref_idx, const int* src_idx, int V, int R,
                     int S, int h, int w, double photomask, int geomask, double condmask_pixel,
                     double condmask_depth, int* geo_sum, double* depth_avg, unsigned char* masks,
                     double* xyz_world, void* stream) {
    if (!depth || !conf || !ref_mats || !pair_mats || !ref_idx || !src_idx || !geo_sum || !depth_avg ||
        !masks || !xyz_world)
        return fail(MVS_ERR_NULL, "mvs_filter_depth: NULL argument");
    if (V < 1 || R < 1 || S < 1 || h < 1 || w < 1 || R > 65535 || h > 32767 || w > 32767)
        return fail(MVS_ERR_BAD_SHAPE, "V,R,S,h,w = %d,%d,%d,%d,%d", V, R, S, h, w);
    FilterParams P{depth, conf, ref_mats, pair_mats, ref_idx, src_idx, V, R, S, h, w,
                   (float)photomask, (float)condmask_depth, condmask_pixel, geomask,
                   geo_sum, depth_avg, masks, xyz_world};
    const int hw = h * w;
    filter_depth_kernel<<<dim3((hw + 255) / 256, R), 256, 0, static_cast<hipStream_t>(stream)>>>(P);
    return check_hip(hipGetLastError(), "filter_depth_kernel");
}

}  // extern "C"
