// warp_variance.hip -- homography warp + variance cost volume for gfx950 (MI355X).
//
// Replaces, fused into one pass and never materialising the sampling grid or the per-view
// warped volumes:
//   models/module.py:96-139   homo_warping (per source view)
//   models/mvsnet.py:145-177  ref-volume repeat, sum / sum-of-squares accumulation, variance
//
// Data layout ("C8-planar", private to the HIP path): a tensor of C channels is stored as C/8
// planes, each a channels-last volume of 8 channels: [C/8][D][h][w][8].  Features arrive NCHW
// fp32 from FeatureNet; nchw_to_c8_kernel re-lays them as [4][N][h][w][8] so that a bilinear tap
// of 8 channels is 32 contiguous bytes and x-adjacent taps share 128-byte lines.  The variance
// volume is written as [4][D][h][w][8], the layout the conv3d kernels stage into LDS one plane
// (= one K-chunk of 8 input channels) at a time.
//
// HBM-bound: algorithmic bytes per map = N*32*h*w*4 (features) + 32*D*h*w*es (volume write).
#include <cstdlib>

#include "mvs_internal.h"
#include "storage.h"
#include "warp_common.h"

#ifndef MVS_ABLATE
#define MVS_ABLATE 0
#endif

namespace mvs {

// ---------------------------------------------------------------------------------------------
// rt[v-1] = rows 0..2 of proj[v] @ inverse(proj[0])      (models/module.py:107-109)
// One thread per source view.  inverse(proj[0]) by cofactors in fp64 (adjugate / determinant), rounded to fp32
// at the end: branch-free and fully unrolled, so the 4x4 system stays in registers -- the Gauss-Jordan form with
// partial pivoting indexed its rows dynamically (scratch memory) and made this side block the longest-running
// block of the transpose launch (12.4 us for 6 us of transposing).  A singular ref_proj gives inf / NaN like
// torch.inverse's result would.
// ---------------------------------------------------------------------------------------------
__device__ void relative_proj_view(const float* __restrict__ proj, float* __restrict__ rt, int N, int v) {
    if (v >= N) return;
    double m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = (double)proj[i];
    // 2x2 minors of rows (0,1) and rows (2,3)
    const double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const double s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const double c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const double id = 1.0 / det;
    double inv[16];
    inv[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    inv[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    inv[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    inv[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    inv[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    inv[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    inv[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    inv[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    inv[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    inv[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    inv[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
    inv[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    inv[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id;
    inv[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    inv[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id;
    inv[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
    const float* ps = proj + (size_t)v * 16;
    float* o = rt + (size_t)(v - 1) * 12;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double row[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += (double)ps[i * 4 + k] * inv[k * 4 + j];
            row[j] = acc;
        }
        o[i * 3 + 0] = (float)row[0];
        o[i * 3 + 1] = (float)row[1];
        o[i * 3 + 2] = (float)row[2];
        o[9 + i] = (float)row[3];
    }
}

__global__ void relative_proj_kernel(const float* __restrict__ proj, float* __restrict__ rt, int N) {
    relative_proj_view(proj, rt, N, blockIdx.x * blockDim.x + threadIdx.x + 1);
}

// ---------------------------------------------------------------------------------------------
// [N][C=32][h][w] -> C8-planar [4][N][h][w][8]; one block transposes 32 channels x 64 pixels
// through LDS.
// ---------------------------------------------------------------------------------------------
// proj != NULL: one extra block row (blockIdx.y == N) computes the relative projections of
// models/module.py:107-109 on the side -- a separate 1-block launch costs 7-8 us of a 950 us map.
template <int DT>
__global__ __launch_bounds__(256) void nchw_to_c8_kernel(const float* __restrict__ in,
                                                         void* __restrict__ out, int N, int hw,
                                                         const float* __restrict__ proj, float* __restrict__ rt) {
    __shared__ float tile[32][65];
    if (blockIdx.y == (unsigned)N) {   // only launched when proj != NULL
        if (blockIdx.x == 0) relative_proj_view(proj, rt, N, (int)threadIdx.x + 1);
        return;
    }
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const float* src = in + (size_t)n * 32 * hw;
#pragma unroll
    for (int c = ty; c < 32; c += 4) {
        const int p = p0 + tx;
        tile[c][tx] = (p < hw) ? src[(size_t)c * hw + p] : 0.0f;
    }
    __syncthreads();
    const int c8 = threadIdx.x & 7, pp = threadIdx.x >> 3;  // 8 channels x 32 pixels per pass
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int q = pass * 32 + pp, p = p0 + q;
            if (p < hw) St<DT>::store1(out, (((size_t)pl * N + n) * hw + p) * 8 + c8, tile[pl * 8 + c8][q]);
        }
}

int launch_nchw_to_c8(const float* in, void* out, int N, int C, int h, int w, int dtype, hipStream_t s,
                      const float* proj, float* rt) {
    (void)C;
    const int hw = h * w;
    if (N > 256) proj = nullptr;   // the side block has one thread per source view
    dim3 grid((hw + 63) / 64, proj ? N + 1 : N);
    if (dtype == MVS_F32) nchw_to_c8_kernel<MVS_F32><<<grid, 256, 0, s>>>(in, out, N, hw, proj, rt);
    else if (dtype == MVS_F16) nchw_to_c8_kernel<MVS_F16><<<grid, 256, 0, s>>>(in, out, N, hw, proj, rt);
    else if (dtype == MVS_BF16) nchw_to_c8_kernel<MVS_BF16><<<grid, 256, 0, s>>>(in, out, N, hw, proj, rt);
    else return fail(MVS_ERR_BAD_DTYPE, "nchw_to_c8: unknown dtype %d", dtype);
    return check_hip(hipGetLastError(), "nchw_to_c8 launch");
}

int launch_relative_proj(const float* proj, float* rt, int N, hipStream_t s) {
    relative_proj_kernel<<<1, 64, 0, s>>>(proj, rt, N);
    return check_hip(hipGetLastError(), "relative_proj launch");
}

// ---------------------------------------------------------------------------------------------
// Fused warp + variance.  Thread = (pixel, half): it owns channels {8*pl + 4*half .. +3} of its
// pixel for all four channel planes pl, so the sampling coordinates of a (pixel, depth, view)
// are computed once and reused for 16 channels.  A wave covers 32 consecutive pixels; each of
// its stores writes 32 px x 32 B = 1 KiB contiguous into one plane of the C8-planar volume.
// Each block walks a slab of depths so consecutive depths re-use the same source lines from
// L1/L2 (the epipolar shift per depth step is a fraction of a pixel).
//   feats_p [4][N][h][w][8], rt [(N-1)][12], dv [D] -> var [4][D][h][w][8]
// ---------------------------------------------------------------------------------------------
#ifndef MVS_WARP_SLAB
#define MVS_WARP_SLAB 8
#endif
constexpr int kWarpDepthSlab = MVS_WARP_SLAB;
constexpr int kWarpPixPerBlock = 128;

template <int DT, bool DEPTH_FASTEST>
__global__ __launch_bounds__(256) void warp_variance_kernel(const float* __restrict__ feats_p,
                                                            const float* __restrict__ rt,
                                                            const float* __restrict__ dv,
                                                            void* __restrict__ var, int N, int D,
                                                            int h, int w) {
    const int half = threadIdx.x & 1;
    const int hw = h * w;
    // Block order.  DEPTH_FASTEST (feature maps larger than the L2s, e.g. 1600x1184 inputs): x =
    // depth slab, y = pixel block, so blocks resident together work on the same pixels at
    // neighbouring depths and share their source footprints in L2 (3.4 -> 2.0 ms at cfg3).
    // Otherwise x = pixel block: the features stay L2-resident anyway and the volume is written
    // plane by plane.
    const int pblk = DEPTH_FASTEST ? blockIdx.y : blockIdx.x;
    const int dblk = DEPTH_FASTEST ? blockIdx.x : blockIdx.y;
    const int p = pblk * kWarpPixPerBlock + (threadIdx.x >> 1);
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const int d0 = dblk * kWarpDepthSlab;
    const int d1 = min(d0 + kWarpDepthSlab, D);
    const size_t plane_stride = (size_t)N * hw * 8;  // floats between channel planes
    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float fx = (float)x, fy = (float)y;
    const float inv_n = 1.0f / (float)N;
    float4 ref[4];
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
        ref[pl] = *reinterpret_cast<const float4*>(feats_p + pl * plane_stride + (size_t)p * 8 + 4 * half);
    const size_t V0 = (size_t)D * hw;
    for (int d = d0; d < d1; ++d) {
        const float depth = dv[d];
        float4 S[4], Q[4];
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            S[pl] = ref[pl];
            Q[pl] = make_float4(ref[pl].x * ref[pl].x, ref[pl].y * ref[pl].y, ref[pl].z * ref[pl].z,
                                ref[pl].w * ref[pl].w);
        }
        for (int v = 1; v < N; ++v) {
            const float* r = rt + (size_t)(v - 1) * 12;
            const float qx = fmaf(r[0], fx, fmaf(r[1], fy, r[2]));
            const float qy = fmaf(r[3], fx, fmaf(r[4], fy, r[5]));
            const float qz = fmaf(r[6], fx, fmaf(r[7], fy, r[8]));
            const Samp t = make_samp(qx, qy, qz, r[9], r[10], r[11], depth, sx, sy, h, w, 0, 0, w, h);
            const float* f0 = feats_p + (size_t)v * hw * 8 + 4 * half;
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) {
                const float* f = f0 + pl * plane_stride;
#if MVS_ABLATE == 4  // diagnostic build: no gathers (timing floor of VALU + volume write)
                (void)f;
                const float4 a = ref[pl], b = ref[pl], c = ref[pl], e = ref[pl];
#else
                const float4 a = *reinterpret_cast<const float4*>(f + (size_t)t.o00 * 8);
                const float4 b = *reinterpret_cast<const float4*>(f + (size_t)t.o01 * 8);
                const float4 c = *reinterpret_cast<const float4*>(f + (size_t)t.o10 * 8);
                const float4 e = *reinterpret_cast<const float4*>(f + (size_t)t.o11 * 8);
#endif
                float4 wv;
                wv.x = fmaf(a.x, t.w00, fmaf(b.x, t.w01, fmaf(c.x, t.w10, e.x * t.w11)));
                wv.y = fmaf(a.y, t.w00, fmaf(b.y, t.w01, fmaf(c.y, t.w10, e.y * t.w11)));
                wv.z = fmaf(a.z, t.w00, fmaf(b.z, t.w01, fmaf(c.z, t.w10, e.z * t.w11)));
                wv.w = fmaf(a.w, t.w00, fmaf(b.w, t.w01, fmaf(c.w, t.w10, e.w * t.w11)));
                S[pl].x += wv.x; S[pl].y += wv.y; S[pl].z += wv.z; S[pl].w += wv.w;
                Q[pl].x = fmaf(wv.x, wv.x, Q[pl].x); Q[pl].y = fmaf(wv.y, wv.y, Q[pl].y);
                Q[pl].z = fmaf(wv.z, wv.z, Q[pl].z); Q[pl].w = fmaf(wv.w, wv.w, Q[pl].w);
            }
        }
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            // var = Q/N - (S/N)^2        (models/mvsnet.py:177)
            float4 o;
            float m;
            m = S[pl].x * inv_n; o.x = fmaf(-m, m, Q[pl].x * inv_n);
            m = S[pl].y * inv_n; o.y = fmaf(-m, m, Q[pl].y * inv_n);
            m = S[pl].z * inv_n; o.z = fmaf(-m, m, Q[pl].z * inv_n);
            m = S[pl].w * inv_n; o.w = fmaf(-m, m, Q[pl].w * inv_n);
            St<DT>::store4(var, ((size_t)pl * V0 + (size_t)d * hw + p) * 8 + 4 * half,
                           (f32x4){o.x, o.y, o.z, o.w});
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 16-bit modes: the C8-planar feature copy is stored in the storage dtype too, so one 16-byte
// load is a whole 8-channel tap -- half the bytes through the vector-L1 path that bounds the fp32
// kernel.  Thread = (pixel, plane pair): 2 planes x 8 channels; arithmetic stays fp32.
//   feats16 [4][N][h][w][8] (DT), rt, dv -> var [4][D][h][w][8] (DT)
// ---------------------------------------------------------------------------------------------
template <int DT, bool DEPTH_FASTEST>
__global__ __launch_bounds__(256) void warp_variance16_kernel(const void* __restrict__ feats16,
                                                              const float* __restrict__ rt,
                                                              const float* __restrict__ dv,
                                                              void* __restrict__ var, int N, int D,
                                                              int h, int w) {
    const int hp = threadIdx.x & 1;  // planes 2hp, 2hp+1
    const int hw = h * w;
    const int pblk = DEPTH_FASTEST ? blockIdx.y : blockIdx.x;
    const int dblk = DEPTH_FASTEST ? blockIdx.x : blockIdx.y;
    const int p = pblk * kWarpPixPerBlock + (threadIdx.x >> 1);
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const int d0 = dblk * kWarpDepthSlab;
    const int d1 = min(d0 + kWarpDepthSlab, D);
    const size_t plane_stride = (size_t)N * hw * 8;  // elements between channel planes
    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float fx = (float)x, fy = (float)y;
    const float inv_n = 1.0f / (float)N;
    const size_t V0 = (size_t)D * hw;
    float ref[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) load8_16<DT>(feats16, (size_t)(2 * hp + q) * plane_stride + (size_t)p * 8, ref[q]);
    for (int d = d0; d < d1; ++d) {
        const float depth = dv[d];
        float S[2][8], Q[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 8; ++c) { S[q][c] = ref[q][c]; Q[q][c] = ref[q][c] * ref[q][c]; }
        for (int v = 1; v < N; ++v) {
            const float* r = rt + (size_t)(v - 1) * 12;
            const float qx = fmaf(r[0], fx, fmaf(r[1], fy, r[2]));
            const float qy = fmaf(r[3], fx, fmaf(r[4], fy, r[5]));
            const float qz = fmaf(r[6], fx, fmaf(r[7], fy, r[8]));
            const Samp t = make_samp(qx, qy, qz, r[9], r[10], r[11], depth, sx, sy, h, w, 0, 0, w, h);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const size_t base = (size_t)(2 * hp + q) * plane_stride + (size_t)v * hw * 8;
                float a[8], b[8], c[8], e[8];
                load8_16<DT>(feats16, base + (size_t)t.o00 * 8, a);
                load8_16<DT>(feats16, base + (size_t)t.o01 * 8, b);
                load8_16<DT>(feats16, base + (size_t)t.o10 * 8, c);
                load8_16<DT>(feats16, base + (size_t)t.o11 * 8, e);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float wv = fmaf(a[k], t.w00, fmaf(b[k], t.w01, fmaf(c[k], t.w10, e[k] * t.w11)));
                    S[q][k] += wv;
                    Q[q][k] = fmaf(wv, wv, Q[q][k]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float m = S[q][k] * inv_n;
                o[k] = fmaf(-m, m, Q[q][k] * inv_n);  // var = Q/N - (S/N)^2   (models/mvsnet.py:177)
            }
            store8_16<DT>(var, ((size_t)(2 * hp + q) * V0 + (size_t)d * hw + p) * 8, o);
        }
    }
}

// MVS_WARP_DEPTH_FASTEST=1 forces the depth-slab-fastest block order (default: only when the feature
// maps exceed the L2s), so tests can reach it at small shapes
static bool force_depth_fastest() {
    static const bool f = [] {
        const char* e = getenv("MVS_WARP_DEPTH_FASTEST");
        return e && e[0] == '1';
    }();
    return f;
}

int launch_warp_variance16(const void* feats16, const float* rt, const float* dv, void* var, int N, int D,
                           int h, int w, int dtype, hipStream_t s) {
    if (h < 2 || w < 2) return fail(MVS_ERR_BAD_SHAPE, "warp_variance: h,w must be >= 2");
    // The tap-cache kernel (second form, warp_variance_tc.hip) is the default here too: cfg5 0.20 ->
    // 0.15 ms, cfg3 1.84 -> 1.39 ms against the plain kernel below (MVS_WARP_TC16=0 selects it; the
    // FIRST form of the tap-cache kernel had been slower than the plain kernel with 16-bit features).
    static const bool use_tc = [] {
        const char* e = getenv("MVS_WARP_TC16");
        return !(e && e[0] == '0');
    }();
    if (use_tc && warp_tc_fits(N, D, h, w, 2, 2))
        return launch_warp_variance_tc16(feats16, rt, dv, var, N, D, h, w, dtype, s);
    const unsigned nd = (D + kWarpDepthSlab - 1) / kWarpDepthSlab;
    const unsigned np = (h * w + kWarpPixPerBlock - 1) / kWarpPixPerBlock;
    const bool depth_fastest = force_depth_fastest() || (size_t)N * h * w * 64 > ((size_t)24 << 20);
    const dim3 grid = depth_fastest ? dim3(nd, np) : dim3(np, nd);
    if (dtype == MVS_F16) {
        if (depth_fastest) warp_variance16_kernel<MVS_F16, true><<<grid, 256, 0, s>>>(feats16, rt, dv, var, N, D, h, w);
        else warp_variance16_kernel<MVS_F16, false><<<grid, 256, 0, s>>>(feats16, rt, dv, var, N, D, h, w);
    } else if (dtype == MVS_BF16) {
        if (depth_fastest) warp_variance16_kernel<MVS_BF16, true><<<grid, 256, 0, s>>>(feats16, rt, dv, var, N, D, h, w);
        else warp_variance16_kernel<MVS_BF16, false><<<grid, 256, 0, s>>>(feats16, rt, dv, var, N, D, h, w);
    } else {
        return fail(MVS_ERR_BAD_DTYPE, "warp_variance16 needs fp16 or bf16 (dtype %d)", dtype);
    }
    return check_hip(hipGetLastError(), "warp_variance16 launch");
}

int launch_warp_variance(const float* feats_p, const float* rt, const float* dv, void* var, int N,
                         int D, int h, int w, int dtype, hipStream_t s) {
    if (h < 2 || w < 2) return fail(MVS_ERR_BAD_SHAPE, "warp_variance: h,w must be >= 2");
    // Tap-cache kernel (warp_variance_tc.hip) for 2..5 views; MVS_WARP_TC=0 keeps the plain gather
    static const bool use_tc = [] {
        const char* e = getenv("MVS_WARP_TC");
        return !(e && e[0] == '0');
    }();
    if (use_tc && warp_tc_fits(N, D, h, w, 4, dtype == MVS_F32 ? 4 : 2))
        return launch_warp_variance_tc(feats_p, rt, dv, var, N, D, h, w, dtype, s);
    const unsigned nd = (D + kWarpDepthSlab - 1) / kWarpDepthSlab;
    const unsigned np = (h * w + kWarpPixPerBlock - 1) / kWarpPixPerBlock;
    float* v = static_cast<float*>(var);
    // all views' features (N x 32 channels x fp32) against the 32 MB of aggregate L2
    const bool depth_fastest = force_depth_fastest() || (size_t)N * h * w * 128 > ((size_t)24 << 20);
    if (depth_fastest) {
        dim3 grid(nd, np);
        if (dtype == MVS_F32) warp_variance_kernel<MVS_F32, true><<<grid, 256, 0, s>>>(feats_p, rt, dv, v, N, D, h, w);
        else if (dtype == MVS_F16) warp_variance_kernel<MVS_F16, true><<<grid, 256, 0, s>>>(feats_p, rt, dv, v, N, D, h, w);
        else if (dtype == MVS_BF16) warp_variance_kernel<MVS_BF16, true><<<grid, 256, 0, s>>>(feats_p, rt, dv, v, N, D, h, w);
        else return fail(MVS_ERR_BAD_DTYPE, "warp_variance: unknown dtype %d", dtype);
    } else {
        dim3 grid(np, nd);
        if (dtype == MVS_F32) warp_variance_kernel<MVS_F32, false><<<grid, 256, 0, s>>>(feats_p, rt, dv, v, N, D, h, w);
        else if (dtype == MVS_F16) warp_variance_kernel<MVS_F16, false><<<grid, 256, 0, s>>>(feats_p, rt, dv, v, N, D, h, w);
        else if (dtype == MVS_BF16) warp_variance_kernel<MVS_BF16, false><<<grid, 256, 0, s>>>(feats_p, rt, dv, v, N, D, h, w);
        else return fail(MVS_ERR_BAD_DTYPE, "warp_variance: unknown dtype %d", dtype);
    }
    return check_hip(hipGetLastError(), "warp_variance launch");
}

// ---------------------------------------------------------------------------------------------
// Stand-alone homo_warping with reference layouts: fea [C][h][w] -> out [C][D][h][w]
// (models/module.py:96-139).  Thread = one (d,y,x); loops over channels (coalesced along x).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void homo_warp_kernel(const float* __restrict__ fea,
                                                        const float* __restrict__ rt,
                                                        const float* __restrict__ dv,
                                                        float* __restrict__ out, int C, int D, int h,
                                                        int w) {
    const size_t hw = (size_t)h * w;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)D * hw) return;
    const int x = (int)(i % w), y = (int)((i / w) % h), d = (int)(i / hw);
    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float fx = (float)x, fy = (float)y;
    const float qx = fmaf(rt[0], fx, fmaf(rt[1], fy, rt[2]));
    const float qy = fmaf(rt[3], fx, fmaf(rt[4], fy, rt[5]));
    const float qz = fmaf(rt[6], fx, fmaf(rt[7], fy, rt[8]));
    const Samp t = make_samp(qx, qy, qz, rt[9], rt[10], rt[11], dv[d], sx, sy, h, w, 0, 0, w, h);
    for (int c = 0; c < C; ++c) {
        const float* f = fea + (size_t)c * hw;
        out[(size_t)c * D * hw + i] =
            fmaf(f[t.o00], t.w00, fmaf(f[t.o01], t.w01, fmaf(f[t.o10], t.w10, f[t.o11] * t.w11)));
    }
}

int launch_homo_warp(const float* fea, const float* rt, const float* dv, float* out, int C, int D,
                     int h, int w, hipStream_t s) {
    const size_t n = (size_t)D * h * w;
    homo_warp_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(fea, rt, dv, out, C, D, h, w);
    return check_hip(hipGetLastError(), "homo_warp launch");
}

}  // namespace mvs
