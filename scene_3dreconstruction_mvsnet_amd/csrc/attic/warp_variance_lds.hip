// warp_variance_lds.hip -- LDS-staged homography warp + variance cost volume (gfx950).
//
// Same contract as warp_variance_kernel (warp_variance.hip; reference models/module.py:96-139 and
// models/mvsnet.py:145-177) but the bilinear gathers are served from LDS instead of L1: the
// global-gather version is bound by the CU's 64 B/clk vector-L1 path (every source texel is
// fetched ~4x by neighbouring pixels and once more per depth), LDS delivers 256 B/clk.
//
// Block (512 threads) = 16 x 8 reference pixels x 4 depth planes x 2 channel planes (16 ch).
//   thread (pixel, e): samples depth d0+e for 8+8 channels, so sampling coordinates are
//   computed exactly once per (pixel, depth, view) in the block.
// Per source view:
//   1. wave 0 projects the 8 corners of the (tile x depth-slab) box -- the map is linear-
//      fractional in x, y and d separately, so its extremes sit on the box vertices -- and
//      reduces them with wavefront shuffles to the source bounding box (all views at once,
//      8 lanes per view);
//   2. the block stages that box of the view's two channel planes into LDS (row segments are
//      contiguous in the C8-planar feature layout: coalesced);
//   3. every thread takes its 4 taps x 16 channels from LDS and accumulates sum / sum of squares
//      in registers across views.
// Views whose box does not fit (strong rotation/zoom, points behind the camera) fall back to
// global gathers for that view only, with identical arithmetic.
#include "mvs_internal.h"
#include "warp_common.h"

namespace mvs {

namespace wl {
constexpr int TW = 16, TH = 8;     // reference-pixel tile
constexpr int DS = 4;              // depth planes per block
constexpr int MAX_BW = 32;         // widest stageable source box (texels): one row = 64 pieces
constexpr int MAX_BH = 16;         // tallest stageable box: 2 planes x 16 rows / 8 waves = 4 rows/wave
constexpr int MAX_TEXELS = MAX_BW * MAX_BH;  // per plane; 2 planes x 512 x 32 B = 32 KiB per buffer
constexpr int NWAVES = 8;
constexpr int ROWS_PER_WAVE = 2 * MAX_BH / NWAVES;
constexpr int MAX_VIEWS = 64;
}  // namespace wl

__global__ __launch_bounds__(512) void warp_variance_lds_kernel(const float* __restrict__ feats_p,
                                                                const float* __restrict__ rt,
                                                                const float* __restrict__ dv,
                                                                float* __restrict__ var, int N,
                                                                int D, int h, int w) {
    using namespace wl;
    __shared__ __attribute__((aligned(16))) float tile[2][2 * MAX_TEXELS * 8];  // 2 x 32 KiB
    __shared__ int4 bbox[MAX_VIEWS];  // x_lo, y_lo, bw, bh (bh < 0: fall back to global gathers)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = tid & 3, pix = tid >> 2;
    const int tilesx = (w + TW - 1) / TW;
    const int tx0 = (blockIdx.x % tilesx) * TW, ty0 = (blockIdx.x / tilesx) * TH;
    const int x = tx0 + (pix & (TW - 1)), y = ty0 + (pix >> 4);
    const int d0 = blockIdx.y * DS;
    const int pp = blockIdx.z;  // channel planes 2pp, 2pp+1
    const int hw = h * w;
    const bool active = x < w && y < h;
    const int xs = min(x, w - 1), ys = min(y, h - 1);  // clamped coordinates keep idle lanes harmless
    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const size_t plane_stride = (size_t)N * hw * 8;  // floats between channel planes of the features
    const float* fplane = feats_p + (size_t)(2 * pp) * plane_stride;

    // ---- source bounding boxes of all views: 8 lanes per view, shuffle min/max reduction ----
    if (wave == 0) {
        const int c = lane & 7;
        const float cxp = (float)((c & 1) ? min(tx0 + TW - 1, w - 1) : tx0);
        const float cyp = (float)((c & 2) ? min(ty0 + TH - 1, h - 1) : ty0);
        const float cd = dv[(c & 4) ? min(d0 + DS - 1, D - 1) : d0];
        for (int v0 = 1; v0 < N; v0 += 8) {
            const int v = v0 + (lane >> 3);
            float ix = 0.f, iy = 0.f, Z = 1.f;
            if (v < N) {
                const float* r = rt + (size_t)(v - 1) * 12;
                const float X = fmaf(fmaf(r[0], cxp, fmaf(r[1], cyp, r[2])), cd, r[9]);
                const float Y = fmaf(fmaf(r[3], cxp, fmaf(r[4], cyp, r[5])), cd, r[10]);
                Z = fmaf(fmaf(r[6], cxp, fmaf(r[7], cyp, r[8])), cd, r[11]);
                ix = (X / Z) * sx - 0.5f;
                iy = (Y / Z) * sy - 0.5f;
            }
            float xmin = ix, xmax = ix, ymin = iy, ymax = iy, zmin = Z;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                xmin = fminf(xmin, __shfl_xor(xmin, m)); xmax = fmaxf(xmax, __shfl_xor(xmax, m));
                ymin = fminf(ymin, __shfl_xor(ymin, m)); ymax = fmaxf(ymax, __shfl_xor(ymax, m));
                zmin = fminf(zmin, __shfl_xor(zmin, m));
            }
            if (c == 0 && v < N) {
                // fminf/fmaxf drop NaNs, so test finiteness on the sum of the extremes
                const bool ok = zmin > 1e-20f && fabsf(xmin) + fabsf(xmax) + fabsf(ymin) + fabsf(ymax) < 1.0e9f;
                int x_lo = 0, y_lo = 0, x_hi = w - 1, y_hi = h - 1;
                if (ok) {
                    x_lo = (int)floorf(fminf(fmaxf(xmin, -4.f), (float)w + 4.f)) - 1;
                    x_hi = (int)floorf(fminf(fmaxf(xmax, -4.f), (float)w + 4.f)) + 2;
                    y_lo = (int)floorf(fminf(fmaxf(ymin, -4.f), (float)h + 4.f)) - 1;
                    y_hi = (int)floorf(fminf(fmaxf(ymax, -4.f), (float)h + 4.f)) + 2;
                    x_lo = min(max(x_lo, 0), w - 2); x_hi = min(max(x_hi, x_lo + 1), w - 1);
                    y_lo = min(max(y_lo, 0), h - 2); y_hi = min(max(y_hi, y_lo + 1), h - 1);
                }
                const int bw = x_hi - x_lo + 1, bh = y_hi - y_lo + 1;
                const bool fits = ok && bw <= MAX_BW && bh <= MAX_BH;
                bbox[v - 1] = fits ? make_int4(x_lo, y_lo, bw, bh) : make_int4(0, 0, w, -h);
            }
        }
    }

    // ---- accumulators: depth d0 + e; planes 2pp, 2pp+1; 8 channels each ----
    float4 S[2][2], Q[2][2];  // [plane][lo/hi]
    {
        const float* rp = fplane + ((size_t)ys * w + xs) * 8;  // view 0 = reference feature
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const float4 lo = *reinterpret_cast<const float4*>(rp + pl * plane_stride);
            const float4 hi = *reinterpret_cast<const float4*>(rp + pl * plane_stride + 4);
            S[pl][0] = lo; S[pl][1] = hi;
            Q[pl][0] = make_float4(lo.x * lo.x, lo.y * lo.y, lo.z * lo.z, lo.w * lo.w);
            Q[pl][1] = make_float4(hi.x * hi.x, hi.y * hi.y, hi.z * hi.z, hi.w * hi.w);
        }
    }
    const float fx = (float)xs, fy = (float)ys;
    const float depth = dv[min(d0 + e, D - 1)];
    __syncthreads();  // bbox[] visible

    // Software pipeline over views: the global loads that stage view v+1 are issued before view
    // v is sampled and land in LDS (other buffer) after it; one barrier per view.
    const int stx = lane >> 1, shq = lane & 1;
    float4 st[ROWS_PER_WAVE];

#define MVS_STAGE_LOAD(V)                                                                          \
    {                                                                                              \
        const int4 nb = bbox[(V) - 1];                                                             \
        const float* np = fplane + (size_t)(V) * hw * 8;                                           \
        _Pragma("unroll") for (int i = 0; i < ROWS_PER_WAVE; ++i) {                                \
            const int row = wave + NWAVES * i;                                                     \
            const int pl = row >= nb.w ? 1 : 0, ty = row - pl * nb.w;                              \
            st[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                               \
            if (row < 2 * nb.w && stx < nb.z)                                                      \
                st[i] = *reinterpret_cast<const float4*>(                                          \
                    np + pl * plane_stride + ((size_t)(nb.y + ty) * w + nb.x + stx) * 8 + shq * 4); \
        }                                                                                          \
    }
#define MVS_STAGE_STORE(V)                                                                         \
    {                                                                                              \
        const int4 nb = bbox[(V) - 1];                                                             \
        float* nbuf = tile[((V) - 1) & 1];                                                         \
        _Pragma("unroll") for (int i = 0; i < ROWS_PER_WAVE; ++i) {                                \
            const int row = wave + NWAVES * i;                                                     \
            const int pl = row >= nb.w ? 1 : 0, ty = row - pl * nb.w;                              \
            if (row < 2 * nb.w && stx < nb.z)                                                      \
                *reinterpret_cast<float4*>(nbuf + ((pl * nb.w + ty) * nb.z + stx) * 8 + shq * 4) = st[i]; \
        }                                                                                          \
    }

    // Software pipeline over views: the global loads that stage view v+1 are issued before view
    // v is sampled and land in LDS (other buffer) after it; one barrier per view.  (For views
    // that fall back to global gathers nb.w < 0, so every staging predicate is false.)
    if (N > 1) {
        MVS_STAGE_LOAD(1)
        MVS_STAGE_STORE(1)
    }
    __syncthreads();

    for (int v = 1; v < N; ++v) {
        const int4 bb = bbox[v - 1];
        const bool use_lds = bb.w > 0;  // block-uniform
        const int bw = bb.z, bh = use_lds ? bb.w : -bb.w;
        const float* vplane = fplane + (size_t)v * hw * 8;
        const float* buf = tile[(v - 1) & 1];
        if (v + 1 < N) MVS_STAGE_LOAD(v + 1)
        const float* r = rt + (size_t)(v - 1) * 12;
        const float qx = fmaf(r[0], fx, fmaf(r[1], fy, r[2]));
        const float qy = fmaf(r[3], fx, fmaf(r[4], fy, r[5]));
        const float qz = fmaf(r[6], fx, fmaf(r[7], fy, r[8]));
        const Samp s = make_samp(qx, qy, qz, r[9], r[10], r[11], depth, sx, sy, h, w, bb.x, bb.y, bw, bh);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            float4 lo, hi;
            if (use_lds) sample8(buf + pl * bh * bw * 8, s, lo, hi);
            else sample8(vplane + pl * plane_stride, s, lo, hi);
            accum(S[pl][0], Q[pl][0], lo);
            accum(S[pl][1], Q[pl][1], hi);
        }
        if (v + 1 < N) MVS_STAGE_STORE(v + 1)  // into the buffer view v-1 used (all waves are past it)
        __syncthreads();
    }
#undef MVS_STAGE_LOAD
#undef MVS_STAGE_STORE

    if (!active) return;
    const float inv_n = 1.0f / (float)N;
    const size_t V0 = (size_t)D * hw;
    const int d = d0 + e;
    if (d >= D) return;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        float* o = var + ((size_t)(2 * pp + pl) * V0 + (size_t)d * hw + (size_t)y * w + x) * 8;
        *reinterpret_cast<float4*>(o) = variance4(S[pl][0], Q[pl][0], inv_n);
        *reinterpret_cast<float4*>(o + 4) = variance4(S[pl][1], Q[pl][1], inv_n);
    }
}

int launch_warp_variance_lds(const float* feats_p, const float* rt, const float* dv, void* var, int N,
                             int D, int h, int w, hipStream_t s) {
    using namespace wl;
    if (N > MAX_VIEWS) return fail(MVS_ERR_BAD_SHAPE, "warp_variance_lds: N=%d > %d", N, MAX_VIEWS);
    dim3 grid(((w + TW - 1) / TW) * ((h + TH - 1) / TH), (D + DS - 1) / DS, 2);
    warp_variance_lds_kernel<<<grid, 512, 0, s>>>(feats_p, rt, dv, static_cast<float*>(var), N, D, h, w);
    return check_hip(hipGetLastError(), "warp_variance_lds launch");
}

}  // namespace mvs
