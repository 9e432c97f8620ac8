// featnet.hip -- FeatureNet (reference models/mvsnet.py:10-30; block ConvBnReLU of
// models/module.py:6-13) as fp32-MFMA implicit-GEMM 2-D convolutions.
//
//   conv0 3->8 k3 | conv1 8->8 k3 | conv2 8->16 k5 s2 | conv3,4 16->16 k3 | conv5 16->32 k5 s2 |
//   conv6 32->32 k3 | feature 32->32 k3 (+bias, no BN / ReLU)             BN (eval) folded on the host
//
// Layout: every activation is C8-planar over the N views, [C/8][N][H][W][8] -- exactly the layout
// the warp kernel gathers from, so the last layer writes the path's feature buffer directly and the
// NCHW -> C8 transpose of the PyTorch hand-off disappears.  The input images are read as NCHW
// [N][3][H][W] by conv0's staging code (channels 3..7 of its only K-chunk are zero).
//
// Kernel scheme (same as conv3d_mfma.hip, one dimension less):
//   M : output pixels, one MFMA tile = 2(y) x 8(x)          N : 16 output channels per N-tile
//   K : (tap, ci) in chunks of 8 input channels (one plane); a k-step of 16 = 2 taps x 8 channels,
//       9 taps -> 5 k-steps, 25 taps -> 13 k-steps
//   block = 4 waves = NT N-tiles x MG M-groups, block tile BY x BX M-tiles; per chunk the halo tile
//   is staged in LDS (voxel stride 32 B for stride 1, 48 B for stride 2; row pitch a multiple of 8
//   voxels: conflict-free ds_read_b128), the (chunk, N-tile) B panel lives in registers; staging of
//   chunk c+1 overlaps the MFMAs of chunk c with unconditional loads (counted vmcnt).
// Cout = 8 layers (conv0, conv1) run with a half-empty N-tile: they are HBM-bound at full
// resolution (2 x 52 MB at 5 x 512 x 640), not MFMA-bound.
#include <cstdlib>

#include "mvs_internal.h"
#include "storage.h"

namespace mvs {

template <int CIN, int COUT, int KW, int S, int BY, int BX>
struct FConv {
    static constexpr int NCH = (CIN + 7) / 8;
    static constexpr int NT = (COUT + 15) / 16;
    static constexpr int MG = 4 / NT;
    static constexpr int MT = BY * BX;
    static constexpr int MPW = MT / MG;
    static constexpr int TAPS = KW * KW;
    static constexpr int KS = (TAPS + 1) / 2;
    static constexpr int PAD = KW / 2;
    static constexpr int VS = (S == 1) ? 8 : 12;
    static constexpr int HY = (2 * BY - 1) * S + KW, HX = (8 * BX - 1) * S + KW;
    static constexpr int HXP = (HX + 7) / 8 * 8;
    static constexpr int TILE_FLOATS = HY * HXP * VS;
    static constexpr int NVOX = HY * HX;
    static constexpr int NPIECE = NVOX * 2;
    static constexpr int PPT = (NPIECE + 255) / 256;
    static constexpr int VPT = (NVOX + 255) / 256;  // image-input staging: voxels per thread
    static_assert(NT == 1 || NT == 2 || NT == 4, "COUT must be <= 64");
    static_assert(MT % MG == 0, "block tile must split evenly over the M-groups");
    static_assert(PPT <= 32, "piece mask is 32 bits");
    static constexpr int tap_off(int tap) {  // LDS float offset of tap (ky,kx); taps >= TAPS = padding
        const int t = tap >= TAPS ? TAPS - 1 : tap;
        return ((t / KW) * HXP + t % KW) * VS;
    }
};

template <int CIN, int COUT, int KW, int S, int BY, int BX, bool RELU, bool IMG_IN>
__global__ __launch_bounds__(256) void fconv_mfma_kernel(
    const float* __restrict__ x,     // IMG_IN ? [N][3][Hi][Wi] : [CIN/8][N][Hi][Wi][8]
    const float* __restrict__ bp,    // [NCH][NT][KS][64][4]
    const float* __restrict__ bias,  // [16 NT]
    float* __restrict__ y,           // [COUT/8][N][Ho][Wo][8]
    int N, int Hi, int Wi, int Ho, int Wo) {
    using G = FConv<CIN, COUT, KW, S, BY, BX>;
    // epilogue staging tile: the block's 2BY x 8BX output pixels x (padded) COUT channels, rows padded (16 floats per 8
    // pixels, row pitch = 16 mod 32 floats) so that the accumulator scatter is conflict-free
    constexpr int OY = 2 * BY, OX = 8 * BX, CP = 16 * G::NT;
    constexpr int RP0 = OX * CP + 16 * (OX / 8), RP = (RP0 % 32 == 16) ? RP0 : RP0 + 16;
    constexpr int OUT_FLOATS = OY * RP, NPL = (COUT + 7) / 8, NUNIT = OY * OX * NPL, UPT = (NUNIT + 255) / 256;
    constexpr int LDS_FLOATS = G::TILE_FLOATS > OUT_FLOATS ? G::TILE_FLOATS : OUT_FLOATS;
    __shared__ __attribute__((aligned(16))) float tile[LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NT, mg = wave / G::NT;
    const int nbx = (Wo + 8 * BX - 1) / (8 * BX);
    // workgroups are dealt round-robin over the 8 XCDs in linear order (speed only, never correctness): XCD k works through
    // the k-th eighth of the (image, row, column) tile sequence, so that tiles sharing halo rows meet in one L2
    int bt, n;
    {
        const int lin = blockIdx.x + blockIdx.y * gridDim.x, tot = gridDim.x * gridDim.y;
        const int k = lin & 7, q = tot >> 3, rem = tot & 7;
        const int b = k * q + min(k, rem) + (lin >> 3);
        n = b / (int)gridDim.x;
        bt = b - n * (int)gridDim.x;
    }
    const int bx = bt % nbx, by = bt / nbx;
    const int ox0 = bx * 8 * BX, oy0 = by * 2 * BY;
    const int ix0 = ox0 * S - G::PAD, iy0 = oy0 * S - G::PAD;
    const size_t HWi = (size_t)Hi * Wi, HWo = (size_t)Ho * Wo;

    const int r = lane & 15, g = lane >> 4, gh = g >> 1;
    const int ry = r >> 3, rx = r & 7;
    int abase[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = t / BX;
        abase[i] = ((2 * ty + ry) * S * G::HXP + (8 * tx + rx) * S) * G::VS + (g & 1) * 4;
    }
    f32x4 acc[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 breg[G::KS];

#define FN_LOAD_B(C)                                                                                \
    {                                                                                               \
        const f32x4* bsrc =                                                                         \
            reinterpret_cast<const f32x4*>(bp) + ((size_t)((C) * G::NT + nt) * G::KS) * 64 + lane;  \
        _Pragma("unroll") for (int ks = 0; ks < G::KS; ++ks) breg[ks] = bsrc[ks * 64];              \
    }
#define FN_MFMA_CHUNK()                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < G::KS; ++ks) {                                          \
        const int koff = gh ? G::tap_off(2 * ks + 1) : G::tap_off(2 * ks);                          \
        f32x4 a[G::MPW];                                                                            \
        _Pragma("unroll") for (int i = 0; i < G::MPW; ++i)                                          \
            a[i] = *reinterpret_cast<const f32x4*>(tile + abase[i] + koff);                         \
        const f32x4 bq = breg[ks];                                                                  \
        _Pragma("unroll") for (int i = 0; i < G::MPW; ++i)                                          \
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[i], 0, 0, 0);           \
        _Pragma("unroll") for (int i = 0; i < G::MPW; ++i)                                          \
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[i], 0, 0, 0);           \
        _Pragma("unroll") for (int i = 0; i < G::MPW; ++i)                                          \
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[i], 0, 0, 0);           \
        _Pragma("unroll") for (int i = 0; i < G::MPW; ++i)                                          \
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[i], 0, 0, 0);           \
    }

    if constexpr (IMG_IN) {
        // one K-chunk: channels 0..2 = R,G,B of the NCHW image, 3..7 = 0
        FN_LOAD_B(0)
        const float* img = x + (size_t)n * 3 * HWi;
#pragma unroll
        for (int i = 0; i < G::VPT; ++i) {
            const int v = tid + i * 256;
            if (v < G::NVOX) {
                const int hx = v % G::HX, hy = v / G::HX;
                const int gy = iy0 + hy, gx = ix0 + hx;
                f32x4 px = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) {
                    const size_t o = (size_t)gy * Wi + gx;
                    px.x = img[o];
                    px.y = img[HWi + o];
                    px.z = img[2 * HWi + o];
                }
                float* dst = tile + (hy * G::HXP + hx) * G::VS;
                *reinterpret_cast<f32x4*>(dst) = px;
                *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
        FN_MFMA_CHUNK()
    } else {
        int goff[G::PPT], loff[G::PPT];
        unsigned inside = 0;
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) {
            const int p = tid + i * 256;
            const int half = p & 1, v = p >> 1;
            const int hx = v % G::HX, hy = v / G::HX;
            const int gy = iy0 + hy, gx = ix0 + hx;
            const bool ok = p < G::NPIECE && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
            goff[i] = ok ? (int)(((size_t)gy * Wi + gx) * 8 + half * 4) : 0;
            inside |= ok ? (1u << i) : 0u;
            loff[i] = (p < G::NPIECE) ? (hy * G::HXP + hx) * G::VS + half * 4 : -1;
        }
        f32x4 stg[G::PPT];
#define FN_LOAD_A(C)                                                                                \
    {                                                                                               \
        const float* plane = x + ((size_t)(C) * N + n) * HWi * 8;                                   \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            stg[i] = *reinterpret_cast<const f32x4*>(plane + goff[i]);                              \
    }
#define FN_STORE_A()                                                                                \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            if (loff[i] >= 0)                                                                       \
                *reinterpret_cast<f32x4*>(tile + loff[i]) =                                         \
                    ((inside >> i) & 1u) ? stg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};                    \
    }
        FN_LOAD_B(0)
        FN_LOAD_A(0)
        FN_STORE_A()
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < G::NCH; ++c) {
            if (c + 1 < G::NCH) FN_LOAD_A(c + 1)
            FN_MFMA_CHUNK()
            if (c + 1 < G::NCH) {
                FN_LOAD_B(c + 1)
                __syncthreads();
                FN_STORE_A()
                __syncthreads();
            }
        }
#undef FN_LOAD_A
#undef FN_STORE_A
    }
#undef FN_LOAD_B
#undef FN_MFMA_CHUNK

    // epilogue: D layout col = lane&15 -> co = 16 nt + col; row m = 4 (lane>>4) + e -> pixel of tile.  bias (+ ReLU) scattered
    // into the LDS tile [oy][ox][co], then every thread owns whole pixels of one C8 plane (32 B) and stores them 16 bytes at a
    // time, contiguous across the wave (the scalar form stored 4-byte pieces in 32-byte runs)
    const int col = lane & 15, co = 16 * nt + col;
    const float bv = co < COUT ? bias[co] : 0.0f;
    __syncthreads();   // the input tile is fully consumed
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = t / BX;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 4 * (lane >> 4) + e;
            const int oy = 2 * ty + (m >> 3), ox = 8 * tx + (m & 7);
            const float v = acc[i][e] + bv;
            tile[oy * RP + ox * CP + (ox >> 3) * 16 + co] = RELU ? fmaxf(v, 0.0f) : v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
        const int u = tid + j * 256;
        const int ox = u % OX;
        int t = u / OX;
        const int oy = t % OY, pl = t / OY;
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (u < NUNIT && gy < Ho && gx < Wo) {
            const float* src = tile + oy * RP + ox * CP + (ox >> 3) * 16 + pl * 8;
            float* dst = y + (((size_t)pl * N + n) * HWo + (size_t)gy * Wo + gx) * 8;
            *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
            *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(src + 4);
        }
    }
}

template <int FMT>
__global__ __launch_bounds__(256) void fconv01_fused_kernel(
    const void* __restrict__ img_v,  // [N][3][H][W] fp32 / uint8, or [N][H][W][3] uint8
    const float* __restrict__ w0,    // [27][8] folded conv0 weights, k = (ci*3+ky)*3+kx ; then bias [8]
    const float* __restrict__ bp,    // conv1 panel [1][1][5][64][4]
    const float* __restrict__ bias,  // conv1 bias
    float* __restrict__ y,           // [1][N][H][W][8]
    int N, int H, int W) {
    using G = FConv<8, 8, 3, 1, 4, 4>;
    constexpr int IY = G::HY + 2, IX = G::HX + 2, IXP = IX + 1;  // 12 x 36 image halo, row pitch 37
    __shared__ __attribute__((aligned(16))) float tile[G::TILE_FLOATS];
    __shared__ float itile[3 * IY * IXP];
    __shared__ __attribute__((aligned(16))) float w0s[28 * 8];  // conv0 weights + bias (broadcast reads)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbx = (W + 31) / 32;
    // workgroups are dealt round-robin over the 8 XCDs in linear order (speed only, never correctness): XCD k works through
    // the k-th eighth of the (image, row, column) tile sequence, so that tiles sharing halo rows meet in one L2
    int bt, n;
    {
        const int lin = blockIdx.x + blockIdx.y * gridDim.x, tot = gridDim.x * gridDim.y;
        const int k = lin & 7, q = tot >> 3, rem = tot & 7;
        const int b = k * q + min(k, rem) + (lin >> 3);
        n = b / (int)gridDim.x;
        bt = b - n * (int)gridDim.x;
    }
    const int bx = bt % nbx, by = bt / nbx;
    const int ox0 = bx * 32, oy0 = by * 8;
    const size_t HW = (size_t)H * W;
    const float* im = static_cast<const float*>(img_v) + (size_t)n * 3 * HW;
    const unsigned char* im8 = static_cast<const unsigned char*>(img_v) + (size_t)n * 3 * HW;

    f32x4 breg[G::KS];
    {
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(bp) + lane;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) breg[ks] = bsrc[ks * 64];
    }
    if (tid < 28 * 8) w0s[tid] = w0[tid];
    // phase 1: RGB halo (origin = output origin - 2)
    for (int i = tid; i < 3 * IY * IX; i += 256) {
        const int c = i / (IY * IX), rem = i - c * (IY * IX);
        const int iy = rem / IX, ix = rem - iy * IX;
        const int gy = oy0 - 2 + iy, gx = ox0 - 2 + ix;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            if constexpr (FMT == 0) v = im[c * HW + (size_t)gy * W + gx];
            else if constexpr (FMT == 1) v = __fdiv_rn((float)im8[c * HW + (size_t)gy * W + gx], 255.0f);
            else v = __fdiv_rn((float)im8[((size_t)gy * W + gx) * 3 + c], 255.0f);
        }
        itile[(c * IY + iy) * IXP + ix] = v;
    }
    __syncthreads();
    // phase 2: conv0 + BN + ReLU on the VALU for the 10 x 34 pixels of conv1's halo; a thread owns
    // pixels tid and tid + 256 so that each (uniform, scalar-loaded) weight feeds two pixels
    {
        const int v0 = tid, v1 = (tid + 256 < G::NVOX) ? tid + 256 : tid;
        const int hy0 = v0 / G::HX, hx0 = v0 - hy0 * G::HX;
        const int hy1 = v1 / G::HX, hx1 = v1 - hy1 * G::HX;
        const float* p0 = itile + hy0 * IXP + hx0;
        const float* p1 = itile + hy1 * IXP + hx1;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 a0[4], a1[4];  // channel pairs: v_pk_fma_f32 does two FMAs per lane per issue
#pragma unroll
        for (int j = 0; j < 4; ++j) a0[j] = a1[j] = (f32x2){w0s[27 * 8 + 2 * j], w0s[27 * 8 + 2 * j + 1]};
#pragma unroll 1
        for (int c = 0; c < 3; ++c)
#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float q0 = p0[(c * IY + ky) * IXP + kx];
                    const float q1 = p1[(c * IY + ky) * IXP + kx];
                    const f32x2* wv = reinterpret_cast<const f32x2*>(w0s + ((c * 3 + ky) * 3 + kx) * 8);
                    const f32x2 q0v = (f32x2){q0, q0}, q1v = (f32x2){q1, q1};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x2 w2 = wv[j];
                        a0[j] = __builtin_elementwise_fma(q0v, w2, a0[j]);
                        a1[j] = __builtin_elementwise_fma(q1v, w2, a1[j]);
                    }
                }
        const bool in0 = (oy0 - 1 + hy0) >= 0 && (oy0 - 1 + hy0) < H && (ox0 - 1 + hx0) >= 0 && (ox0 - 1 + hx0) < W;
        const bool in1 = (oy0 - 1 + hy1) >= 0 && (oy0 - 1 + hy1) < H && (ox0 - 1 + hx1) >= 0 && (ox0 - 1 + hx1) < W;
        f32x4 lo, hi;
        lo.x = in0 ? fmaxf(a0[0][0], 0.f) : 0.f; lo.y = in0 ? fmaxf(a0[0][1], 0.f) : 0.f;
        lo.z = in0 ? fmaxf(a0[1][0], 0.f) : 0.f; lo.w = in0 ? fmaxf(a0[1][1], 0.f) : 0.f;
        hi.x = in0 ? fmaxf(a0[2][0], 0.f) : 0.f; hi.y = in0 ? fmaxf(a0[2][1], 0.f) : 0.f;
        hi.z = in0 ? fmaxf(a0[3][0], 0.f) : 0.f; hi.w = in0 ? fmaxf(a0[3][1], 0.f) : 0.f;
        float* dst = tile + (hy0 * G::HXP + hx0) * G::VS;
        *reinterpret_cast<f32x4*>(dst) = lo;
        *reinterpret_cast<f32x4*>(dst + 4) = hi;
        if (tid + 256 < G::NVOX) {
            lo.x = in1 ? fmaxf(a1[0][0], 0.f) : 0.f; lo.y = in1 ? fmaxf(a1[0][1], 0.f) : 0.f;
            lo.z = in1 ? fmaxf(a1[1][0], 0.f) : 0.f; lo.w = in1 ? fmaxf(a1[1][1], 0.f) : 0.f;
            hi.x = in1 ? fmaxf(a1[2][0], 0.f) : 0.f; hi.y = in1 ? fmaxf(a1[2][1], 0.f) : 0.f;
            hi.z = in1 ? fmaxf(a1[3][0], 0.f) : 0.f; hi.w = in1 ? fmaxf(a1[3][1], 0.f) : 0.f;
            dst = tile + (hy1 * G::HXP + hx1) * G::VS;
            *reinterpret_cast<f32x4*>(dst) = lo;
            *reinterpret_cast<f32x4*>(dst + 4) = hi;
        }
    }
    __syncthreads();
    // phase 3: conv1 on the MFMA (NT = 1, 4 M-groups of 4 M-tiles)
    const int r = lane & 15, g = lane >> 4, gh = g >> 1;
    const int ry = r >> 3, rx = r & 7;
    f32x4 acc[G::MPW];
    int abase[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = wave * G::MPW + i;
        const int tx = t % 4, ty = t / 4;
        abase[i] = ((2 * ty + ry) * G::HXP + (8 * tx + rx)) * G::VS + (g & 1) * 4;
        acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        const int koff = gh ? G::tap_off(2 * ks + 1) : G::tap_off(2 * ks);
        f32x4 a[G::MPW];
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) a[i] = *reinterpret_cast<const f32x4*>(tile + abase[i] + koff);
        const f32x4 bq = breg[ks];
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[i], 0, 0, 0);
    }
    // epilogue: bias + ReLU scattered into the LDS tile [oy 8][ox 32][8 channels] (row pitch 272 floats = 16 mod 32), then every
    // thread owns whole pixels (32 B) and stores them 16 bytes at a time, contiguous across the wave
    constexpr int ORP = 32 * 8 + 16;
    static_assert(8 * ORP <= G::TILE_FLOATS, "the staging tile re-uses the input tile");
    const int col = lane & 15;
    const float bv = col < 8 ? bias[col] : 0.0f;
    __syncthreads();   // the conv0_1 input tile is fully consumed
    if (col < 8) {
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) {
            const int t = wave * G::MPW + i;
            const int tx = t % 4, ty = t / 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * (lane >> 4) + e;
                tile[(2 * ty + (m >> 3)) * ORP + (8 * tx + (m & 7)) * 8 + col] = fmaxf(acc[i][e] + bv, 0.0f);
            }
        }
    }
    __syncthreads();
    float* yimg = y + (size_t)n * HW * 8;
    {
        const int ox = tid & 31, oy = tid >> 5;   // 256 threads = 8 rows x 32 pixels
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (gy < H && gx < W) {
            const float* src = tile + oy * ORP + ox * 8;
            float* dst = yimg + ((size_t)gy * W + gx) * 8;
            *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
            *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(src + 4);
        }
    }
}

int launch_feature_conv01(const void* imgs, int fmt, float* y, const float* blob, int N, int H, int W, hipStream_t s) {
    const FeatBlob L = feat_blob_layout();
    if ((size_t)N * H * W * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "feature_net: activation plane exceeds 31-bit offsets");
    dim3 grid(((W + 31) / 32) * ((H + 7) / 8), N);
    const float *w0 = blob + L.l0_direct_off, *bp = blob + L.panel_off[1], *b1 = blob + L.bias_off[1];
    switch (fmt) {
        case MVS_IMG_F32_CHW: fconv01_fused_kernel<0><<<grid, 256, 0, s>>>(imgs, w0, bp, b1, y, N, H, W); break;
        case MVS_IMG_U8_CHW: fconv01_fused_kernel<1><<<grid, 256, 0, s>>>(imgs, w0, bp, b1, y, N, H, W); break;
        case MVS_IMG_U8_HWC: fconv01_fused_kernel<2><<<grid, 256, 0, s>>>(imgs, w0, bp, b1, y, N, H, W); break;
        default: return fail(MVS_ERR_BAD_DTYPE, "feature_net: unknown image format %d", fmt);
    }
    return check_hip(hipGetLastError(), "fconv01_fused launch");
}

template <int CIN, int COUT, int KW, int S, int BY, int BX, bool RELU, bool IMG_IN>
static int run_fconv(const float* x, float* y, const float* bp, const float* bias, int N, int Hi, int Wi,
                     hipStream_t s) {
    using G = FConv<CIN, COUT, KW, S, BY, BX>;
    const int Ho = (Hi + 2 * G::PAD - KW) / S + 1, Wo = (Wi + 2 * G::PAD - KW) / S + 1;
    if ((size_t)N * Hi * Wi * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "feature_net: activation plane exceeds 31-bit offsets");
    dim3 grid(((Wo + 8 * BX - 1) / (8 * BX)) * ((Ho + 2 * BY - 1) / (2 * BY)), N);
    fconv_mfma_kernel<CIN, COUT, KW, S, BY, BX, RELU, IMG_IN><<<grid, 256, 0, s>>>(x, bp, bias, y, N, Hi, Wi,
                                                                                Ho, Wo);
    return check_hip(hipGetLastError(), "fconv_mfma launch");
}

// one FeatureNet layer l (0..7) at input resolution Hi x Wi
int launch_feature_layer(int l, const float* x, float* y, const float* blob, int N, int Hi, int Wi,
                         hipStream_t s) {
    const FeatBlob L = feat_blob_layout();
    const float* bp = blob + L.panel_off[l];
    const float* bias = blob + L.bias_off[l];
    switch (l) {
        case 0: return run_fconv<3, 8, 3, 1, 4, 4, true, true>(x, y, bp, bias, N, Hi, Wi, s);
        case 1: return run_fconv<8, 8, 3, 1, 4, 4, true, false>(x, y, bp, bias, N, Hi, Wi, s);
        case 2: return run_fconv<8, 16, 5, 2, 2, 4, true, false>(x, y, bp, bias, N, Hi, Wi, s);
        case 3:
        case 4: return run_fconv<16, 16, 3, 1, 2, 4, true, false>(x, y, bp, bias, N, Hi, Wi, s);
        case 5: return run_fconv<16, 32, 5, 2, 2, 2, true, false>(x, y, bp, bias, N, Hi, Wi, s);
        case 6: return run_fconv<32, 32, 3, 1, 2, 2, true, false>(x, y, bp, bias, N, Hi, Wi, s);
        case 7: return run_fconv<32, 32, 3, 1, 2, 2, false, false>(x, y, bp, bias, N, Hi, Wi, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "feature layer %d outside [0,8)", l);
    }
}

// images [N][3][H][W] -> C8-planar features [4][N][H/4][W/4][8]; bufA / bufB hold N*8*H*W floats each
int launch_feature_net_c8(const void* imgs, int fmt, const float* blob, float* feats_c8, float* bufA, float* bufB,
                          int N, int H, int W, hipStream_t s) {
    const int H2 = (H - 1) / 2 + 1, W2 = (W - 1) / 2 + 1, H4 = (H2 - 1) / 2 + 1, W4 = (W2 - 1) / 2 + 1;
    int st;
    static const bool split01 = [] {   // MVS_FEAT_SPLIT01=1: conv0 and conv1 as separate kernels
        const char* e = getenv("MVS_FEAT_SPLIT01");
        return e && e[0] == '1';
    }();
    if (split01) {
        if (fmt != MVS_IMG_F32_CHW)
            return fail(MVS_ERR_BAD_DTYPE, "MVS_FEAT_SPLIT01=1 (cross-check kernels) takes fp32 images only");
        if ((st = launch_feature_layer(0, static_cast<const float*>(imgs), bufA, blob, N, H, W, s))) return st;
        if ((st = launch_feature_layer(1, bufA, bufB, blob, N, H, W, s))) return st;
    } else if ((st = launch_feature_conv01(imgs, fmt, bufB, blob, N, H, W, s))) {
        return st;
    }
    if ((st = launch_feature_layer(2, bufB, bufA, blob, N, H, W, s))) return st;
    if ((st = launch_feature_layer(3, bufA, bufB, blob, N, H2, W2, s))) return st;
    if ((st = launch_feature_layer(4, bufB, bufA, blob, N, H2, W2, s))) return st;
    if ((st = launch_feature_layer(5, bufA, bufB, blob, N, H2, W2, s))) return st;
    if ((st = launch_feature_layer(6, bufB, bufA, blob, N, H4, W4, s))) return st;
    return launch_feature_layer(7, bufA, feats_c8, blob, N, H4, W4, s);
}

// C8-planar [C/8][N][hw][8] fp32 -> NCHW [N][C][hw] (API parity with FeatureNet.forward's output)
__global__ __launch_bounds__(256) void c8_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         int N, int C, int hw) {
    __shared__ float tile[64][9];
    const int n = blockIdx.y, pl = blockIdx.z, p0 = blockIdx.x * 64;
    const int c8 = threadIdx.x & 7, q = threadIdx.x >> 3;  // 8 channels x 32 pixels per pass
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = p0 + q + 32 * pass;
        tile[q + 32 * pass][c8] = (p < hw) ? in[(((size_t)pl * N + n) * hw + p) * 8 + c8] : 0.f;
    }
    __syncthreads();
    const int px = threadIdx.x & 63, cc = threadIdx.x >> 6;  // 64 pixels x 4 channels per pass
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int c = cc + 4 * pass;
        if (p0 + px < hw) out[((size_t)n * C + pl * 8 + c) * hw + p0 + px] = tile[px][c];
    }
}

int launch_c8_to_nchw(const float* in, float* out, int N, int C, int h, int w, hipStream_t s) {
    const int hw = h * w;
    c8_to_nchw_kernel<<<dim3((hw + 63) / 64, N, C / 8), 256, 0, s>>>(in, out, N, C, hw);
    return check_hip(hipGetLastError(), "c8_to_nchw launch");
}

// fp32 C8-planar features -> 16-bit copy for the 16-bit gather (storage dtype modes)
template <typename T>
__global__ void narrow_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (T)in[i];
}
int launch_narrow_features(const float* in, void* out, size_t n, int dtype, hipStream_t s) {
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (dtype == MVS_F16) narrow_kernel<_Float16><<<nb, 256, 0, s>>>(in, static_cast<_Float16*>(out), n);
    else if (dtype == MVS_BF16) narrow_kernel<__bf16><<<nb, 256, 0, s>>>(in, static_cast<__bf16*>(out), n);
    else return fail(MVS_ERR_BAD_DTYPE, "narrow_features: dtype %d", dtype);
    return check_hip(hipGetLastError(), "narrow_features launch");
}

// Host-side packing of one BN-folded layer: w [cout][cin][k][k] -> bp [NCH][NT][KS][64][4]
void pack_fconv_weights(const float* w, int cin, int cout, int k, float* bp) {
    const int nch = (cin + 7) / 8, nt = (cout + 15) / 16, taps = k * k, ksn = (taps + 1) / 2;
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < nt; ++t)
            for (int ks = 0; ks < ksn; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const int g = lane >> 4, n = lane & 15;
                        const int tap = 2 * ks + (g >> 1);
                        const int ci = 8 * c + 4 * (g & 1) + j4, co = 16 * t + n;
                        float v = 0.0f;
                        if (tap < taps && ci < cin && co < cout) v = w[((size_t)co * cin + ci) * taps + tap];
                        bp[((((size_t)c * nt + t) * ksn + ks) * 64 + lane) * 4 + j4] = v;
                    }
}

}  // namespace mvs
