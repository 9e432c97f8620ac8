// conv3d_mfma16.hip -- 16-bit MFMA convolutions for the fp16 / bf16 storage modes (gfx950).
//
// When the private volumes are stored in fp16 or bf16 (mvs_dtype MVS_F16 / MVS_BF16; BASELINE.json
// configs 4 / 2) the CostRegNet convolutions (reference models/mvsnet.py:35-73) run on
// v_mfma_f32_16x16x32_{f16,bf16}: operands are the stored 16-bit activations and BN-folded weights
// rounded (RNE) to the same 16-bit type, products are exact and accumulation is fp32.  That is
// 16x the fp32 MFMA rate, which turns these layers from MFMA-bound into HBM/LDS-bound.
// (MVS_MFMA16=0 keeps fp32 arithmetic on 16-bit storage: conv3d_mfma.hip.)
//
// The C8-planar layout makes the operand mapping trivial: one voxel of one plane is 8 channels =
// 16 bytes = exactly one lane's A fragment (k = 8*(lane>>4) + j  <->  tap (lane>>4), channel j), so
// a k-step of 32 is 4 taps x 8 channels and the halo tile is staged and read as raw 16-byte pieces.
//   convg16   : conv1..conv6, taps padded 27 -> 28 = 7 k-steps per 8-channel chunk
//   conv0p16  : conv0 in the Toeplitz-pair form (N = 2 x-outputs x 8 channels): the 4 x-taps of a
//               pair at fixed (kz, ky) are one k-step, 9 per chunk
//   deconvg16 : conv7/9/11, (z,y) parity classes with the x parity folded into N; 5 k-steps per chunk
#include <cstdlib>
#include <cstring>

#include "mvs_internal.h"
#include "storage.h"
#include "split_ops.h"

namespace mvs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

template <int DT>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    if (DT == MVS_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// host: fp32 -> 16-bit (RNE) for the weight panels
// ---------------------------------------------------------------------------------------------
static inline uint16_t to_bits16(float v, int dt) {
    uint16_t out;
    if (dt == MVS_F16) {
        const _Float16 h = (_Float16)v;
        std::memcpy(&out, &h, 2);
    } else {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x0040u);  // NaN
        u += 0x7FFFu + ((u >> 16) & 1u);
        out = (uint16_t)(u >> 16);
    }
    return out;
}

// =============================================================================================
// convg16: conv1..conv6
// =============================================================================================
template <int CIN, int COUT, int S, int BZ, int BY, int BX>
struct ConvG16 {
    static constexpr int NT = COUT / 16;
    static constexpr int MG = 4 / NT;
    static constexpr int MT = BZ * BY * BX;
    static constexpr int MPW = MT / MG;
    static constexpr int NCH = CIN / 8;
    static constexpr int KS = 7;
    static constexpr int VS = 8;  // 16-bit elements per voxel of a chunk plane (16 B)
    static constexpr int HZ = (BZ - 1) * S + 3, HY = (2 * BY - 1) * S + 3, HX = (8 * BX - 1) * S + 3;
    static constexpr int HXP = ((HX + 7) / 8 * 8) | 8;  // row pitch: odd multiple of 8 voxels
    static constexpr int TILE_ELEMS = HZ * HY * HXP * VS;
    static constexpr int NPIECE = HZ * HY * HX;  // 16-byte pieces (one voxel each)
    static constexpr int PPT = (NPIECE + 255) / 256;
    static_assert(NT == 1 || NT == 2 || NT == 4, "COUT must be 16, 32 or 64");
    static_assert(MT % MG == 0, "block tile must split evenly over the M-groups");
    static_assert(PPT <= 32, "piece mask is 32 bits");
    static constexpr int tap_off(int tap) {
        const int t = tap > 26 ? 26 : tap;
        return (((t / 9) * HY + (t / 3) % 3) * HXP + t % 3) * VS;
    }
};

template <int DT, int CIN, int COUT, int S, int BZ, int BY, int BX>
__global__ __launch_bounds__(256) void convg16_mfma_kernel(
    const void* __restrict__ x,               // [CIN/8][Di][Hi][Wi][8] 16-bit
    const unsigned short* __restrict__ bp,    // [NCH][NT][7][64][8] 16-bit
    const float* __restrict__ bias,           // [COUT]
    void* __restrict__ y,                     // [COUT/8][Do][Ho][Wo][8] 16-bit
    int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
    using G = ConvG16<CIN, COUT, S, BZ, BY, BX>;
    // epilogue staging tile (fp32): the block's BZ x 2BY x 8BX output voxels x COUT channels, rows padded (16 floats per 8
    // voxels, row pitch = 16 mod 32 floats) so that the accumulator scatter is conflict-free
    constexpr int OY = 2 * BY, OX = 8 * BX;
    constexpr int RP0 = OX * COUT + 16 * (OX / 8), RP = (RP0 % 32 == 16) ? RP0 : RP0 + 16;
    constexpr int OUT_FLOATS = BZ * OY * RP, NUNIT = BZ * OY * OX * (COUT / 8), UPT = (NUNIT + 255) / 256;
    constexpr int LDS_BYTES = G::TILE_ELEMS * 2 > OUT_FLOATS * 4 ? G::TILE_ELEMS * 2 : OUT_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    unsigned short* tile = reinterpret_cast<unsigned short*>(lds_raw);
    float* otile = reinterpret_cast<float*>(lds_raw);
    const unsigned short* xs = static_cast<const unsigned short*>(x);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NT, mg = wave / G::NT;
    const int nbx = (Wo + 8 * BX - 1) / (8 * BX), nby = (Ho + 2 * BY - 1) / (2 * BY);
    int b;   // XCD-aware tile order (see convgs_mfma_kernel)
    {
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ox0 = bx * 8 * BX, oy0 = by * 2 * BY, oz0 = bz * BZ;
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;
    const size_t Vin = (size_t)Di * Hi * Wi, Vout = (size_t)Do * Ho * Wo;

    int goff[G::PPT], loff[G::PPT];
    unsigned inside = 0;
#pragma unroll
    for (int i = 0; i < G::PPT; ++i) {
        const int v = tid + i * 256;
        const int hx = v % G::HX, t = v / G::HX;
        const int hy = t % G::HY, hz = t / G::HY;
        const int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = v < G::NPIECE && gz >= 0 && gz < Di && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
        goff[i] = ok ? (int)((((size_t)gz * Hi + gy) * Wi + gx) * 8) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = (v < G::NPIECE) ? ((hz * G::HY + hy) * G::HXP + hx) * G::VS : -1;
    }

    // A fragment: lane (r = lane&15 -> voxel (ry, rx) of the M-tile, g = lane>>4 -> tap 4ks+g)
    const int r = lane & 15, g = lane >> 4;
    const int ry = r >> 3, rx = r & 7;
    int abase[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
        abase[i] = (((tz * S) * G::HY + (2 * ty + ry) * S) * G::HXP + (8 * tx + rx) * S) * G::VS;
    }
    int koff[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
        koff[ks] = g == 0 ? G::tap_off(4 * ks) : g == 1 ? G::tap_off(4 * ks + 1)
                 : g == 2 ? G::tap_off(4 * ks + 2) : G::tap_off(4 * ks + 3);

    f32x4 acc[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 breg[G::KS];
    u32x4 stg[G::PPT];

#define MVS_LOAD_B(C)                                                                               \
    {                                                                                               \
        const u32x4* bsrc =                                                                         \
            reinterpret_cast<const u32x4*>(bp) + ((size_t)((C) * G::NT + nt) * G::KS) * 64 + lane;  \
        _Pragma("unroll") for (int ks = 0; ks < G::KS; ++ks) breg[ks] = bsrc[ks * 64];              \
    }
#define MVS_LOAD_A(C)                                                                               \
    {                                                                                               \
        const unsigned short* plane = xs + (size_t)(C) * Vin * 8;                                   \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            stg[i] = *reinterpret_cast<const u32x4*>(plane + goff[i]);                              \
    }
#define MVS_STORE_A()                                                                               \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            if (loff[i] >= 0)                                                                       \
                *reinterpret_cast<u32x4*>(tile + loff[i]) =                                         \
                    ((inside >> i) & 1u) ? stg[i] : (u32x4){0u, 0u, 0u, 0u};                        \
    }

    MVS_LOAD_B(0)
    MVS_LOAD_A(0)
    MVS_STORE_A()
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < G::NCH; ++c) {
        if (c + 1 < G::NCH) MVS_LOAD_A(c + 1)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
#pragma unroll
            for (int i = 0; i < G::MPW; ++i) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(tile + abase[i] + koff[ks]);
                acc[i] = mfma16<DT>(a, breg[ks], acc[i]);
            }
        }
        if (c + 1 < G::NCH) {
            MVS_LOAD_B(c + 1)
            __syncthreads();
            MVS_STORE_A()
            __syncthreads();
        }
    }
#undef MVS_LOAD_B
#undef MVS_LOAD_A
#undef MVS_STORE_A

    // epilogue: ReLU(acc + bias) scattered into the LDS tile [oz][oy][ox][co], then every thread owns whole voxels of one C8
    // plane (8 channels) and stores them 16 bytes at a time, contiguous across the wave (the scalar form stored single
    // elements: 2- or 4-byte pieces in 16- or 32-byte runs)
    const int n = lane & 15, co = 16 * nt + n;
    const float bv = bias[co];
    __syncthreads();   // the input tile is fully consumed
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 4 * (lane >> 4) + e;
            const int oy = 2 * ty + (m >> 3), ox = 8 * tx + (m & 7);
            otile[(tz * OY + oy) * RP + ox * COUT + (ox >> 3) * 16 + co] = fmaxf(acc[i][e] + bv, 0.0f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
        const int u = tid + j * 256;
        const int ox = u % OX;
        int t = u / OX;
        const int oy = t % OY; t /= OY;
        const int ozl = t % BZ, pl = t / BZ;
        const int gz = oz0 + ozl, gy = oy0 + oy, gx = ox0 + ox;
        if (u < NUNIT && gz < Do && gy < Ho && gx < Wo) {
            const float* src = otile + (ozl * OY + oy) * RP + ox * COUT + (ox >> 3) * 16 + pl * 8;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
            const size_t uo = ((size_t)pl * Vout + ((size_t)gz * Ho + gy) * Wo + gx) * 8;
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            store8_16<DT>(y, uo, v);
        }
    }
}

template <int DT, int CIN, int COUT, int S, int BZ, int BY, int BX>
static int run_convg16(const void* x, void* y, const unsigned short* bp, const float* bias, int Di, int Hi,
                       int Wi, hipStream_t s) {
    const int Do = (Di - 1) / S + 1, Ho = (Hi - 1) / S + 1, Wo = (Wi - 1) / S + 1;
    if ((size_t)Di * Hi * Wi * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "convg16_mfma: plane exceeds 31-bit offsets");
    const int nb = ((Wo + 8 * BX - 1) / (8 * BX)) * ((Ho + 2 * BY - 1) / (2 * BY)) * ((Do + BZ - 1) / BZ);
    convg16_mfma_kernel<DT, CIN, COUT, S, BZ, BY, BX><<<nb, 256, 0, s>>>(x, bp, bias, y, Di, Hi, Wi, Do,
                                                                         Ho, Wo);
    return check_hip(hipGetLastError(), "convg16_mfma launch");
}

// does a grid of BZ x BY x BX M-tiles (2 x 8 outputs each) over the OUTPUT of a stride-S layer (S = 1 with the input
// dims for the transposed layers) still give every CU a block and a half?
static bool tile_fills_chip(int Di, int Hi, int Wi, int S, int BZ, int BY, int BX) {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n;
    }();
    static const int force = [] {   // MVS_DEEP_TILES=1 / 0: always / never (tests, A/B runs)
        const char* e = getenv("MVS_DEEP_TILES");
        return e ? atoi(e) : -1;
    }();
    if (force >= 0) return force != 0;
    const int Do = (Di - 1) / S + 1, Ho = (Hi - 1) / S + 1, Wo = (Wi - 1) / S + 1;
    const long nb = (long)((Wo + 8 * BX - 1) / (8 * BX)) * ((Ho + 2 * BY - 1) / (2 * BY)) * ((Do + BZ - 1) / BZ);
    return 2 * nb >= 3 * (long)cus;
}

template <int DT>
static int launch_convg16_dt(int layer, const void* x, void* y, const unsigned short* bp, const float* bias,
                             int Di, int Hi, int Wi, hipStream_t s) {
    switch (layer) {
        case 1: return run_convg16<DT, 8, 16, 2, 2, 2, 2>(x, y, bp, bias, Di, Hi, Wi, s);
        case 2: return run_convg16<DT, 16, 16, 1, 2, 4, 2>(x, y, bp, bias, Di, Hi, Wi, s);
        case 3: return run_convg16<DT, 16, 32, 2, 4, 1, 1>(x, y, bp, bias, Di, Hi, Wi, s);   // cfg5 sweep: 1x2x2 0.0115, 4x1x1 0.0107 ms
        // conv4 .. conv6: z-deep block tiles (halo planes re-used) wherever they still give every CU a block and a half --
        // round-4 sweep, cfg3 bf16 (1600x1184x256): conv4 1x2x2 -> 4x2x1 0.0574 -> 0.0456 ms, conv5 1x1x1 -> 2x1x1 0.0296 -> 0.0236,
        // conv6 1x1x1 -> 4x1x1 0.0479 -> 0.0243; cfg5 fp16 (640x512x192): conv4 0.0133 -> 0.0118, conv5 / conv6 keep 1x1x1
        // (2x1x1 / 4x1x1 would leave 240 / 120 blocks for 256 CUs: 0.0104 / 0.0145 against 0.0106 / 0.0134)
        case 4:
            if (tile_fills_chip(Di, Hi, Wi, 1, 4, 2, 1)) return run_convg16<DT, 32, 32, 1, 4, 2, 1>(x, y, bp, bias, Di, Hi, Wi, s);
            return run_convg16<DT, 32, 32, 1, 1, 2, 2>(x, y, bp, bias, Di, Hi, Wi, s);
        case 5:
            if (tile_fills_chip(Di, Hi, Wi, 2, 2, 1, 1)) return run_convg16<DT, 32, 64, 2, 2, 1, 1>(x, y, bp, bias, Di, Hi, Wi, s);
            return run_convg16<DT, 32, 64, 2, 1, 1, 1>(x, y, bp, bias, Di, Hi, Wi, s);
        case 6:
            if (tile_fills_chip(Di, Hi, Wi, 1, 4, 1, 1)) return run_convg16<DT, 64, 64, 1, 4, 1, 1>(x, y, bp, bias, Di, Hi, Wi, s);
            return run_convg16<DT, 64, 64, 1, 1, 1, 1>(x, y, bp, bias, Di, Hi, Wi, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "convg16_mfma: layer %d not covered", layer);
    }
}

// wfold [27][cin][cout] -> bp [cin/8][cout/16][7][64][8] (16-bit)
void pack_convg16_weights(const float* wfold, int cin, int cout, int dt, unsigned short* bp) {
    const int nch = cin / 8, nt = cout / 16;
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < nt; ++t)
            for (int ks = 0; ks < 7; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int g = lane >> 4, n = lane & 15;
                        const int tap = 4 * ks + g, ci = 8 * c + j, co = 16 * t + n;
                        const float v = tap < 27 ? wfold[((size_t)tap * cin + ci) * cout + co] : 0.0f;
                        bp[((((size_t)c * nt + t) * 7 + ks) * 64 + lane) * 8 + j] = to_bits16(v, dt);
                    }
}

// =============================================================================================
// conv0p16: conv0 (32 -> 8) in the Toeplitz-pair form
// =============================================================================================
namespace c16 {
constexpr int TZ = 2, TY = 8, TX = 32;
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int VS = 8;
constexpr int TILE_ELEMS = HZ * HY * HX * VS;
constexpr int NPIECE = HZ * HY * HX;             // 1360
constexpr int PPT = (NPIECE + 255) / 256;        // 6
constexpr int KS = 9;                            // (kz, ky) pairs; the 4 x-taps of a pair = one k-step
}  // namespace c16

template <int DT>
__global__ __launch_bounds__(256) void conv0p16_mfma_kernel(
    const void* __restrict__ x,             // [4][D][H][W][8] 16-bit
    const unsigned short* __restrict__ bp,  // [4 chunks][9][64][8] 16-bit Toeplitz panel
    const float* __restrict__ bias,         // [8]
    void* __restrict__ y,                   // [D][H][W][8] 16-bit
    int D, int H, int W) {
    using namespace c16;
    __shared__ __attribute__((aligned(16))) unsigned short tile[TILE_ELEMS];
    const unsigned short* xs = static_cast<const unsigned short*>(x);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    const size_t V = (size_t)D * H * W;

    int goff[PPT], loff[PPT];
    unsigned inside = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int v = tid + i * 256;
        const int hx = v % HX, t = v / HX;
        const int hy = t % HY, hz = t / HY;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = v < NPIECE && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? (int)((((size_t)gz * H + gy) * W + gx) * 8) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = (v < NPIECE) ? v * VS : -1;
    }

    // lane (r = pair index, g = kx' in 0..3): halo x = 2r + g
    const int r = lane & 15, g = lane >> 4;
    const int zt = wave >> 1, yt0 = 4 * (wave & 1);
    const unsigned short* abase = tile + ((zt * HY + yt0) * HX + 2 * r + g) * VS;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 breg[KS];
    u32x4 stg[PPT];

#define MVS_LOAD_B(C)                                                                         \
    {                                                                                         \
        const u32x4* bsrc = reinterpret_cast<const u32x4*>(bp) + (size_t)(C) * KS * 64 + lane; \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) breg[ks] = bsrc[ks * 64];           \
    }
#define MVS_LOAD_A(C)                                                                         \
    {                                                                                         \
        const unsigned short* plane = xs + (size_t)(C) * V * 8;                               \
        _Pragma("unroll") for (int i = 0; i < PPT; ++i)                                       \
            stg[i] = *reinterpret_cast<const u32x4*>(plane + goff[i]);                        \
    }
#define MVS_STORE_A()                                                                         \
    {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < PPT; ++i)                                       \
            if (loff[i] >= 0)                                                                 \
                *reinterpret_cast<u32x4*>(tile + loff[i]) =                                   \
                    ((inside >> i) & 1u) ? stg[i] : (u32x4){0u, 0u, 0u, 0u};                  \
    }

    MVS_LOAD_B(0)
    MVS_LOAD_A(0)
    MVS_STORE_A()
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (c < 3) MVS_LOAD_A(c + 1)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int koff = (((ks / 3) * HY + ks % 3) * HX) * VS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(abase + koff + i * HX * VS);
                acc[i] = mfma16<DT>(a, breg[ks], acc[i]);
            }
        }
        if (c < 3) {
            MVS_LOAD_B(c + 1)
            __syncthreads();
            MVS_STORE_A()
            __syncthreads();
        }
    }
#undef MVS_LOAD_B
#undef MVS_LOAD_A
#undef MVS_STORE_A

    const int n = lane & 15, jj = n >> 3, co = n & 7;
    const float bv = bias[co];
    const int gz = z0 + zt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gy = y0 + yt0 + i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 4 * (lane >> 4) + e;
            const int gx = x0 + 2 * m + jj;
            if (gz < D && gy < H && gx < W)
                St<DT>::store1(y, (((size_t)gz * H + gy) * W + gx) * 8 + co, fmaxf(acc[i][e] + bv, 0.0f));
        }
    }
}

// =============================================================================================
// conv0z16: conv0 (32 -> 8) for 16-bit storage as a z-MARCHING kernel (default; conv0p16 above is the small-shape
// form).  In the 16-bit modes conv0 is not matrix-bound (the 16-bit MFMA is 16x the fp32 rate) but bound by the
// bytes it moves: conv0p16 re-stages a 4 x 10 x 34 halo per 2 x 8 x 32 outputs (2.66x the input through L2 ->
// LDS, 8 block barriers per tile, 2-byte scattered stores) and ran at 29 % of HBM at cfg3 (1.05 ms; this kernel:
// 0.58 ms = 52 %; cfg5 0.131 -> 0.077 ms).  Here a block owns an
// 8 x 32 (y, x) column of the volume and marches along z:
//   * a ring of four input planes (all 32 channels, 10 x 34 halo) lives in LDS; every input plane is fetched
//     ONCE per column (halo only in y / x: 1.33x), requested three steps ahead into registers and written into the
//     slot of the plane that has just left the stencil -- one barrier per output plane;
//   * 512 threads: wave w owns output row w; the whole 36-k-step Toeplitz panel (4 chunks x 9 (kz, ky) steps,
//     144 VGPRs) stays in registers, a step is 36 ds_read_b128 + 36 MFMAs per wave;
//   * epilogue: ReLU(acc + bias) is packed to 16 bit, transposed through a wave-private LDS strip (one wave's
//     LDS operations execute in order) and leaves as 16-byte stores, one voxel per lane.
// Same panel and same k order as conv0p16 (chunk-major, then kz, ky): identical results.
// =============================================================================================
// C0Z_TY = rows (= waves) per block: 8 -> one block per CU (default); 4 -> two independent 4-wave blocks per CU
// (-DC0Z_TY=4) measured slower: 0.632 vs 0.590 ms at cfg3, 0.0876 vs 0.0818 at cfg5 (halo 1.5x instead of 1.25x in y)
#ifndef C0Z_TY
#define C0Z_TY 8
#endif
#ifndef C0Z_CACHE_C
#define C0Z_CACHE_C 4
#endif
namespace c0z {
constexpr int TY = C0Z_TY, TX = 32, HY = TY + 2, HX = TX + 2;
constexpr int CH = HY * HX * 8;                  // 16-bit elements of one chunk of one plane
constexpr int SLOT = 4 * CH;                     // one plane: [chunk][hy][hx][8]
constexpr int RING = 4;
constexpr int THREADS = 64 * TY;                 // 512
constexpr int NPIECE = 4 * HY * HX;              // 16-byte pieces of one plane (1360)
constexpr int PPT = (NPIECE + THREADS - 1) / THREADS;   // 3
constexpr int OUTW = TX * 8;                     // 16-bit elements of one output row
}  // namespace c0z

template <int DT>
__global__ __launch_bounds__(c0z::THREADS) void conv0z16_mfma_kernel(
    const void* __restrict__ x,             // [4][D][H][W][8] 16-bit
    const unsigned short* __restrict__ bp,  // [4 chunks][9][64][8] 16-bit Toeplitz panel
    const float* __restrict__ bias,         // [8]
    void* __restrict__ y,                   // [D][H][W][8] 16-bit
    int D, int H, int W, int ZC, int nbx, int nby) {
    using namespace c0z;
    __shared__ __attribute__((aligned(16))) unsigned short ring[RING * SLOT];
    __shared__ __attribute__((aligned(16))) unsigned short outt[TY * OUTW];
    const unsigned short* xs = static_cast<const unsigned short*>(x);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Blocks are dealt round-robin over the 8 XCDs (blockIdx.x % 8 names the XCD: speed only, never correctness)
    // and every XCD has its own L2: XCD k works through the k-th eighth of the (z chunk, row, column) sequence, so
    // that columns sharing a y / x halo run on the same L2 at about the same time
    int b;
    {
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;   // XCD k runs q (+1 if k < rem) blocks
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int x0 = bx * TX, y0 = by * TY;
    const int za = bz * ZC, zb = min(za + ZC, D);
    const size_t HW8 = (size_t)H * W * 8, V8 = (size_t)D * HW8;

    // staging: piece p = tid + i * THREADS -> (chunk, hy, hx)
    size_t goff[PPT];
    int loff[PPT];
    unsigned okxy = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = tid + i * THREADS;
        const int hx = p % HX, t = p / HX;
        const int hy = t % HY, c = t / HY;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = p < NPIECE && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? (size_t)c * V8 + ((size_t)gy * W + gx) * 8 : 0;
        okxy |= ok ? (1u << i) : 0u;
        loff[i] = p < NPIECE ? p * 8 : -1;
    }
    auto load_plane = [&](int gz, u32x4 (&st)[PPT]) {   // raw loads (plane clamped into the volume); masked when stored
        const size_t zo = (size_t)min(max(gz, 0), D - 1) * HW8;
#pragma unroll
        for (int i = 0; i < PPT; ++i) st[i] = *reinterpret_cast<const u32x4*>(xs + goff[i] + zo);
    };
    auto store_plane = [&](int gz, const u32x4 (&st)[PPT]) {   // plane gz -> slot (gz + 1) & 3, zeros outside the volume
        unsigned short* slot = ring + ((gz + 1) & (RING - 1)) * SLOT;
        const bool zok = gz >= 0 && gz < D;
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            if (loff[i] >= 0)
                *reinterpret_cast<u32x4*>(slot + loff[i]) = (zok && ((okxy >> i) & 1u)) ? st[i] : (u32x4){0u, 0u, 0u, 0u};
    };

    // the whole panel in registers: breg[c][kz * 3 + ky]
    u32x4 breg[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) breg[c][ks] = reinterpret_cast<const u32x4*>(bp)[(c * 9 + ks) * 64 + lane];

    // prologue: planes za-1, za, za+1 into the ring; planes za+2 and za+3 in flight
    u32x4 sa[PPT], sb[PPT], sc[PPT];
    load_plane(za - 1, sa);
    load_plane(za, sb);
    load_plane(za + 1, sc);
    store_plane(za - 1, sa);
    store_plane(za, sb);
    store_plane(za + 1, sc);
    load_plane(za + 2, sa);   // `sa` holds the plane that step za writes, `sb` the one step za + 1 writes
    load_plane(za + 3, sb);
    __syncthreads();

    // lane (r = pair index, g = kx' in 0..3): halo x = 2r + g; wave = output row
    const int r = lane & 15, g = lane >> 4;
    const int aoff = (wave * HX + 2 * r + g) * 8;
    const int n = lane & 15, jj = n >> 3, co = n & 7;
    const float bv = bias[co];
    unsigned short* orow = outt + wave * OUTW;
    const int gy = y0 + wave;
    // epilogue store role: lanes 0..31 own voxel x0 + lane of the wave's row
    const bool st_ok = lane < TX && gy < H && x0 + lane < W;
    const unsigned st_off = (unsigned)(((size_t)gy * W + x0 + (lane & 31)) * 16);   // bytes; the launcher checks < 4 GiB
    const unsigned zstep_b = (unsigned)(HW8 * 2);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, (int)(unsigned)(V8 * 2), 0x00020000);

    // one output plane: `nxt` receives plane z + 4 (first read in step z + 3), `cur` holds plane z + 2 (requested two
    // steps ago) and is written into the ring at the end: three planes (65 KB per CU) are in flight, the three
    // register sets rotate (the loop is unrolled by three: a register copy would wait for the loads just issued).
    // The kernel moves 26 KB per step and CU, so what it needs is bytes in flight: at cfg3 two sets (and a branch-free
    // VMEM stream, below) gave 0.655-0.685 ms, three 0.579, four 0.577
    // The kernel is bound by its LDS reads (round-4 counters at cfg3: LDS busy 64 % of the kernel's cycles, 36 ds_read_b128 for
    // 36 MFMAs per wave and step).  Plane z + 1's fragments, read as kz = 2 of step z, are the kz = 1 fragments of step
    // z + 1: those of the first CACHE_C chunks stay in registers (in place: kz = 1 has consumed a register before kz = 2
    // refills it): 24 instead of 36 reads per step at CACHE_C = 4 (48 VGPRs, 250 in all, no spills).  Measured, cfg3 / cfg5:
    // CACHE_C 0: 0.600-0.609 / 0.080-0.081 ms, 2: 0.592-0.595 / 0.079, 3: 0.590-0.594 / 0.078-0.080, 4: 0.588-0.589 / 0.078-0.079
    // -- the reads were not what bounds it (2.4 GB in 0.59 ms = 4.1 TB/s).
    constexpr int CACHE_C = C0Z_CACHE_C;
    u32x4 fcache[CACHE_C > 0 ? CACHE_C : 1][3];
    if (CACHE_C > 0) {   // plane za (the first step's kz = 1)
        const unsigned short* pl = ring + ((za + 1) & (RING - 1)) * SLOT + aoff;
#pragma unroll
        for (int c = 0; c < CACHE_C; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) fcache[c][ky] = *reinterpret_cast<const u32x4*>(pl + c * CH + ky * HX * 8);
    }
    auto step = [&](int z, u32x4 (&cur)[PPT], u32x4 (&nxt)[PPT]) {
        load_plane(z + 4, nxt);   // unconditional (the plane index is clamped): hipcc then keeps counted vmcnt waits
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
            const unsigned short* pl = ring + ((z + kz) & (RING - 1)) * SLOT + aoff;   // plane z - 1 + kz
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    u32x4 a;
                    if (kz == 1 && c < CACHE_C) {
                        a = fcache[c][ky];
                    } else {
                        a = *reinterpret_cast<const u32x4*>(pl + c * CH + ky * HX * 8);
                        if (kz == 2 && c < CACHE_C) fcache[c][ky] = a;
                    }
                    acc = mfma16<DT>(a, breg[c][kz * 3 + ky], acc);
                }
        }
        // epilogue: element e of acc = pair m = 4 g + e -> voxel x = 2 m + jj, channel co
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = fmaxf(acc[e] + bv, 0.0f);
            unsigned short bits;
            if (DT == MVS_F16) { const _Float16 hv = (_Float16)v; bits = __builtin_bit_cast(unsigned short, hv); }
            else { const __bf16 hv = (__bf16)v; bits = __builtin_bit_cast(unsigned short, hv); }
            orow[(2 * (4 * g + e) + jj) * 8 + co] = bits;
        }
        __builtin_amdgcn_wave_barrier();
        // raw buffer store: lanes without a voxel carry an offset beyond the descriptor's range (dropped by the
        // hardware) -- an exec-masked store could be branched around, and every branch around a VMEM instruction
        // makes the compiler fall back to s_waitcnt vmcnt(0), i.e. to waiting for the loads just issued
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *reinterpret_cast<const u16x8*>(orow + (lane & 31) * 8)),
                                               yrs, (int)((st_ok && z < zb) ? st_off + (unsigned)z * zstep_b : 0xFFFFFFF0u), 0, 0);
        __builtin_amdgcn_wave_barrier();
        // plane z + 2 replaces plane z - 2, which no step reads any more
        store_plane(z + 2, cur);   // (a plane beyond zb lands in a slot nobody reads any more)
        __syncthreads();
    };
#pragma unroll 1
    for (int z = za; z < zb; z += 3) {   // up to two surplus steps at the end of a chunk: their stores are dropped
        step(z, sa, sc);
        step(z + 1, sb, sa);
        step(z + 2, sc, sb);
    }
}

// =============================================================================================
// convz16: conv1 8->16 s2, conv2 16->16 and conv3 16->32 s2 (models/mvsnet.py:38-41) for 16-bit storage as
// z-MARCHING kernels, the scheme of conv0z16 above with the standard operand mapping of convg16 (k = 4 taps x 8
// channels per MFMA, 7 k-steps per 8-channel chunk, same panel): a 512-thread block owns an 8 x TX column of the
// OUTPUT, a ring of input planes (all chunks) lives in LDS -- four planes for stride 1, five for stride 2, where two
// new planes enter per output plane --, requested three steps ahead into rotating register sets, the whole panel in
// registers, one block barrier per output plane, branch-free VMEM, 16-byte stores through a wave-private strip.
// The tile kernels re-staged a 3-plane halo per one or two output planes and wrote 2-byte pieces.
// =============================================================================================
template <int CIN, int COUT, int S>
struct ConvZ16 {
    static constexpr int TY = 8, TX = (S == 1) ? 32 : 16;      // output column
    static constexpr int NCH = CIN / 8, NT = COUT / 16;
    static constexpr int MPW = (TY / 2) * (TX / 8) / 8;          // M-tiles (2 x 8 outputs) per wave: 2 / 1
    static constexpr int HY = (TY - 1) * S + 3, HX = (TX - 1) * S + 3;
    static constexpr int HXP = ((HX + 7) / 8 * 8) | 8;           // row pitch: odd multiple of 8 voxels (convg16's rule)
    static constexpr int CHS = HY * HXP * 8;                     // 16-bit elements of one chunk of one plane
    static constexpr int SLOT = NCH * CHS;
    static constexpr int RING = (S == 1) ? 4 : 5;
    static constexpr int THREADS = 512;
    static constexpr int NPIECE = NCH * HY * HX;                 // 16-byte pieces of one plane
    static constexpr int PPT = (NPIECE + THREADS - 1) / THREADS;
    static constexpr int STRIP = 16 * 16 * MPW;                  // 16-bit elements: [M-tile][voxel 16][channel 16]
    static_assert(MPW == 1 || MPW == 2, "one or two M-tiles per wave");
    static_assert((RING * SLOT + 8 * STRIP * NT) * 2 <= 160 * 1024, "ring exceeds the CU's LDS");
};

template <int DT, int CIN, int COUT, int S>
__global__ __launch_bounds__(512) void convz16_mfma_kernel(
    const void* __restrict__ x,             // [CIN/8][Di][Hi][Wi][8] 16-bit
    const unsigned short* __restrict__ bp,  // [NCH][NT][7][64][8] 16-bit (pack_convg16_weights)
    const float* __restrict__ bias,         // [COUT]
    void* __restrict__ y,                   // [COUT/8][Do][Ho][Wo][8] 16-bit
    int Di, int Hi, int Wi, int Do, int Ho, int Wo, int ZC, int nbx, int nby) {
    using G = ConvZ16<CIN, COUT, S>;
    __shared__ __attribute__((aligned(16))) unsigned short ring[G::RING * G::SLOT];
    __shared__ __attribute__((aligned(16))) unsigned short strips[8 * G::STRIP * G::NT];
    const unsigned short* xs = static_cast<const unsigned short*>(x);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b;
    {   // XCD k works through the k-th eighth of the (z chunk, row, column) sequence
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ox0 = bx * G::TX, oy0 = by * G::TY;
    const int za = bz * ZC, zb = min(za + ZC, Do);
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1;
    const size_t HW8 = (size_t)Hi * Wi * 8, Vin8 = (size_t)Di * HW8;

    // staging: piece p = tid + i * THREADS -> (chunk, hy, hx)
    size_t goff[G::PPT];
    int loff[G::PPT];
    unsigned okxy = 0;
#pragma unroll
    for (int i = 0; i < G::PPT; ++i) {
        const int p = tid + i * G::THREADS;
        const int hx = p % G::HX, t = p / G::HX;
        const int hy = t % G::HY, c = t / G::HY;
        const int gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = p < G::NPIECE && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
        goff[i] = ok ? (size_t)c * Vin8 + ((size_t)gy * Wi + gx) * 8 : 0;
        okxy |= ok ? (1u << i) : 0u;
        loff[i] = p < G::NPIECE ? c * G::CHS + (hy * G::HXP + hx) * 8 : -1;
    }
    auto load_plane = [&](int gz, u32x4 (&st)[G::PPT]) {   // raw loads (plane clamped into the volume); masked when stored
        const size_t zo = (size_t)min(max(gz, 0), Di - 1) * HW8;
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) st[i] = *reinterpret_cast<const u32x4*>(xs + goff[i] + zo);
    };
    auto slot_of = [&](int gz) { return ((gz + 1) % G::RING) * G::SLOT; };   // gz >= -1
    auto store_plane = [&](int gz, const u32x4 (&st)[G::PPT]) {
        unsigned short* slot = ring + slot_of(gz);
        const bool zok = gz >= 0 && gz < Di;
#pragma unroll
        for (int i = 0; i < G::PPT; ++i)
            if (loff[i] >= 0)
                *reinterpret_cast<u32x4*>(slot + loff[i]) = (zok && ((okxy >> i) & 1u)) ? st[i] : (u32x4){0u, 0u, 0u, 0u};
    };

    // the whole panel in registers: breg[c][nt][ks]
    u32x4 breg[G::NCH][G::NT][7];
#pragma unroll
    for (int c = 0; c < G::NCH; ++c)
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < 7; ++ks)
                breg[c][nt][ks] = reinterpret_cast<const u32x4*>(bp)[((c * G::NT + nt) * 7 + ks) * 64 + lane];

    // A fragment: lane (r -> voxel (ry, rx) of the M-tile, g -> tap 4 ks + g)
    const int r = lane & 15, g = lane >> 4;
    int mt_off[G::MPW];   // M-tile origin inside a chunk plane; wave -> M-tiles MPW * wave ..
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = G::MPW * wave + i;
        const int tx = t % (G::TX / 8), ty = t / (G::TX / 8);
        mt_off[i] = ((S * (2 * ty + (r >> 3))) * G::HXP + S * (8 * tx + (r & 7))) * 8;
    }
    int kin[7], kzl[7];   // this lane's tap of k-step ks: offset inside a plane, plane kz
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        const int tap = min(4 * ks + g, 26);   // tap 27 is padding (zero weights): any valid address
        kzl[ks] = tap / 9;
        kin[ks] = (((tap / 3) % 3) * G::HXP + tap % 3) * 8;
    }
    const int n = lane & 15;
    float bv[G::NT];
#pragma unroll
    for (int nt = 0; nt < G::NT; ++nt) bv[nt] = bias[16 * nt + n];
    unsigned short* strip = strips + wave * G::STRIP * G::NT;
    // epilogue store role: lane -> (M-tile sm, voxel sv, 8-channel half sh) of each N-tile
    const int sm = lane >> 5, sv = (lane >> 1) & 15, sh = lane & 1;
    const bool s_have = sm < G::MPW;
    const int st_t = G::MPW * wave + (s_have ? sm : 0);
    const int sgy = oy0 + 2 * (st_t / (G::TX / 8)) + (sv >> 3), sgx = ox0 + 8 * (st_t % (G::TX / 8)) + (sv & 7);
    const bool st_ok = s_have && sgy < Ho && sgx < Wo;
    const size_t Vout8 = (size_t)Do * Ho * Wo * 8;
    const unsigned st_off = (unsigned)(((size_t)sh * Vout8 + ((size_t)sgy * Wo + sgx) * 8) * 2);   // + 2 nt planes
    const unsigned zstep_b = (unsigned)((size_t)Ho * Wo * 8 * 2), nt_b = (unsigned)(2 * Vout8 * 2);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, (int)(unsigned)((COUT / 8) * Vout8 * 2), 0x00020000);

    // prologue: input planes S za - 1 .. S za + 1 into the ring; the planes of steps za and za + 1 in flight
    u32x4 w[3][S][G::PPT];   // [register set][plane of the step][piece]
    {
        u32x4 t0[G::PPT], t1[G::PPT], t2[G::PPT];
        load_plane(S * za - 1, t0);
        load_plane(S * za, t1);
        load_plane(S * za + 1, t2);
#pragma unroll
        for (int q = 0; q < S; ++q) load_plane(S * za + 2 + q, w[0][q]);
#pragma unroll
        for (int q = 0; q < S; ++q) load_plane(S * (za + 1) + 2 + q, w[1][q]);
        store_plane(S * za - 1, t0);
        store_plane(S * za, t1);
        store_plane(S * za + 1, t2);
    }
    __syncthreads();

    // one output plane zo: set `nx` receives the planes step zo + 2 will write, set `cur` (requested two steps ago) goes
    // into the ring at the end, into the slots of planes no step reads any more
    auto step = [&](int zo, u32x4 (&cur)[S][G::PPT], u32x4 (&nx)[S][G::PPT]) {
#pragma unroll
        for (int q = 0; q < S; ++q) load_plane(S * (zo + 2) + 2 + q, nx[q]);
        int sb[3];
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) sb[kz] = slot_of(S * zo - 1 + kz);
        f32x4 acc[G::MPW][G::NT];
#pragma unroll
        for (int i = 0; i < G::MPW; ++i)
#pragma unroll
            for (int nt = 0; nt < G::NT; ++nt) acc[i][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const int sl = (kzl[ks] == 0 ? sb[0] : kzl[ks] == 1 ? sb[1] : sb[2]) + kin[ks];
#pragma unroll
            for (int c = 0; c < G::NCH; ++c)
#pragma unroll
                for (int i = 0; i < G::MPW; ++i) {
                    const u32x4 a = *reinterpret_cast<const u32x4*>(ring + sl + c * G::CHS + mt_off[i]);
#pragma unroll
                    for (int nt = 0; nt < G::NT; ++nt) acc[i][nt] = mfma16<DT>(a, breg[c][nt][ks], acc[i][nt]);
                }
        }
        // epilogue: element e of acc[i][nt] = voxel m = 4 g + e of M-tile i, channel 16 nt + n -> strip[nt][i][m][n]
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt)
#pragma unroll
            for (int i = 0; i < G::MPW; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaxf(acc[i][nt][e] + bv[nt], 0.0f);
                    unsigned short bits;
                    if (DT == MVS_F16) { const _Float16 hv = (_Float16)v; bits = __builtin_bit_cast(unsigned short, hv); }
                    else { const __bf16 hv = (__bf16)v; bits = __builtin_bit_cast(unsigned short, hv); }
                    strip[nt * G::STRIP + (i * 16 + 4 * g + e) * 16 + n] = bits;
                }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt) {
            const u16x8 o = *reinterpret_cast<const u16x8*>(strip + nt * G::STRIP + ((s_have ? sm : 0) * 16 + sv) * 16 + sh * 8);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrs,
                                                   (int)((st_ok && zo < zb) ? st_off + (unsigned)nt * nt_b + (unsigned)zo * zstep_b : 0xFFFFFFF0u), 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < S; ++q) store_plane(S * zo + 2 + q, cur[q]);
        __syncthreads();
    };
#pragma unroll 1
    for (int zo = za; zo < zb; zo += 3) {   // up to two surplus steps at the end of a chunk: their stores are dropped
        step(zo, w[0], w[2]);
        step(zo + 1, w[1], w[0]);
        step(zo + 2, w[2], w[1]);
    }
}

template <int DT, int CIN, int COUT, int S>
static int run_convz16(const void* x, void* y, const unsigned short* bp, const float* bias, int Di, int Hi, int Wi,
                       int Do, int Ho, int Wo, int cus, hipStream_t s) {
    using G = ConvZ16<CIN, COUT, S>;
    const int nbx = (Wo + G::TX - 1) / G::TX, nby = (Ho + G::TY - 1) / G::TY, ncol = nbx * nby;
    int best = 1;
    double best_eff = 0.0;
    for (int nz = 1; nz <= (Do + 3) / 4; ++nz) {   // z chunks of >= 4 output planes: best fill of the last round
        const int zc = (Do + nz - 1) / nz, nzc = (Do + zc - 1) / zc;
        const long nb = (long)ncol * nzc;
        const double eff = (double)nb / (double)(((nb + cus - 1) / cus) * cus) * zc / (zc + 2.5);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = nz; }
    }
    const int ZC = (Do + best - 1) / best, nzc = (Do + ZC - 1) / ZC;
    convz16_mfma_kernel<DT, CIN, COUT, S><<<ncol * nzc, G::THREADS, 0, s>>>(x, bp, bias, y, Di, Hi, Wi, Do, Ho, Wo, ZC, nbx, nby);
    return check_hip(hipGetLastError(), "convz16_mfma launch");
}

// layers 1..3 through the z-marching kernel when their columns fill the chip (MVS_CONVZ16=0/1 = never / always)
template <int DT>
static int try_convz16(int layer, const void* x, void* y, const unsigned short* bp, const float* bias, int Di, int Hi,
                       int Wi, hipStream_t s, bool* taken) {
    *taken = false;
    static const int zm = [] {
        const char* e = getenv("MVS_CONVZ16");
        return e ? atoi(e) : -1;
    }();
    if (zm == 0 || layer < 1 || layer > 3) return MVS_OK;
    const int S = layer == 2 ? 1 : 2, cout = layer == 3 ? 32 : 16;
    const int Do = (Di - 1) / S + 1, Ho = (Hi - 1) / S + 1, Wo = (Wi - 1) / S + 1;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int tx = S == 1 ? 32 : 16;
    const long ncol = (long)((Wo + tx - 1) / tx) * ((Ho + 7) / 8);
    if ((size_t)Do * Ho * Wo * cout * 2 >= ((size_t)1 << 32) - 64 || Do < 4) return MVS_OK;
    // enough columns x 8-plane chunks to fill the chip (cfg5's conv3 -- 12 columns of 48 planes -- measured 0.016 vs
    // 0.012 ms for the tile kernel: short chunks pay the two-step prologue too often)
    if (zm != 1 && ncol * (Do / 8) * 4 < (long)cus * 3) return MVS_OK;
    // round 4, after the tile kernels got z-deep tiles, the XCD-aware order and staged epilogues: conv3 is faster on the tile
    // kernel at every bench size (cfg3 0.0433-0.0448 against 0.0456-0.0463 ms, cfg5 0.0108 against 0.0154), conv2 on it at cfg5's
    // size (0.0178 against 0.0204-0.0214) but not at cfg3's (0.106 against 0.087); conv1 stays here (cfg5 0.0215 / 0.0256, cfg3
    // 0.113 / 0.148)
    if (zm != 1 && (layer == 3 || (layer == 2 && ncol * (Do / 8) < (long)cus * 3))) return MVS_OK;
    *taken = true;
    switch (layer) {
        case 1: return run_convz16<DT, 8, 16, 2>(x, y, bp, bias, Di, Hi, Wi, Do, Ho, Wo, cus, s);
        case 2: return run_convz16<DT, 16, 16, 1>(x, y, bp, bias, Di, Hi, Wi, Do, Ho, Wo, cus, s);
        default: return run_convz16<DT, 16, 32, 2>(x, y, bp, bias, Di, Hi, Wi, Do, Ho, Wo, cus, s);
    }
}

// wfold [27][32][8] -> bp [4][9][64][8] (16-bit): k = (g = kx', j = ci of the chunk), n = (jj, co)
void pack_conv0p16_weights(const float* wfold, int dt, unsigned short* bp) {
    for (int c = 0; c < 4; ++c)
        for (int ks = 0; ks < 9; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int g = lane >> 4, n = lane & 15;
                    const int kz = ks / 3, ky = ks % 3, jj = n >> 3, co = n & 7;
                    const int kx = g - jj, ci = 8 * c + j;
                    float v = 0.0f;
                    if (kx >= 0 && kx <= 2) v = wfold[((size_t)(kz * 9 + ky * 3 + kx) * 32 + ci) * 8 + co];
                    bp[(((size_t)c * 9 + ks) * 64 + lane) * 8 + j] = to_bits16(v, dt);
                }
}

// =============================================================================================
// deconvg16: conv7 / conv9 / conv11 (+ skip)
// =============================================================================================
// Deconv16Tap / deconv16_tap(ks, q): mvs_internal.h (shared with conv11_prob.hip)

template <int CIN, int COUT, int BZ, int BY, int BX>
struct DeconvG16 {
    static constexpr int NTT = 2 * COUT / 16;
    static constexpr int MG = 4 / NTT;
    static constexpr int MT = BZ * BY * BX;
    static constexpr int MPW = MT / MG;
    static constexpr int NCH = CIN / 8;
    static constexpr int KS = 5;
    static constexpr int VS = 8;
    static constexpr int HZ = BZ + 1, HY = 2 * BY + 1, HX = 8 * BX + 1;
    static constexpr int HXP = ((HX + 7) / 8 * 8) | 8;
    static constexpr int TILE_ELEMS = HZ * HY * HXP * VS;
    static constexpr int NPIECE = HZ * HY * HX;
    static constexpr int PPT = (NPIECE + 255) / 256;
    static_assert(NTT == 1 || NTT == 2 || NTT == 4, "COUT must be 8, 16 or 32");
    static_assert(MT % MG == 0, "block tile must split evenly over the M-groups");
};

template <int DT, int CIN, int COUT, int BZ, int BY, int BX>
__global__ __launch_bounds__(256) void deconvg16_mfma_kernel(
    const void* __restrict__ x, const unsigned short* __restrict__ bp, const float* __restrict__ bias,
    const void* __restrict__ skip, void* __restrict__ y, int Di, int Hi, int Wi) {
    using G = DeconvG16<CIN, COUT, BZ, BY, BX>;
    // epilogue staging tile (fp32; deconvg_mfma's scheme, see deconvgs_mfma_kernel): one z parity of the block's output
    constexpr int OY = 4 * BY, OX = 16 * BX;
    constexpr int RP0 = OX * COUT + 16 * (OX / 8), RP = (RP0 % 32 == 16) ? RP0 : RP0 + 16;
    constexpr int OUT_FLOATS = BZ * OY * RP, NUNIT = BZ * OY * OX * (COUT / 8), UPT = (NUNIT + 255) / 256;
    constexpr int LDS_BYTES = G::TILE_ELEMS * 2 > OUT_FLOATS * 4 ? G::TILE_ELEMS * 2 : OUT_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    unsigned short* tile = reinterpret_cast<unsigned short*>(lds_raw);
    float* otile = reinterpret_cast<float*>(lds_raw);
    const unsigned short* xs = static_cast<const unsigned short*>(x);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NTT, mg = wave / G::NTT;
    const int nbx = (Wi + 8 * BX - 1) / (8 * BX), nby = (Hi + 2 * BY - 1) / (2 * BY);
    int b;   // XCD-aware tile order (see convgs_mfma_kernel)
    {
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ix0 = bx * 8 * BX, iy0 = by * 2 * BY, iz0 = bz * BZ;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t Vin = (size_t)Di * Hi * Wi, Vout = Vin * 8;

    int goff[G::PPT], loff[G::PPT];
    unsigned inside = 0;
#pragma unroll
    for (int i = 0; i < G::PPT; ++i) {
        const int v = tid + i * 256;
        const int hx = v % G::HX, t = v / G::HX;
        const int hy = t % G::HY, hz = t / G::HY;
        const int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = v < G::NPIECE && gz < Di && gy < Hi && gx < Wi;
        goff[i] = ok ? (int)((((size_t)gz * Hi + gy) * Wi + gx) * 8) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = (v < G::NPIECE) ? ((hz * G::HY + hy) * G::HXP + hx) * G::VS : -1;
    }

    const int r = lane & 15, g = lane >> 4;
    const int ry = r >> 3, rx = r & 7;
    int abase[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
        abase[i] = ((tz * G::HY + 2 * ty + ry) * G::HXP + 8 * tx + rx + (g & 1)) * G::VS;
    }
    int koff[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        const Deconv16Tap t0 = deconv16_tap(ks, 0), t1 = deconv16_tap(ks, 1);
        koff[ks] = (g >> 1) ? (t1.dz * G::HY + t1.dy) * G::HXP * G::VS : (t0.dz * G::HY + t0.dy) * G::HXP * G::VS;
    }

    f32x4 acc[4][G::MPW];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) acc[c][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 breg[G::KS];
    u32x4 stg[G::PPT];

#define MVS_LOAD_B(C)                                                                               \
    {                                                                                               \
        const u32x4* bsrc =                                                                         \
            reinterpret_cast<const u32x4*>(bp) + ((size_t)((C) * G::NTT + nt) * G::KS) * 64 + lane; \
        _Pragma("unroll") for (int ks = 0; ks < G::KS; ++ks) breg[ks] = bsrc[ks * 64];              \
    }
#define MVS_LOAD_A(C)                                                                               \
    {                                                                                               \
        const unsigned short* plane = xs + (size_t)(C) * Vin * 8;                                   \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            stg[i] = *reinterpret_cast<const u32x4*>(plane + goff[i]);                              \
    }
#define MVS_STORE_A()                                                                               \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            if (loff[i] >= 0)                                                                       \
                *reinterpret_cast<u32x4*>(tile + loff[i]) =                                         \
                    ((inside >> i) & 1u) ? stg[i] : (u32x4){0u, 0u, 0u, 0u};                        \
    }

    MVS_LOAD_B(0)
    MVS_LOAD_A(0)
    MVS_STORE_A()
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < G::NCH; ++c) {
        if (c + 1 < G::NCH) MVS_LOAD_A(c + 1)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            constexpr int dummy = 0;
            (void)dummy;
            const int cls = deconv16_tap(ks, 0).cls;
#pragma unroll
            for (int i = 0; i < G::MPW; ++i) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(tile + abase[i] + koff[ks]);
                acc[cls][i] = mfma16<DT>(a, breg[ks], acc[cls][i]);
            }
        }
        if (c + 1 < G::NCH) {
            MVS_LOAD_B(c + 1)
            __syncthreads();
            MVS_STORE_A()
            __syncthreads();
        }
    }
#undef MVS_LOAD_B
#undef MVS_LOAD_A
#undef MVS_STORE_A

    // epilogue: ReLU(acc + bias) scattered into the LDS tile [oz][oy][ox][co] (fp32), then every thread owns whole voxels of one
    // C8 plane (8 channels = 16 B of 16-bit storage): skip add with 16-byte loads / stores contiguous across the wave (the
    // scalar form stored 2-byte elements).  Two passes, one per output z parity; a pass requests its skip values first.
    const int Do = 2 * Di;
    const int nn = 16 * nt + (lane & 15);
    const int px = nn / COUT, co = nn % COUT;
    const float bv = bias[co];
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
        size_t uo[UPT];
        int usrc[UPT];
        float sk16[UPT][8];
#pragma unroll
        for (int j = 0; j < UPT; ++j) {
            const int u = tid + j * 256;
            const int ox = u % OX;
            int t = u / OX;
            const int oy = t % OY; t /= OY;
            const int ozl = t % BZ, pl = t / BZ;
            const int gz = 2 * (iz0 + ozl) + pz, gy = 2 * iy0 + oy, gx = 2 * ix0 + ox;
            const bool ok = u < NUNIT && gz < Do && gy < Ho && gx < Wo;
            usrc[j] = ok ? (ozl * OY + oy) * RP + ox * COUT + (ox >> 3) * 16 + pl * 8 : -1;
            uo[j] = ok ? ((size_t)pl * Vout + ((size_t)gz * Ho + gy) * Wo + gx) * 8 : 0;
            load8_16<DT>(skip, uo[j], sk16[j]);
        }
        __syncthreads();  // input tile (pass 0) / previous pass's staging tile fully consumed
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) {
            const int t = mg * G::MPW + i;
            const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * (lane >> 4) + e;
                const int ly = 2 * ty + (m >> 3), lx = 8 * tx + (m & 7);  // input voxel inside the block tile
                const int ox = 2 * lx + px;
#pragma unroll
                for (int py = 0; py < 2; ++py)
                    otile[(tz * OY + 2 * ly + py) * RP + ox * COUT + (ox >> 3) * 16 + co] =
                        fmaxf(acc[2 * pz + py][i][e] + bv, 0.0f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < UPT; ++j) {
            if (usrc[j] < 0) continue;
            const float* src = otile + usrc[j];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
            const float v[8] = {lo.x + sk16[j][0], lo.y + sk16[j][1], lo.z + sk16[j][2], lo.w + sk16[j][3],
                                hi.x + sk16[j][4], hi.y + sk16[j][5], hi.z + sk16[j][6], hi.w + sk16[j][7]};
            store8_16<DT>(y, uo[j], v);
        }
    }
}

template <int DT, int CIN, int COUT, int BZ, int BY, int BX>
static int run_deconvg16(const void* x, const void* skip, void* y, const unsigned short* bp,
                         const float* bias, int Di, int Hi, int Wi, hipStream_t s) {
    if ((size_t)Di * Hi * Wi * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "deconvg16_mfma: plane exceeds 31-bit offsets");
    const int nb = ((Wi + 8 * BX - 1) / (8 * BX)) * ((Hi + 2 * BY - 1) / (2 * BY)) * ((Di + BZ - 1) / BZ);
    deconvg16_mfma_kernel<DT, CIN, COUT, BZ, BY, BX><<<nb, 256, 0, s>>>(x, bp, bias, skip, y, Di, Hi, Wi);
    return check_hip(hipGetLastError(), "deconvg16_mfma launch");
}

// wfold [27][cin][cout] -> bp [cin/8][2*cout/16][5][64][8] (16-bit)
void pack_deconvg16_weights(const float* wfold, int cin, int cout, int dt, unsigned short* bp) {
    const int nch = cin / 8, ntt = 2 * cout / 16;
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < ntt; ++t)
            for (int ks = 0; ks < 5; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int g = lane >> 4, n = lane & 15, dx = g & 1;
                        const Deconv16Tap tp = deconv16_tap(ks, g >> 1);
                        const int ci = 8 * c + j;
                        const int nn = 16 * t + n, px = nn / cout, co = nn % cout;
                        const int kx = px == 0 ? (dx == 0 ? 1 : -1) : (dx == 0 ? 2 : 0);
                        const float v = (!tp.valid || kx < 0)
                                            ? 0.0f
                                            : wfold[((size_t)(tp.kz * 9 + tp.ky * 3 + kx) * cin + ci) * cout + co];
                        bp[((((size_t)c * ntt + t) * 5 + ks) * 64 + lane) * 8 + j] = to_bits16(v, dt);
                    }
}

// =============================================================================================
// dispatch for layers 0..9 in the 16-bit storage modes
// =============================================================================================
template <int DT>
static int launch_layer16_dt(int layer, const void* x, const void* skip, void* y, const unsigned short* bp,
                             const float* bias, int Di, int Hi, int Wi, hipStream_t s) {
    if (layer == 0) {
        using namespace c16;
        if ((size_t)Di * Hi * Wi * 8 >= ((size_t)1 << 31))
            return fail(MVS_ERR_BAD_SHAPE, "conv0p16_mfma: plane exceeds 31-bit offsets");
        // z-marching kernel once its (y, x) columns can fill the chip; MVS_CONV0Z16=0 keeps the tile kernel (A/B runs)
        static const int zmarch = [] {   // 1 = also at small shapes (tests), 0 = never
            const char* e = getenv("MVS_CONV0Z16");
            return e ? atoi(e) : -1;
        }();
        const int ncol = ((Wi + c0z::TX - 1) / c0z::TX) * ((Hi + c0z::TY - 1) / c0z::TY);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        if (zmarch != 0 && Di >= 8 && (size_t)Di * Hi * Wi * 16 < ((size_t)1 << 32) - 64 &&
            (zmarch == 1 || (size_t)ncol * (Di / 8) >= (size_t)cus / 2)) {
            // z chunks (>= 8 planes each; every chunk re-reads 2 planes): the split that fills the last round of
            // one-block-per-CU best
            int best = 1;
            double best_eff = 0.0;
            for (int nz = 1; nz <= Di / 8; ++nz) {
                const int zc = (Di + nz - 1) / nz, nzc = (Di + zc - 1) / zc;
                const long nb = (long)ncol * nzc;
                const long slots = (long)cus * (8 / c0z::TY);   // blocks resident at once
                const double eff = (double)nb / (double)(((nb + slots - 1) / slots) * slots) * zc / (zc + 2.0);
                if (eff > best_eff + 1e-9) { best_eff = eff; best = nz; }
            }
            const int ZC = (Di + best - 1) / best, nzc = (Di + ZC - 1) / ZC;
            conv0z16_mfma_kernel<DT><<<ncol * nzc, c0z::THREADS, 0, s>>>(x, bp, bias, y, Di, Hi, Wi, ZC,
                                                                          (Wi + c0z::TX - 1) / c0z::TX,
                                                                          (Hi + c0z::TY - 1) / c0z::TY);
            return check_hip(hipGetLastError(), "conv0z16_mfma launch");
        }
        const int nb = ((Wi + TX - 1) / TX) * ((Hi + TY - 1) / TY) * ((Di + TZ - 1) / TZ);
        conv0p16_mfma_kernel<DT><<<nb, 256, 0, s>>>(x, bp, bias, y, Di, Hi, Wi);
        return check_hip(hipGetLastError(), "conv0p16_mfma launch");
    }
    if (layer <= 3) {
        bool taken = false;
        const int st = try_convz16<DT>(layer, x, y, bp, bias, Di, Hi, Wi, s, &taken);
        if (taken) return st;
    }
    if (layer <= 6) return launch_convg16_dt<DT>(layer, x, y, bp, bias, Di, Hi, Wi, s);
    switch (layer) {
        // the same for the transposed layers (tiles over the INPUT grid): cfg3 conv7 1x1x1 -> 2x1x1 0.0486 -> 0.0389, conv9 1x4x1
        // -> 4x2x1 0.1059 -> 0.1026; cfg5 conv9 0.0171 -> 0.0164, conv7 stays (2x1x1: 240 blocks)
        case 7:
            if (tile_fills_chip(Di, Hi, Wi, 1, 2, 1, 1)) return run_deconvg16<DT, 64, 32, 2, 1, 1>(x, skip, y, bp, bias, Di, Hi, Wi, s);
            return run_deconvg16<DT, 64, 32, 1, 1, 1>(x, skip, y, bp, bias, Di, Hi, Wi, s);
        case 8:
            if (tile_fills_chip(Di, Hi, Wi, 1, 4, 2, 1)) return run_deconvg16<DT, 32, 16, 4, 2, 1>(x, skip, y, bp, bias, Di, Hi, Wi, s);
            return run_deconvg16<DT, 32, 16, 1, 4, 1>(x, skip, y, bp, bias, Di, Hi, Wi, s);
        case 9: return run_deconvg16<DT, 16, 8, 1, 4, 2>(x, skip, y, bp, bias, Di, Hi, Wi, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "mfma16: layer %d not covered", layer);
    }
}

int launch_layer_mfma16(int layer, const void* x, const void* skip, void* y, const void* panel,
                        const float* bias, int Di, int Hi, int Wi, int dtype, hipStream_t s) {
    const unsigned short* bp = static_cast<const unsigned short*>(panel);
    if (dtype == MVS_F16) return launch_layer16_dt<MVS_F16>(layer, x, skip, y, bp, bias, Di, Hi, Wi, s);
    if (dtype == MVS_BF16) return launch_layer16_dt<MVS_BF16>(layer, x, skip, y, bp, bias, Di, Hi, Wi, s);
    return fail(MVS_ERR_BAD_DTYPE, "mfma16 kernels need fp16 or bf16 storage (dtype %d)", dtype);
}

// =============================================================================================
// SPLIT-OPERAND kernels for fp32 volumes (round 4): the tile kernels above with every fp32 operand written as the sum
// of three bf16 numbers (a = a1 + a2 + a3, RNE, exact residuals) and the fp32 product evaluated as the six leading
// cross products a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the
// dropped terms are <= 2^-26 of the product (csrc/conv0_split.hip has the argument), the per-layer bounds of the
// fp32-MFMA kernels hold unchanged, and six bf16 MFMAs cost 6/16 of the fp32 MFMA time they replace.  Inputs and
// outputs are the fp32 C8-planar volumes; a staged voxel (8 channels, two 16-byte loads) is split once, on its way
// into LDS, into three 16-byte bf16 fragments (three tiles); the weights are split on the host (three panels in the
// layout of the 16-bit kernels: pack_split_panels).
//   convgs   : conv2 .. conv4 (the tile scheme of convg16)
// =============================================================================================
template <int CIN, int COUT, int S, int BZ, int BY, int BX>
__global__ __launch_bounds__(256) void convgs_mfma_kernel(
    const float* __restrict__ x,              // [CIN/8][Di][Hi][Wi][8] fp32
    const unsigned short* __restrict__ bp,    // [3 pieces][NCH][NT][7][64][8] bf16
    const float* __restrict__ bias,           // [COUT]
    float* __restrict__ y,                    // [COUT/8][Do][Ho][Wo][8] fp32
    int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
    using G = ConvG16<CIN, COUT, S, BZ, BY, BX>;
    // epilogue staging tile (fp32): the block's BZ x 2BY x 8BX output voxels x COUT channels, rows padded (16 floats per 8
    // voxels, row pitch = 16 mod 32 floats) so that the accumulator scatter is conflict-free
    constexpr int OY = 2 * BY, OX = 8 * BX;
    constexpr int RP0 = OX * COUT + 16 * (OX / 8), RP = (RP0 % 32 == 16) ? RP0 : RP0 + 16;
    constexpr int OUT_FLOATS = BZ * OY * RP, NUNIT = BZ * OY * OX * (COUT / 8), UPT = (NUNIT + 255) / 256;
    constexpr int LDS_BYTES = 3 * G::TILE_ELEMS * 2 > OUT_FLOATS * 4 ? 3 * G::TILE_ELEMS * 2 : OUT_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    unsigned short* tile = reinterpret_cast<unsigned short*>(lds_raw);
    float* otile = reinterpret_cast<float*>(lds_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NT, mg = wave / G::NT;
    const int nbx = (Wo + 8 * BX - 1) / (8 * BX), nby = (Ho + 2 * BY - 1) / (2 * BY);
    // blocks are dealt round-robin over the 8 XCDs (blockIdx.x % 8 names the XCD: speed only, never correctness): XCD k works
    // through the k-th eighth of the (z, row, column) tile sequence, so that tiles sharing halo planes / rows meet in one L2
    int b;
    {
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ox0 = bx * 8 * BX, oy0 = by * 2 * BY, oz0 = bz * BZ;
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;
    const size_t Vin = (size_t)Di * Hi * Wi, Vout = (size_t)Do * Ho * Wo;
    constexpr size_t PANEL = (size_t)G::NCH * G::NT * G::KS * 64;   // u32x4 fragments per piece

    // staging: piece p = tid + i * 256 = one voxel (8 channels) of the halo tile; threads beyond the tile shadow its
    // last voxel (no branch in the staging code)
    int goff[G::PPT], loff[G::PPT];
    unsigned inside = 0;
#pragma unroll
    for (int i = 0; i < G::PPT; ++i) {
        const int v = min(tid + i * 256, G::NPIECE - 1);
        const int hx = v % G::HX, t = v / G::HX;
        const int hy = t % G::HY, hz = t / G::HY;
        const int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = gz >= 0 && gz < Di && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
        goff[i] = ok ? (int)((((size_t)gz * Hi + gy) * Wi + gx) * 8) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = ((hz * G::HY + hy) * G::HXP + hx) * G::VS;
    }

    // A fragment: lane (r = lane&15 -> voxel (ry, rx) of the M-tile, g = lane>>4 -> tap 4ks+g)
    const int r = lane & 15, g = lane >> 4;
    const int ry = r >> 3, rx = r & 7;
    int abase[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
        abase[i] = (((tz * S) * G::HY + (2 * ty + ry) * S) * G::HXP + (8 * tx + rx) * S) * G::VS;
    }
    int koff[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
        koff[ks] = g == 0 ? G::tap_off(4 * ks) : g == 1 ? G::tap_off(4 * ks + 1)
                 : g == 2 ? G::tap_off(4 * ks + 2) : G::tap_off(4 * ks + 3);

    f32x4 acc[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 breg[3][G::KS];
    f32x4 stg[G::PPT][2];

    auto load_b = [&](int c) {
        const u32x4* bsrc = reinterpret_cast<const u32x4*>(bp) + ((size_t)(c * G::NT + nt) * G::KS) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) breg[q][ks] = bsrc[q * PANEL + ks * 64];
    };
    auto load_a = [&](int c) {
        const float* plane = x + (size_t)c * Vin * 8;
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) {
            stg[i][0] = *reinterpret_cast<const f32x4*>(plane + goff[i]);
            stg[i][1] = *reinterpret_cast<const f32x4*>(plane + goff[i] + 4);
        }
    };
    auto store_a = [&]() {
        const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) {
            const bool in = (inside >> i) & 1u;
            u32x4 p1, p2, p3;
            gs_split8(in ? stg[i][0] : z4, in ? stg[i][1] : z4, p1, p2, p3);
            *reinterpret_cast<u32x4*>(tile + loff[i]) = p1;
            *reinterpret_cast<u32x4*>(tile + G::TILE_ELEMS + loff[i]) = p2;
            *reinterpret_cast<u32x4*>(tile + 2 * G::TILE_ELEMS + loff[i]) = p3;
        }
    };

    load_b(0);
    load_a(0);
    store_a();
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < G::NCH; ++c) {
        if (c + 1 < G::NCH) load_a(c + 1);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
#pragma unroll
            for (int i = 0; i < G::MPW; ++i) {
                const unsigned short* ap = tile + abase[i] + koff[ks];
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(ap);
                const u32x4 a2 = *reinterpret_cast<const u32x4*>(ap + G::TILE_ELEMS);
                const u32x4 a3 = *reinterpret_cast<const u32x4*>(ap + 2 * G::TILE_ELEMS);
                acc[i] = gs_mfma(a3, breg[0][ks], acc[i]);   // the small terms first
                acc[i] = gs_mfma(a1, breg[2][ks], acc[i]);
                acc[i] = gs_mfma(a2, breg[1][ks], acc[i]);
                acc[i] = gs_mfma(a2, breg[0][ks], acc[i]);
                acc[i] = gs_mfma(a1, breg[1][ks], acc[i]);
                acc[i] = gs_mfma(a1, breg[0][ks], acc[i]);
            }
        }
        if (c + 1 < G::NCH) {
            load_b(c + 1);
            __syncthreads();
            store_a();
            __syncthreads();
        }
    }

    // epilogue: ReLU(acc + bias) scattered into the LDS tile [oz][oy][ox][co], then every thread owns whole voxels of one C8
    // plane (8 channels) and stores them 16 bytes at a time, contiguous across the wave (the scalar form stored single
    // elements: 2- or 4-byte pieces in 16- or 32-byte runs)
    const int n = lane & 15, co = 16 * nt + n;
    const float bv = bias[co];
    __syncthreads();   // the input tile is fully consumed
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 4 * (lane >> 4) + e;
            const int oy = 2 * ty + (m >> 3), ox = 8 * tx + (m & 7);
            otile[(tz * OY + oy) * RP + ox * COUT + (ox >> 3) * 16 + co] = fmaxf(acc[i][e] + bv, 0.0f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
        const int u = tid + j * 256;
        const int ox = u % OX;
        int t = u / OX;
        const int oy = t % OY; t /= OY;
        const int ozl = t % BZ, pl = t / BZ;
        const int gz = oz0 + ozl, gy = oy0 + oy, gx = ox0 + ox;
        if (u < NUNIT && gz < Do && gy < Ho && gx < Wo) {
            const float* src = otile + (ozl * OY + oy) * RP + ox * COUT + (ox >> 3) * 16 + pl * 8;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
            const size_t uo = ((size_t)pl * Vout + ((size_t)gz * Ho + gy) * Wo + gx) * 8;
            *reinterpret_cast<f32x4*>(y + uo) = lo;
            *reinterpret_cast<f32x4*>(y + uo + 4) = hi;
        }
    }
}

template <int CIN, int COUT, int S, int BZ, int BY, int BX>
static int run_convgs(const void* x, void* y, const unsigned short* bp, const float* bias, int Di, int Hi, int Wi,
                      hipStream_t s) {
    const int Do = (Di - 1) / S + 1, Ho = (Hi - 1) / S + 1, Wo = (Wi - 1) / S + 1;
    if ((size_t)Di * Hi * Wi * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "convgs_mfma: plane exceeds 31-bit offsets");
    const int nb = ((Wo + 8 * BX - 1) / (8 * BX)) * ((Ho + 2 * BY - 1) / (2 * BY)) * ((Do + BZ - 1) / BZ);
    convgs_mfma_kernel<CIN, COUT, S, BZ, BY, BX><<<nb, 256, 0, s>>>(static_cast<const float*>(x), bp, bias,
                                                                    static_cast<float*>(y), Di, Hi, Wi, Do, Ho, Wo);
    return check_hip(hipGetLastError(), "convgs_mfma launch");
}

// deconvgs: the transposed layers conv7 / conv9 (+ skip) for fp32 volumes with split operands -- deconvg16's tile scheme
// and tap order (deconv16_tap: 5 k-steps per 8-channel chunk, 4 (pz, py) classes, x parity folded into N), fp32 in / out,
// three bf16 tiles, three panels (pack_split_panels), six MFMAs per k-step with the small terms first.
template <int CIN, int COUT, int BZ, int BY, int BX>
__global__ __launch_bounds__(256) void deconvgs_mfma_kernel(
    const float* __restrict__ x,              // [CIN/8][Di][Hi][Wi][8] fp32
    const unsigned short* __restrict__ bp,    // [3 pieces][NCH][NTT][5][64][8] bf16
    const float* __restrict__ bias,           // [COUT]
    const float* __restrict__ skip,           // [COUT/8][2 Di][2 Hi][2 Wi][8] fp32
    float* __restrict__ y, int Di, int Hi, int Wi) {
    using G = DeconvG16<CIN, COUT, BZ, BY, BX>;
    // epilogue staging tile (deconvg_mfma's scheme): one z parity of the block's 2BZ x 4BY x 16BX output voxels x COUT channels,
    // rows padded (16 floats per 8 voxels, row pitch = 16 mod 32) so that the accumulator scatter is conflict-free
    constexpr int OY = 4 * BY, OX = 16 * BX;
    constexpr int RP0 = OX * COUT + 16 * (OX / 8), RP = (RP0 % 32 == 16) ? RP0 : RP0 + 16;
    constexpr int OUT_FLOATS = BZ * OY * RP, NUNIT = BZ * OY * OX * (COUT / 8), UPT = (NUNIT + 255) / 256;
    constexpr int LDS_BYTES = 3 * G::TILE_ELEMS * 2 > OUT_FLOATS * 4 ? 3 * G::TILE_ELEMS * 2 : OUT_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    unsigned short* tile = reinterpret_cast<unsigned short*>(lds_raw);
    float* otile = reinterpret_cast<float*>(lds_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NTT, mg = wave / G::NTT;
    const int nbx = (Wi + 8 * BX - 1) / (8 * BX), nby = (Hi + 2 * BY - 1) / (2 * BY);
    // blocks are dealt round-robin over the 8 XCDs (blockIdx.x % 8 names the XCD: speed only, never correctness): XCD k works
    // through the k-th eighth of the (z, row, column) tile sequence, so that tiles sharing halo planes / rows meet in one L2
    int b;
    {
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ix0 = bx * 8 * BX, iy0 = by * 2 * BY, iz0 = bz * BZ;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t Vin = (size_t)Di * Hi * Wi, Vout = Vin * 8;
    constexpr size_t PANEL = (size_t)G::NCH * G::NTT * G::KS * 64;   // u32x4 fragments per piece

    // staging: piece p = tid + i * 256 = one voxel (8 channels) of the halo tile; threads beyond the tile shadow its
    // last voxel (no branch in the staging code)
    int goff[G::PPT], loff[G::PPT];
    unsigned inside = 0;
#pragma unroll
    for (int i = 0; i < G::PPT; ++i) {
        const int v = min(tid + i * 256, G::NPIECE - 1);
        const int hx = v % G::HX, t = v / G::HX;
        const int hy = t % G::HY, hz = t / G::HY;
        const int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = gz < Di && gy < Hi && gx < Wi;
        goff[i] = ok ? (int)((((size_t)gz * Hi + gy) * Wi + gx) * 8) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = ((hz * G::HY + hy) * G::HXP + hx) * G::VS;
    }

    const int r = lane & 15, g = lane >> 4;
    const int ry = r >> 3, rx = r & 7;
    int abase[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
        abase[i] = ((tz * G::HY + 2 * ty + ry) * G::HXP + 8 * tx + rx + (g & 1)) * G::VS;
    }
    int koff[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
        const Deconv16Tap t0 = deconv16_tap(ks, 0), t1 = deconv16_tap(ks, 1);
        koff[ks] = (g >> 1) ? (t1.dz * G::HY + t1.dy) * G::HXP * G::VS : (t0.dz * G::HY + t0.dy) * G::HXP * G::VS;
    }

    f32x4 acc[4][G::MPW];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) acc[c][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 breg[3][G::KS];
    f32x4 stg[G::PPT][2];

    auto load_b = [&](int c) {
        const u32x4* bsrc = reinterpret_cast<const u32x4*>(bp) + ((size_t)(c * G::NTT + nt) * G::KS) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) breg[q][ks] = bsrc[q * PANEL + ks * 64];
    };
    auto load_a = [&](int c) {
        const float* plane = x + (size_t)c * Vin * 8;
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) {
            stg[i][0] = *reinterpret_cast<const f32x4*>(plane + goff[i]);
            stg[i][1] = *reinterpret_cast<const f32x4*>(plane + goff[i] + 4);
        }
    };
    auto store_a = [&]() {
        const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) {
            const bool in = (inside >> i) & 1u;
            u32x4 p1, p2, p3;
            gs_split8(in ? stg[i][0] : z4, in ? stg[i][1] : z4, p1, p2, p3);
            *reinterpret_cast<u32x4*>(tile + loff[i]) = p1;
            *reinterpret_cast<u32x4*>(tile + G::TILE_ELEMS + loff[i]) = p2;
            *reinterpret_cast<u32x4*>(tile + 2 * G::TILE_ELEMS + loff[i]) = p3;
        }
    };

    load_b(0);
    load_a(0);
    store_a();
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < G::NCH; ++c) {
        if (c + 1 < G::NCH) load_a(c + 1);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const int cls = deconv16_tap(ks, 0).cls;
#pragma unroll
            for (int i = 0; i < G::MPW; ++i) {
                const unsigned short* ap = tile + abase[i] + koff[ks];
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(ap);
                const u32x4 a2 = *reinterpret_cast<const u32x4*>(ap + G::TILE_ELEMS);
                const u32x4 a3 = *reinterpret_cast<const u32x4*>(ap + 2 * G::TILE_ELEMS);
                acc[cls][i] = gs_mfma(a3, breg[0][ks], acc[cls][i]);   // the small terms first
                acc[cls][i] = gs_mfma(a1, breg[2][ks], acc[cls][i]);
                acc[cls][i] = gs_mfma(a2, breg[1][ks], acc[cls][i]);
                acc[cls][i] = gs_mfma(a2, breg[0][ks], acc[cls][i]);
                acc[cls][i] = gs_mfma(a1, breg[1][ks], acc[cls][i]);
                acc[cls][i] = gs_mfma(a1, breg[0][ks], acc[cls][i]);
            }
        }
        if (c + 1 < G::NCH) {
            load_b(c + 1);
            __syncthreads();
            store_a();
            __syncthreads();
        }
    }

    // epilogue: col n -> (px, co); row m -> input voxel of the tile; class -> (pz, py).  ReLU(acc + bias) is scattered into the
    // LDS tile [oz][oy][ox][co]; then every thread owns whole voxels of one C8 plane (32 B) and does the skip add with 16-byte
    // loads / stores that are contiguous across the wave (the scalar form: 4-byte stores in 32-byte runs and one exposed skip
    // load per element).  Two passes, one per output z parity; the skip values of a pass are requested before its scatter.
    const int Do = 2 * Di;
    const int nn = 16 * nt + (lane & 15);
    const int px = nn / COUT, co = nn % COUT;
    const float bv = bias[co];
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
        size_t uo[UPT];
        int usrc[UPT];
        f32x4 sk0[UPT], sk1[UPT];
#pragma unroll
        for (int j = 0; j < UPT; ++j) {
            const int u = tid + j * 256;
            const int ox = u % OX;
            int t = u / OX;
            const int oy = t % OY; t /= OY;
            const int ozl = t % BZ, pl = t / BZ;
            const int gz = 2 * (iz0 + ozl) + pz, gy = 2 * iy0 + oy, gx = 2 * ix0 + ox;
            const bool ok = u < NUNIT && gz < Do && gy < Ho && gx < Wo;
            usrc[j] = ok ? (ozl * OY + oy) * RP + ox * COUT + (ox >> 3) * 16 + pl * 8 : -1;
            uo[j] = ok ? ((size_t)pl * Vout + ((size_t)gz * Ho + gy) * Wo + gx) * 8 : 0;
            sk0[j] = *reinterpret_cast<const f32x4*>(skip + uo[j]);
            sk1[j] = *reinterpret_cast<const f32x4*>(skip + uo[j] + 4);
        }
        __syncthreads();  // input tiles (pass 0) / previous pass's staging tile fully consumed
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) {
            const int t = mg * G::MPW + i;
            const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * (lane >> 4) + e;
                const int ly = 2 * ty + (m >> 3), lx = 8 * tx + (m & 7);  // input voxel inside the block tile
                const int ox = 2 * lx + px;
#pragma unroll
                for (int py = 0; py < 2; ++py)
                    otile[(tz * OY + 2 * ly + py) * RP + ox * COUT + (ox >> 3) * 16 + co] =
                        fmaxf(acc[2 * pz + py][i][e] + bv, 0.0f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < UPT; ++j) {
            if (usrc[j] < 0) continue;
            const float* src = otile + usrc[j];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
            *reinterpret_cast<f32x4*>(y + uo[j]) = lo + sk0[j];
            *reinterpret_cast<f32x4*>(y + uo[j] + 4) = hi + sk1[j];
        }
    }
}

template <int CIN, int COUT, int BZ, int BY, int BX>
static int run_deconvgs(const void* x, const void* skip, void* y, const unsigned short* bp, const float* bias,
                        int Di, int Hi, int Wi, hipStream_t s) {
    if ((size_t)Di * Hi * Wi * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "deconvgs_mfma: plane exceeds 31-bit offsets");
    const int nb = ((Wi + 8 * BX - 1) / (8 * BX)) * ((Hi + 2 * BY - 1) / (2 * BY)) * ((Di + BZ - 1) / BZ);
    deconvgs_mfma_kernel<CIN, COUT, BZ, BY, BX><<<nb, 256, 0, s>>>(static_cast<const float*>(x), bp, bias,
                                                                   static_cast<const float*>(skip),
                                                                   static_cast<float*>(y), Di, Hi, Wi);
    return check_hip(hipGetLastError(), "deconvgs_mfma launch");
}

// fp32 volumes, split operands: conv2, conv3, conv4 (measured at cfg2 against the fp32-MFMA kernels: 0.0588 -> 0.0534,
// 0.0310 -> 0.0276, 0.0396 -> 0.0329 ms with the 16-bit kernels' block tiles; a sweep of 8-20 tiles per layer
// (tools/gpu/tile_sweep.sh, BZ x BY x BX M-tiles of 2 x 8 outputs): conv3 1x2x2 -> 2x2x1 0.0275 -> 0.0238, conv4 1x2x2 ->
// 4x2x1 0.0325 -> 0.0254 -- z-deep, x-narrow tiles re-use the halo planes --, conv2 stays at 2x4x2).  Measured and NOT selected: conv5 / conv6 on this tile kernel (0.0182 / 0.0289
// against 0.0176 / 0.0242 ms for the all-K-resident split-K fp32 kernels: on 7,680 voxels the chunk pipeline is the cost,
// not the matrix pipe), conv1 on it (0.083 ms) and as a z-marching kernel with three bf16 rings (attic/conv1_split_zmarch.hip:
// 0.054 against 0.046 ms for the fp32-MFMA z-marching kernel)
// conv9 likewise (deconvgs<32, 16, 2, 4, 1>: 0.0354 -> 0.0266 ms at cfg2; tiles of 1x4x1 / 2x2x1 / 1x2x2 / 1x2x1 M-tiles per
// block: 0.0278 / 0.0277 / 0.0301 / 0.0316); conv7 on the same kernel measured 0.0248 against 0.0234 ms for the split-K
// fp32 kernel and is not selected.
bool split_layer_covers(int layer) {
    static const int deconv = [] {   // MVS_SPLIT_DECONV: bit 0 = conv7, bit 1 = conv9; default 2
        const char* e = getenv("MVS_SPLIT_DECONV");
        return e ? atoi(e) : 2;
    }();
    return (layer >= 2 && layer <= 4) || (layer == 7 && (deconv & 1)) || (layer == 8 && (deconv & 2));
}
int launch_layer_split(int layer, const void* x, const void* skip, void* y, const void* panel, const float* bias,
                       int Di, int Hi, int Wi, hipStream_t s) {
    const unsigned short* bp = static_cast<const unsigned short*>(panel);
    switch (layer) {
        case 2: return run_convgs<16, 16, 1, 2, 4, 2>(x, y, bp, bias, Di, Hi, Wi, s);
        case 3: return run_convgs<16, 32, 2, 2, 2, 1>(x, y, bp, bias, Di, Hi, Wi, s);
        case 4: return run_convgs<32, 32, 1, 4, 2, 1>(x, y, bp, bias, Di, Hi, Wi, s);
        case 7: return run_deconvgs<64, 32, 1, 1, 1>(x, skip, y, bp, bias, Di, Hi, Wi, s);
        case 8: return run_deconvgs<32, 16, 2, 4, 1>(x, skip, y, bp, bias, Di, Hi, Wi, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "split kernels: layer %d not covered", layer);
    }
}

// 16-bit elements of the panel of layer l (0..9)
size_t mfma16_panel_elems(int layer) {
    const LayerSpec& S = kLayers[layer];
    if (layer == 0) return (size_t)4 * 9 * 64 * 8;
    if (layer <= 6) return (size_t)(S.cin / 8) * (S.cout / 16) * 7 * 64 * 8;
    return (size_t)(S.cin / 8) * (2 * S.cout / 16) * 5 * 64 * 8;
}

void pack_mfma16_panel(int layer, const float* wfold, int dt, void* out) {
    unsigned short* bp = static_cast<unsigned short*>(out);
    const LayerSpec& S = kLayers[layer];
    if (layer == 0) pack_conv0p16_weights(wfold, dt, bp);
    else if (layer <= 6) pack_convg16_weights(wfold, S.cin, S.cout, dt, bp);
    else pack_deconvg16_weights(wfold, S.cin, S.cout, dt, bp);
}

// the three bf16 pieces of every folded weight, each packed like the bf16 panel of the 16-bit kernels:
// out = [3 pieces][mfma16_panel_elems(layer)] 16-bit
void pack_split_panels(int layer, const float* wfold, void* out) {
    const LayerSpec& S = kLayers[layer];
    const size_t n = (size_t)27 * S.cin * S.cout, elems = mfma16_panel_elems(layer);
    float* piece = new float[n];
    float* rest = new float[n];
    for (size_t i = 0; i < n; ++i) rest[i] = wfold[i];
    auto widen = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float v; std::memcpy(&v, &u, 4); return v; };
    for (int q = 0; q < 3; ++q) {
        for (size_t i = 0; i < n; ++i) {
            piece[i] = widen(to_bits16(rest[i], MVS_BF16));
            rest[i] -= piece[i];   // exact in fp32
        }
        pack_mfma16_panel(layer, piece, MVS_BF16, static_cast<unsigned short*>(out) + (size_t)q * elems);
    }
    delete[] piece;
    delete[] rest;
}

}  // namespace mvs
