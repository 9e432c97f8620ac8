// conv0_split.hip -- conv0 (32 -> 8 channels, reference models/mvsnet.py:36, block models/module.py:26-33) for fp32
// volumes with SPLIT OPERANDS on the bf16 matrix cores: Winograd F(4,3) along z as in conv_winograd.hip, but every
// transformed activation and every transformed weight is written as the sum of three bf16 numbers
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)        (round to nearest even)
// and the fp32 product a * b is evaluated as the six leading cross products
//     a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1)
// on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  A bf16 x bf16 product is exact in fp32; |a2| <= 2^-9 |a|,
// |a3| <= 2^-18 |a| (same for b), so the dropped terms a2 b3 + a3 b2 + a3 b3 are <= 2^-26 |a b| -- a quarter of an
// fp32 ulp of the product.  Arithmetic is therefore fp32-equivalent (not bit-equal: the MFMA adds its 32 products
// in its own order), at 16x the fp32 MFMA rate per instruction: 6 x 1/16 of the matrix-pipe time, and far less
// power than the fp32 MFMA path, which runs power-limited at ~52 % executed-MFMA utilisation.
//
// Structure (one block = one 4 (z) x 4 (y) x 32 (x) output tile, 256 threads, two blocks per CU):
//   stage : per chunk of 8 input channels, six input planes z0-1 .. z0+4 of the 6 x 34 halo arrive by raw buffer
//           loads (hardware zero fill), are transformed in fp32 exactly as in conv0_w43_mfma (U0 .. U5) and stored in
//           LDS as three bf16 tiles [piece][plane][hy][hx][8 channels] (16 bytes per voxel and piece = one lane's A
//           fragment); chunk c+1's loads are in flight during chunk c's MFMAs
//   MFMA  : Toeplitz-pair form of conv0p16 (conv3d_mfma16.hip): M = 16 x-pairs, N = 2 x-outputs x 8 channels,
//           K = 4 x-taps x 8 channels; the 12 work units (plane t, 2 rows) are dealt three per wave as in
//           conv0_w43_mfma; a unit reads its 4 halo rows of a piece once (12 ds_read_b128) for 36 MFMAs; the wave's
//           18 B fragments (2 planes x 3 ky x 3 pieces) live in registers and are re-loaded per chunk
//   end   : the six m_t meet in LDS, y0 .. y3 (+ bias, ReLU) as in conv0_w43_mfma.
// Selected by MVS_CONV0_SPLIT (csrc/conv3d_direct.hip); the fp32-MFMA kernels stay selectable.
#include <cstring>

#include "mvs_internal.h"
#include "storage.h"

// `make ablate50`: phase clocks (s_memtime) of the four waves of one block, printed to stderr.  Product builds
// leave MVS_ABLATE at 0.
#ifndef MVS_ABLATE
#define MVS_ABLATE 0
#endif

namespace mvs {

namespace c43s {
constexpr int TZ = 4, TY = 4, TX = 32;
constexpr int HY = TY + 2, HX = TX + 2;
constexpr int NT_PLANES = 6;
constexpr int VOX = HY * HX;                    // 204 voxels per halo plane
constexpr int PLANE_E = VOX * 8;                // bf16 elements of one (piece, plane)
constexpr int PIECE_E = NT_PLANES * PLANE_E;    // 9,792
constexpr int TILE_E = 3 * PIECE_E;             // 29,376 elements = 58,752 B
constexpr int NCOL = VOX * 2;                   // z-columns of 4-channel pieces (y, x, half): 408
constexpr int CPT = (NCOL + 255) / 256;         // 2
constexpr int NPOS = TY * TX;                   // 128 (y, x) positions
constexpr int EXS = 12;                         // floats per position in the exchange tile
constexpr int BFRAG = 4 * NT_PLANES * 3 * 3;    // B fragments (64 lanes x 16 B each): [chunk][t][ky][piece]
static_assert(NT_PLANES * NPOS * EXS * 4 <= TILE_E * 2, "exchange tile must fit in the input tile");
}  // namespace c43s

typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int su32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sbf16x4 __attribute__((ext_vector_type(4)));

// v = p1 + p2 + p3 (+ <= 2^-27 |v|), each a bf16 (RNE); the subtractions are exact in fp32
__device__ __forceinline__ void split3(const f32x4 v, su32x2& p1, su32x2& p2, su32x2& p3) {
    const sbf16x4 h1 = __builtin_convertvector(v, sbf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h1, f32x4);
    const sbf16x4 h2 = __builtin_convertvector(r1, sbf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(h2, f32x4);
    const sbf16x4 h3 = __builtin_convertvector(r2, sbf16x4);
    p1 = __builtin_bit_cast(su32x2, h1);
    p2 = __builtin_bit_cast(su32x2, h2);
    p3 = __builtin_bit_cast(su32x2, h3);
}

__device__ __forceinline__ f32x4 mfma_bf16(su32x4 a, su32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sbf16x8, a), __builtin_bit_cast(sbf16x8, b), c, 0, 0, 0);
}

// PAR = wave parity: even waves run units on planes (p, p, p+1), odd waves (p, p+1, p+1)
template <int PAR>
__device__ __forceinline__ void c43s_chunk_mfmas(const unsigned short* __restrict__ tile, const int (&aoff)[3],
                                                 const su32x4 (&B)[2][3][3], f32x4 (&acc)[3][2]) {
    using namespace c43s;
#pragma unroll
    for (int p = 0; p < 3; ++p) {            // A piece
        su32x4 a[3][4];                      // [unit][halo row]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                a[i][j] = *reinterpret_cast<const su32x4*>(tile + p * PIECE_E + aoff[i] + j * HX * 8);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int q = 0; q < 3 - p; ++q)  // B piece: (p, q) in {(0,0),(0,1),(0,2),(1,0),(1,1),(2,0)}
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int pl = i == 0 ? 0 : (i == 2 ? 1 : PAR);
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)   // six independent accumulators back to back
                        acc[i][rr] = mfma_bf16(a[i][rr + ky], B[pl][ky][q], acc[i][rr]);
                }
    }
}

__global__ __launch_bounds__(256, 2) void conv0_w43s_kernel(
    const float* __restrict__ x,             // [4][D][H][W][8] fp32
    const unsigned short* __restrict__ bp,   // [4 chunks][6 t][3 ky][3 pieces][64 lanes][8] bf16 Toeplitz panel
    const float* __restrict__ bias,          // [8]
    float* __restrict__ y,                   // [D][H][W][8] fp32
    int D, int H, int W
#if MVS_ABLATE == 50
    , long long* __restrict__ dbg
#endif
    ) {
    using namespace c43s;
#if MVS_ABLATE == 50
#define MVS_TICK(slot) if (dbg && blockIdx.x == 2000 && (threadIdx.x & 63) == 0) dbg[(threadIdx.x >> 6) * 32 + (slot)] = clock64();
#else
#define MVS_TICK(slot)
#endif
    __shared__ __attribute__((aligned(16))) unsigned short tile[TILE_E];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    // block -> tile: each XCD owns a band of tile rows and walks it x-fastest, then row, then z (see conv0_w43_mfma)
    int bx, by, bz;
    if (nby % 8 == 0 && gridDim.x % 8 == 0) {
        const int xcd = blockIdx.x & 7, rows = nby >> 3;
        int i = blockIdx.x >> 3;
        bx = i % nbx; i /= nbx;
        by = xcd * rows + i % rows;
        bz = i / rows;
    } else {
        int b = blockIdx.x;
        bx = b % nbx; b /= nbx;
        by = b % nby;
        bz = b / nby;
    }
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    const size_t HW8 = (size_t)H * W * 8, V8 = (size_t)D * HW8;

    // staging: a thread owns up to CPT z-columns (hy, hx, half) and stages all six planes of each
    unsigned boff[CPT][NT_PLANES];
    int loff[CPT];            // in 8-byte units inside one (piece, plane)
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int col = tid + i * 256;
        const int half = col & 1, v = col >> 1;
        const int hx = v % HX, hy = v / HX;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = col < NCOL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        loff[i] = col < NCOL ? v * 2 + half : -1;
#pragma unroll
        for (int q = 0; q < NT_PLANES; ++q) {
            const int gz = z0 - 1 + q;
            // a piece outside the volume gets a byte offset beyond the buffer's range: the hardware returns zeros
            boff[i][q] = (ok && gz >= 0 && gz < D) ? (unsigned)(((size_t)gz * HW8 + ((size_t)gy * W + gx) * 8 + half * 4) * 4)
                                                   : 0x80000000u;
        }
    }
    f32x4 stg[CPT][NT_PLANES];
    auto load_chunk = [&](int c) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x + (size_t)c * V8), (short)0, (int)(V8 * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int q = 0; q < NT_PLANES; ++q)
                stg[i][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)boff[i][q], 0, 0));
    };
    auto store_chunk = [&]() {
        su32x2* t8 = reinterpret_cast<su32x2*>(tile);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            if (loff[i] >= 0) {
                const f32x4(&d)[NT_PLANES] = stg[i];
                const f32x4 t1 = d[4] - 4.0f * d[2], t2 = d[3] - 4.0f * d[1];
                const f32x4 t3 = d[4] - d[2], t4 = 2.0f * (d[3] - d[1]);
                f32x4 u[NT_PLANES];
                u[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
                u[1] = t1 + t2;
                u[2] = t1 - t2;
                u[3] = t3 + t4;
                u[4] = t3 - t4;
                u[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
#pragma unroll
                for (int t = 0; t < NT_PLANES; ++t) {
                    su32x2 p1, p2, p3;
                    split3(u[t], p1, p2, p3);
                    su32x2* dst = t8 + t * (PLANE_E / 4) + loff[i];
                    dst[0] = p1;
                    dst[PIECE_E / 4] = p2;
                    dst[2 * (PIECE_E / 4)] = p3;
                }
            }
    };

    // work units of this wave: u = 3 wave + i -> (plane u >> 1, rows 2 (u & 1) .. + 1); lane (r, g): halo x = 2 r + g
    const int r = lane & 15, g = lane >> 4;
    const int u0 = 3 * wave, pA = u0 >> 1;
    int aoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = u0 + i, pl = u >> 1, mg = u & 1;
        aoff[i] = pl * PLANE_E + ((2 * mg) * HX + 2 * r + g) * 8;
    }
    su32x4 B[2][3][3];
    auto load_b = [&](int c) {
        const su32x4* bsrc = reinterpret_cast<const su32x4*>(bp) + (size_t)((c * NT_PLANES + pA) * 9) * 64 + lane;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int q = 0; q < 3; ++q) B[pl][ky][q] = bsrc[((pl * 3 + ky) * 3 + q) * 64];
    };

    f32x4 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    MVS_TICK(0)
    load_chunk(0);
    load_b(0);
    store_chunk();
    MVS_TICK(1)
    __syncthreads();
    MVS_TICK(2)

#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (c < 3) load_chunk(c + 1);
        MVS_TICK(3 + 5 * c)
        if (wave & 1) c43s_chunk_mfmas<1>(tile, aoff, B, acc);
        else c43s_chunk_mfmas<0>(tile, aoff, B, acc);
        MVS_TICK(4 + 5 * c)
        if (c < 3) {
            load_b(c + 1);    // younger than the chunk's activation loads: store_chunk's wait leaves these in flight
            __syncthreads();  // every wave is done reading chunk c's tiles
            MVS_TICK(5 + 5 * c)
            store_chunk();
            MVS_TICK(6 + 5 * c)
            __syncthreads();
            MVS_TICK(7 + 5 * c)
        }
    }

    // exchange: acc[i][rr][e] = (plane, row 2 mg + rr, x = 2 (4 g + e) + jj, channel co), n = lane & 15 = (jj, co)
    __syncthreads();
    float* ex = reinterpret_cast<float*>(tile);
    {
        const int n = lane & 15, jj = n >> 3, co = n & 7;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = u0 + i, pl = u >> 1, mg = u & 1;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    ex[(pl * NPOS + (2 * mg + rr) * TX + 2 * (4 * g + e) + jj) * EXS + co] = acc[i][rr][e];
        }
    }
    __syncthreads();
    const int pos = tid >> 1, ch = tid & 1;  // (y, x) of the 4 x 32 tile, channels 4 ch .. 4 ch + 3
    const int gy = y0 + (pos >> 5), gx = x0 + (pos & 31);
    if (gy >= H || gx >= W) return;
    f32x4 M[NT_PLANES];
#pragma unroll
    for (int q = 0; q < NT_PLANES; ++q) M[q] = *reinterpret_cast<const f32x4*>(ex + (q * NPOS + pos) * EXS + ch * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + ch * 4);
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 s12 = M[1] + M[2], d12 = M[1] - M[2], s34 = M[3] + M[4], d34 = M[3] - M[4];
    f32x4 o[TZ];
    o[0] = (M[0] + s12) + s34;
    o[1] = d12 + 2.0f * d34;
    o[2] = s12 + 4.0f * s34;
    o[3] = (d12 + 8.0f * d34) + M[5];
#pragma unroll
    for (int q = 0; q < TZ; ++q) {
        if (z0 + q >= D) break;
        const f32x4 v = __builtin_elementwise_max(o[q] + bv, zero);
        *reinterpret_cast<f32x4*>(y + (((size_t)(z0 + q) * H + gy) * W + gx) * 8 + ch * 4) = v;
    }
    MVS_TICK(24)
#undef MVS_TICK
}

int launch_conv0_wino43_split(const void* x, void* y, const void* bp, const float* bias, int D, int H, int W,
                              int dtype, hipStream_t s) {
    using namespace c43s;
    if (dtype != MVS_F32) return fail(MVS_ERR_BAD_DTYPE, "conv0_wino43_split: fp32 volumes only (dtype %d)", dtype);
    if ((size_t)D * H * W * 8 * 4 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "conv0_wino43_split: plane of %zu bytes exceeds 31-bit buffer offsets",
                    (size_t)D * H * W * 8 * 4);
    const int nb = ((W + TX - 1) / TX) * ((H + TY - 1) / TY) * ((D + TZ - 1) / TZ);
#if MVS_ABLATE == 50
    static long long* dbg = nullptr;
    if (!dbg && hipMalloc(&dbg, 128 * sizeof(long long)) != hipSuccess) dbg = nullptr;
    if (dbg) (void)hipMemsetAsync(dbg, 0, 128 * sizeof(long long), s);
    conv0_w43s_kernel<<<nb, 256, 0, s>>>(static_cast<const float*>(x), static_cast<const unsigned short*>(bp), bias,
                                         static_cast<float*>(y), D, H, W, dbg);
    if (dbg && nb > 2000) {
        long long h[128];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 4; ++w) {
            const long long* t = h + w * 32;
            fprintf(stderr, "conv0s wave %d: stage0 %lld bar %lld |", w, t[1] - t[0], t[2] - t[1]);
            for (int c = 0; c < 4; ++c)
                fprintf(stderr, " c%d: ld %lld mfma %lld bar %lld st %lld bar %lld |", c, t[3 + 5 * c] - (c ? t[2 + 5 * c] : t[2]),
                        t[4 + 5 * c] - t[3 + 5 * c], c < 3 ? t[5 + 5 * c] - t[4 + 5 * c] : 0, c < 3 ? t[6 + 5 * c] - t[5 + 5 * c] : 0,
                        c < 3 ? t[7 + 5 * c] - t[6 + 5 * c] : 0);
            fprintf(stderr, " end %lld total %lld\n", t[24] - t[19], t[24] - t[0]);
        }
    }
#else
    conv0_w43s_kernel<<<nb, 256, 0, s>>>(static_cast<const float*>(x), static_cast<const unsigned short*>(bp), bias,
                                         static_cast<float*>(y), D, H, W);
#endif
    return check_hip(hipGetLastError(), "conv0_w43s launch");
}

// ---------------------------------------------------------------------------------------------
// host: wfold [27][32][8] (tap = kz*9 + ky*3 + kx) -> bp [4 chunks][6 t][3 ky][3 pieces][64 lanes][8] bf16.
// The z taps are transformed by G of F(4,3) in double and rounded once to fp32 (the values conv0_w43_mfma uses),
// then split into three bf16 pieces.  Lane (n = lane & 15, g = lane >> 4), element j: k = (halo x offset g,
// channel j of the chunk), column n = (x-output jj = n >> 3 of the pair, co = n & 7); tap kx = g - jj.
// ---------------------------------------------------------------------------------------------
static inline uint16_t bf16_rne(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x0040u);  // NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}

void pack_conv0_wino43_split_weights(const float* wfold, void* out) {
    uint16_t* bp = static_cast<uint16_t*>(out);
    for (int c = 0; c < 4; ++c)
        for (int t = 0; t < 6; ++t)
            for (int ky = 0; ky < 3; ++ky)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = lane & 15, g = lane >> 4, jj = n >> 3, co = n & 7, kx = g - jj;
                        float gv = 0.0f;
                        if (kx >= 0 && kx <= 2) {
                            const int tap = ky * 3 + kx, ci = 8 * c + j;
                            const double g0 = wfold[((size_t)(0 * 9 + tap) * 32 + ci) * 8 + co];
                            const double g1 = wfold[((size_t)(1 * 9 + tap) * 32 + ci) * 8 + co];
                            const double g2 = wfold[((size_t)(2 * 9 + tap) * 32 + ci) * 8 + co];
                            double gd;
                            switch (t) {
                                case 0: gd = g0 / 4.0; break;
                                case 1: gd = -(g0 + g1 + g2) / 6.0; break;
                                case 2: gd = -(g0 - g1 + g2) / 6.0; break;
                                case 3: gd = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
                                case 4: gd = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
                                default: gd = g2; break;
                            }
                            gv = (float)gd;
                        }
                        const uint16_t h1 = bf16_rne(gv);
                        const float r1 = gv - bf16_to_f32(h1);
                        const uint16_t h2 = bf16_rne(r1);
                        const float r2 = r1 - bf16_to_f32(h2);
                        const uint16_t h3 = bf16_rne(r2);
                        const uint16_t hs[3] = {h1, h2, h3};
                        for (int q = 0; q < 3; ++q)
                            bp[((((size_t)(c * 6 + t) * 3 + ky) * 3 + q) * 64 + lane) * 8 + j] = hs[q];
                    }
}

}  // namespace mvs
