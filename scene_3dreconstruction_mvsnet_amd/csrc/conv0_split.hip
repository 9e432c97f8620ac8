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
#include <cstdlib>
#include <cstring>
#include <type_traits>

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

// ---------------------------------------------------------------------------------------------
// Second form (default of MVS_CONV0_SPLIT=1): PERSISTENT blocks with PRODUCER and CONSUMER waves.
//
// Phase clocks of the first form (make ablate50): a tile takes 40.7 k cycles of which a wave's 432 MFMAs are 6.9 k
// -- per chunk: 3.0 k MFMA phase, 1.2 k barrier, 2.2-5.8 k wait-for-loads + transform + split, 8 k cold start per
// tile.  The bf16 matrix pipe and the vector ALUs DO overlap (unlike the fp32 MFMA), so the two kinds of work
// belong on different waves that run at the same time:
//   * one block per CU, 512 threads, looping over its tiles (tile index = block + j * grid: a tile keeps its XCD);
//   * waves 4-7 PRODUCE: raw buffer loads two chunk-steps ahead (two register sets), Winograd transform in fp32,
//     three-way bf16 split (18 VALU per 4 values: v_cvt_pk_bf16_f32, shift / mask, v_pk_add_f32), ds_write_b64
//     into the LDS buffer of the NEXT step (two 58.7 KB buffers); at a tile's first step they also run the output
//     transform + stores of the tile the consumers have just finished (own 36.9 KB exchange tile);
//   * waves 0-3 CONSUME: 108 MFMAs per step on the current buffer, A fragments read one (piece, unit) ahead, the
//     next step's 18 B fragments requested at the start of the step into a second register set;
//   * ONE block barrier per chunk-step; nothing of a tile's prologue or epilogue is exposed except for a block's
//     first and last tile.
// ---------------------------------------------------------------------------------------------
constexpr int c43p_MAXT = 512;   // tiles per block the origin table holds (the launcher sizes the grid accordingly)
typedef float sf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sbf16x2 __attribute__((ext_vector_type(2)));

// packed = (bf16(a.x), bf16(a.y)) RNE; returns a - widen(packed) (exact).  (Tried: the residual as one
// v_dot2c_f32_bf16 per value, D += h . (-1, 0) -- 14 instead of 18 instructions per 4 values.  The build was not faster
// (0.283 vs 0.276-0.283 ms) and its results were WRONG (heavy-tailed test 7.5e4 x its bound): the packed-bf16 inline
// constant hipcc emits for (-1, 0) is not what the instruction reads.  Dropped.)
__device__ __forceinline__ sf32x2 split_stage(const sf32x2 a, unsigned& packed) {
    const sbf16x2 h = __builtin_convertvector(a, sbf16x2);
    packed = __builtin_bit_cast(unsigned, h);
    const sf32x2 w = {__uint_as_float(packed << 16), __uint_as_float(packed & 0xFFFF0000u)};
    return a - w;
}
__device__ __forceinline__ void split3x(const f32x4 v, su32x2& p1, su32x2& p2, su32x2& p3) {
    unsigned a0, a1, b0, b1, c0, c1;
    const sf32x2 r0 = split_stage((sf32x2){v.x, v.y}, a0), r1 = split_stage((sf32x2){v.z, v.w}, a1);
    const sf32x2 q0 = split_stage(r0, b0), q1 = split_stage(r1, b1);
    c0 = __builtin_bit_cast(unsigned, __builtin_convertvector(q0, sbf16x2));
    c1 = __builtin_bit_cast(unsigned, __builtin_convertvector(q1, sbf16x2));
    p1 = (su32x2){a0, a1};
    p2 = (su32x2){b0, b1};
    p3 = (su32x2){c0, c1};
}

// One chunk-step of a consumer wave: 9 (A piece p, unit i) combos; a combo reads its 4 A fragments (halo rows) once for
// 6 (3 - p) MFMAs on two alternating accumulators.  The A fragments are requested TWO combos ahead into a ring of
// three register sets: with a distance of one the ds_read latency under load (~250 cycles against the 96-288 cycles
// of a combo's MFMAs) was exposed at every combo -- 4.3 k cycles per step for 1.7 k of MFMAs.  The scheduler is held to
// this order (sched_barrier): left alone it hoists every ds_read of the unrolled step and spills.
template <int PAR>
__device__ __forceinline__ void c43p_step_mfmas(const unsigned short* __restrict__ tile, const int (&aoff)[3],
                                                const su32x4 (&B)[2][3][3], su32x4 (&Bn)[2][3][3],
                                                __amdgpu_buffer_rsrc_t brs, unsigned bvoff, unsigned bnext,
                                                f32x4 (&acc)[3][2]) {
    using namespace c43s;
    constexpr int NC = 9;
    su32x4 a[3][4];
    auto request = [&](int n) {
        const int pn = n / 3, in = n % 3;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            a[n % 3][j] = *reinterpret_cast<const su32x4*>(tile + pn * PIECE_E + aoff[in] + j * HX * 8);
    };
    request(0);
    request(1);
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        const int p = n / 3, i = n % 3;
        if (n + 2 < NC) request(n + 2);
        // two of the NEXT step's 18 B fragments per combo: issued as one burst at the start of the step the 18 loads
        // took 1.0-2.3 k cycles to get into the CU's vector-memory queue (64 B/clk: a fragment is 16 cycles, and the
        // producers' activation loads share the queue) with the MFMAs waiting behind them
        {
            const int f0 = 2 * n, f1 = 2 * n + 1;
            Bn[f0 / 9][(f0 / 3) % 3][f0 % 3] = __builtin_bit_cast(su32x4, __builtin_amdgcn_raw_buffer_load_b128(
                brs, (int)bvoff, (int)(bnext + (unsigned)f0 * 1024u), 0));
            Bn[f1 / 9][(f1 / 3) % 3][f1 % 3] = __builtin_bit_cast(su32x4, __builtin_amdgcn_raw_buffer_load_b128(
                brs, (int)bvoff, (int)(bnext + (unsigned)f1 * 1024u), 0));
        }
        const int pl = i == 0 ? 0 : (i == 2 ? 1 : PAR);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int q = 0; q < 3 - p; ++q)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
                    acc[i][rr] = mfma_bf16(a[n % 3][rr + ky], B[pl][ky][q], acc[i][rr]);
#if MVS_ABLATE != 58
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

__global__ __launch_bounds__(512) void conv0_w43p_kernel(
    const float* __restrict__ x,             // [4][D][H][W][8] fp32
    const unsigned short* __restrict__ bp,   // [4 chunks][6 t][3 ky][3 pieces][64 lanes][8] bf16 Toeplitz panel
    const float* __restrict__ bias,          // [8]
    float* __restrict__ y,                   // [D][H][W][8] fp32
    int D, int H, int W, int nb
#if MVS_ABLATE == 55
    , long long* __restrict__ dbg
#endif
    ) {
    using namespace c43s;
#if MVS_ABLATE == 55   // s_memtime stamps of one consumer and one producer wave of block 100 during its tile 5
#define MVS_PTICK(k, slot) if (dbg && blockIdx.x == 100 && (k) == 5 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 3) == 0) dbg[(threadIdx.x >> 8) * 64 + (slot)] = clock64();
#else
#define MVS_PTICK(k, slot)
#endif
    __shared__ __attribute__((aligned(16))) unsigned short buf[2 * TILE_E];
    __shared__ __attribute__((aligned(16))) float ex[NT_PLANES * NPOS * EXS];
    __shared__ int org[c43p_MAXT][4];   // origins of this block's tiles (three runtime divisions each: once, not per step)

#if MVS_ABLATE == 75   // consumers on two SIMDs, producers on the other two (waves w and w + 4 share a SIMD): physical waves
                       // {0, 1, 4, 5} take the consumer roles 0..3, {2, 3, 6, 7} the producer roles 4..7
    const int lane = threadIdx.x & 63, pw_ = threadIdx.x >> 6;
    const int wave = ((pw_ & 2) << 1) | ((pw_ & 4) >> 1) | (pw_ & 1), tid = wave * 64 + lane;
#else
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#endif
    // (producers on the older wave half instead -- waves 0-3 win the issue arbitration -- measured the same: 0.2744 /
    // 0.2750 vs 0.2735 / 0.2767 ms; `make ablate75`, consumers and producers on SEPARATE SIMD pairs: 0.284 vs 0.278 ms)
    const bool consumer = __builtin_amdgcn_readfirstlane(wave) < 4;   // wave-uniform by construction; tell the compiler
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    const int G = gridDim.x;
    const int ntile = (nb - (int)blockIdx.x + G - 1) / G;   // this block's tiles: blockIdx.x + j * G
    const size_t HW8 = (size_t)H * W * 8, V8 = (size_t)D * HW8;
    // tile index -> origin; every XCD owns a band of tile rows and walks it x-fastest, then row, then z
    // (conv0_w43_mfma); G is a multiple of 8 whenever nb is, so a block's tiles stay on its XCD's band
    auto origin = [&](int idx, int& x0, int& y0, int& z0) {
        int bx, by, bz;
        if (nby % 8 == 0 && G % 8 == 0) {
            const int xcd = idx & 7, rows = nby >> 3;
            int i = idx >> 3;
            bx = i % nbx; i /= nbx;
            by = xcd * rows + i % rows;
            bz = i / rows;
        } else {
            int b = idx;
            bx = b % nbx; b /= nbx;
            by = b % nby;
            bz = b / nby;
        }
        x0 = bx * TX; y0 = by * TY; z0 = bz * TZ;
    };

    for (int j = tid; j < ntile; j += 512) {
        int x0, y0, z0;
        origin((int)blockIdx.x + j * G, x0, y0, z0);
        org[j][0] = x0; org[j][1] = y0; org[j][2] = z0;
    }
    __syncthreads();
    // ---------------- producer state ----------------
    const int ptid = tid & 255;
    unsigned boff[CPT][NT_PLANES];
    int loff[CPT];            // in 8-byte units inside one (piece, plane)
    int chy[CPT], chx[CPT], crel[CPT];   // the column's halo row / column and its float offset from the tile origin
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        // threads beyond the 408 columns shadow the last column (same loads, same values to the same LDS address):
        // no branch inside the staging code
        const int col = min(ptid + i * 256, NCOL - 1);
        loff[i] = col;    // (v * 2 + half) = col
        const int half = col & 1, v = col >> 1;
        chx[i] = v % HX - 1;
        chy[i] = v / HX - 1;
        crel[i] = (chy[i] * W + chx[i]) * 8 + half * 4;
    }
    // per tile: 32-bit arithmetic only (the launcher guarantees < 2^31 bytes per chunk plane); the first form of this
    // function (64-bit products, two divisions per column) took 1.4 k cycles of the producers' step
    auto set_tile = [&](int j) {   // j-th tile of this block
        const int x0 = org[j][0], y0 = org[j][1], z0 = org[j][2];
        const int hw8 = (int)HW8, tbase = (y0 * W + x0) * 8;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int gy = y0 + chy[i], gx = x0 + chx[i];
            const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const int cb = tbase + crel[i];
#pragma unroll
            for (int q = 0; q < NT_PLANES; ++q) {
                const int gz = z0 - 1 + q;   // wave-uniform
                boff[i][q] = (ok && (unsigned)gz < (unsigned)D) ? (unsigned)(gz * hw8 + cb) * 4u
                                                                  : 0x80000000u;   // beyond the descriptor: zeros
            }
        }
    };
    f32x4 stg[2][CPT][NT_PLANES];
    auto issue_loads = [&](int c, f32x4 (&st)[CPT][NT_PLANES]) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x + (size_t)c * V8), (short)0, (int)(V8 * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int q = 0; q < NT_PLANES; ++q)
                st[i][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)boff[i][q], 0, 0));
    };
    // transform + split of one staged chunk (`st`) into `dstbuf`; LOAD = true: the raw loads of chunk `cn` (two steps
    // ahead) into `sn` are issued one per (column, plane) segment in between -- as one burst the 12 loads took
    // 1.8-2.5 k cycles to enter the CU's vector-memory queue, with the whole transform waiting behind them
    // (the staging sets are addressed by compile-time index, never passed as references: a reference parameter next
    // to the by-reference capture made hipcc keep `stg` in scratch memory)
    auto process = [&](auto set_tag, unsigned short* dstbuf, auto load_tag, int cn) {
        constexpr int SET = decltype(set_tag)::value;      // staged chunk: stg[SET]; loads go to stg[SET ^ 1]
        constexpr bool LOAD = decltype(load_tag)::value;
        su32x2* t8 = reinterpret_cast<su32x2*>(dstbuf);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x + (size_t)cn * V8), (short)0, (int)(V8 * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            f32x4 d[NT_PLANES];
#pragma unroll
            for (int q = 0; q < NT_PLANES; ++q) d[q] = stg[SET][i][q];
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
                if constexpr (LOAD)
                    stg[SET ^ 1][i][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)boff[i][t], 0, 0));
                su32x2 p1, p2, p3;
#if MVS_ABLATE == 74   // diagnostic: no split arithmetic
                p1 = p2 = p3 = (su32x2){__float_as_uint(u[t].x), __float_as_uint(u[t].z)};
#else
                split3x(u[t], p1, p2, p3);
#endif
                su32x2* dst = t8 + t * (PLANE_E / 4) + loff[i];
                dst[0] = p1;
                dst[PIECE_E / 4] = p2;
                dst[2 * (PIECE_E / 4)] = p3;
                // one plane at a time, in this order (also: interleaving the six splits of both columns for ILP
                // costs more registers than the two staging sets leave)
#if MVS_ABLATE != 56
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
    };
    // output transform + stores of a finished tile: thread -> (position, 4 channels).  Raw buffer stores: `live` =
    // false (a block's first step has no finished tile) or a position outside the volume puts the offset beyond the
    // descriptor and the hardware drops the store -- no branch around a VMEM instruction, so hipcc keeps COUNTED
    // vmcnt waits for the prefetched activations (a conditional store or load makes it wait for the loads just issued)
    const f32x4 obias = *reinterpret_cast<const f32x4*>(bias + (ptid & 1) * 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, (int)(V8 * 4), 0x00020000);
    auto output = [&](int j, bool live) {   // j-th tile of this block
        const int x0 = org[j][0], y0 = org[j][1], z0 = org[j][2];
        const int pos = ptid >> 1, ch = ptid & 1;
        const int gy = y0 + (pos >> 5), gx = x0 + (pos & 31);
        const bool ok = live && gy < H && gx < W;
        f32x4 M[NT_PLANES];
#pragma unroll
        for (int q = 0; q < NT_PLANES; ++q) M[q] = *reinterpret_cast<const f32x4*>(ex + (q * NPOS + pos) * EXS + ch * 4);
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 s12 = M[1] + M[2], d12 = M[1] - M[2], s34 = M[3] + M[4], d34 = M[3] - M[4];
        f32x4 o[TZ];
        o[0] = (M[0] + s12) + s34;
        o[1] = d12 + 2.0f * d34;
        o[2] = s12 + 4.0f * s34;
        o[3] = (d12 + 8.0f * d34) + M[5];
        const unsigned base = (unsigned)((((size_t)z0 * H + gy) * W + gx) * 8 + ch * 4) * 4u;
#pragma unroll
        for (int q = 0; q < TZ; ++q) {
            const f32x4 v = __builtin_elementwise_max(o[q] + obias, zero);
            const unsigned off = (ok && z0 + q < D) ? base + (unsigned)q * (unsigned)(HW8 * 4) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), yrs, (int)off, 0, 0);
        }
    };

    // ---------------- consumer state ----------------
    // work units of a consumer wave: u = 3 wave + i -> (plane u >> 1, rows 2 (u & 1) .. + 1); lane (r, g): halo x = 2 r + g
    const int r = lane & 15, g = lane >> 4;
    const int cw = wave & 3, u0 = 3 * cw, pA = u0 >> 1;
    int aoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = u0 + i, pl = u >> 1, mg = u & 1;
        aoff[i] = pl * PLANE_E + ((2 * mg) * HX + 2 * r + g) * 8;
    }
    // B fragments by raw buffer loads: ONE vector offset (the lane's 16 bytes + the wave's first plane), the fragment
    // in the scalar offset -- plain pointers made hipcc materialise (and spill) 35 64-bit addresses
    su32x4 B[2][2][3][3];   // two sets: the next step's 18 fragments are requested at the start of a step
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(bp), (short)0, 4 * NT_PLANES * 9 * 1024, 0x00020000);
    const unsigned bvoff = (unsigned)lane * 16u + (unsigned)pA * 9u * 1024u;
    auto load_b = [&](int c, su32x4 (&Bd)[2][3][3]) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    Bd[pl][ky][q] = __builtin_bit_cast(su32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        brs, (int)bvoff, (int)((unsigned)((c * NT_PLANES * 3 + pl * 3 + ky) * 3 + q) * 1024u), 0));
    };
    f32x4 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // acc[i][rr][e] = (plane, row 2 mg + rr, x = 2 (4 g + e) + jj, channel co), n = lane & 15 = (jj, co)
    auto exchange = [&]() {
        const int n = lane & 15, jj = n >> 3, co = n & 7;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = u0 + i, pl = u >> 1, mg = u & 1;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ex[(pl * NPOS + (2 * mg + rr) * TX + 2 * (4 * g + e) + jj) * EXS + co] = acc[i][rr][e];
                    acc[i][rr][e] = 0.0f;
                }
        }
    };

    // The two roles run DISJOINT loops (each with the same number of block barriers: one per chunk-step plus the one
    // after the prologue), so that the register allocator overlays the producers' staging sets with the consumers'
    // B fragments and accumulators instead of keeping both alive.  A loop iteration is one tile = four chunk-steps
    // (c is a compile-time constant, PB = c & 1 = LDS buffer / register set of the step).  EVERY vector-memory
    // instruction is unconditional -- the loads beyond a block's last step re-read its last tile, the B fragments
    // wrap around -- because one branch around a VMEM instruction makes hipcc wait for the loads it has just issued
    // (first build: `if (s + 2 < S) issue_loads` -> s_waitcnt vmcnt(8) / (1) / (0) right after the 12 new loads:
    // every step paid a full L2-miss latency, 6.0 k cycles per step with no unit busier than 40 %).
    if (consumer) {
        int kk = 0;
        auto cstep = [&](auto c_tag) {
            constexpr int c = decltype(c_tag)::value, PB = c & 1;
            MVS_PTICK(kk, 4 * c + 0)
            // the tile finished a step ago: output transform + stores (the consumers have the slack: ~1.7 k cycles of
            // every step they wait for the producers)
            if (c == 1) output(kk > 0 ? kk - 1 : 0, kk > 0);   // (c == 1: the step where the producers are slowest)
            constexpr unsigned bnext = (unsigned)(((c + 1) & 3) * NT_PLANES * 9) * 1024u;   // next step's chunk (wraps around)
            MVS_PTICK(kk, 4 * c + 1)
#if MVS_ABLATE == 71   // diagnostic: no MFMA phase
            if (D < 0)
#endif
            {
            if (cw & 1) c43p_step_mfmas<1>(buf + PB * TILE_E, aoff, B[PB], B[PB ^ 1], brs, bvoff, bnext, acc);
            else c43p_step_mfmas<0>(buf + PB * TILE_E, aoff, B[PB], B[PB ^ 1], brs, bvoff, bnext, acc);
            }
            MVS_PTICK(kk, 4 * c + 2)
            if (c == 3) exchange();   // the producers read it during the next step (after the barrier)
            MVS_PTICK(kk, 4 * c + 3)
        };
        load_b(0, B[0]);
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < ntile; ++k) {
            kk = k;
            cstep(std::integral_constant<int, 0>{});
            __syncthreads();
            cstep(std::integral_constant<int, 1>{});
            __syncthreads();
            cstep(std::integral_constant<int, 2>{});
            __syncthreads();
            cstep(std::integral_constant<int, 3>{});
            __syncthreads();
            MVS_PTICK(k, 16)
        }
        output(ntile - 1, true);
    } else {
#if MVS_ABLATE != 57
        __builtin_amdgcn_s_setprio(2);   // the producers are the critical path: their VALU wins the issue arbitration
#endif
        // step (k, c): loads of step + 2, output of tile k - 1 (c == 0), transform + split of step + 1
        auto pstep = [&](int k, auto c_tag) {
            constexpr int c = decltype(c_tag)::value, PB = c & 1;
            MVS_PTICK(k, 4 * c + 0)
            if (c == 2 && k + 1 < ntile) set_tile(k + 1);   // the loads below start the next tile
            MVS_PTICK(k, 4 * c + 1)
            MVS_PTICK(k, 4 * c + 2)
            // (after the last step: into a buffer nobody reads, loads that re-read the last tile)
            process(std::integral_constant<int, PB ^ 1>{}, buf + (PB ^ 1) * TILE_E, std::true_type{}, (c + 2) & 3);
            MVS_PTICK(k, 4 * c + 3)
        };
        // prologue: step 0 staged, step 1 in flight
        set_tile(0);
        issue_loads(0, stg[0]);
        issue_loads(1, stg[1]);
        process(std::integral_constant<int, 0>{}, buf, std::false_type{}, 0);
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < ntile; ++k) {
            pstep(k, std::integral_constant<int, 0>{});
            __syncthreads();
            pstep(k, std::integral_constant<int, 1>{});
            __syncthreads();
            pstep(k, std::integral_constant<int, 2>{});
            __syncthreads();
            pstep(k, std::integral_constant<int, 3>{});
            __syncthreads();
            MVS_PTICK(k, 16)
        }
    }
}

int launch_conv0_wino43_split(const void* x, void* y, const void* bp, const float* bias, int D, int H, int W,
                              int dtype, hipStream_t s) {
    using namespace c43s;
    if (dtype != MVS_F32) return fail(MVS_ERR_BAD_DTYPE, "conv0_wino43_split: fp32 volumes only (dtype %d)", dtype);
    if ((size_t)D * H * W * 8 * 4 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "conv0_wino43_split: plane of %zu bytes exceeds 31-bit buffer offsets",
                    (size_t)D * H * W * 8 * 4);
    const int nb = ((W + TX - 1) / TX) * ((H + TY - 1) / TY) * ((D + TZ - 1) / TZ);
    // MVS_CONV0_SPLIT=2: the first form (one tile per 4-wave block); default: persistent producer / consumer blocks
    static const bool form1 = [] {
        const char* e = getenv("MVS_CONV0_SPLIT");
        return e && e[0] == '2';
    }();

    if (!form1) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        int grid = nb < cus ? nb : cus;
        if ((nb + grid - 1) / grid > c43p_MAXT) grid = (nb + c43p_MAXT - 1) / c43p_MAXT;   // (> 130 k tiles: never at 31-bit volumes)
#if MVS_ABLATE == 55
        static long long* dbgp = nullptr;
        if (!dbgp && hipMalloc(&dbgp, 128 * sizeof(long long)) != hipSuccess) dbgp = nullptr;
        if (dbgp) (void)hipMemsetAsync(dbgp, 0, 128 * sizeof(long long), s);
        conv0_w43p_kernel<<<grid, 512, 0, s>>>(static_cast<const float*>(x), static_cast<const unsigned short*>(bp), bias,
                                               static_cast<float*>(y), D, H, W, nb, dbgp);
        if (dbgp && grid > 100) {
            long long h[128];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, dbgp, sizeof(h), hipMemcpyDeviceToHost);
            for (int role = 0; role < 2; ++role) {
                const long long* t = h + role * 64;
                fprintf(stderr, "conv0p %s:", role ? "producer (loads | output | process | barrier)" : "consumer (B loads | mfma | exchange | barrier)");
                for (int c = 0; c < 4; ++c)
                    fprintf(stderr, " c%d: %lld %lld %lld %lld |", c, t[4 * c + 1] - t[4 * c], t[4 * c + 2] - t[4 * c + 1],
                            t[4 * c + 3] - t[4 * c + 2], (c < 3 ? t[4 * c + 4] : t[16]) - t[4 * c + 3]);
                fprintf(stderr, " tile %lld\n", t[16] - t[0]);
            }
        }
#else
        conv0_w43p_kernel<<<grid, 512, 0, s>>>(static_cast<const float*>(x), static_cast<const unsigned short*>(bp), bias,
                                               static_cast<float*>(y), D, H, W, nb);
#endif
        return check_hip(hipGetLastError(), "conv0_w43p launch");
    }
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
