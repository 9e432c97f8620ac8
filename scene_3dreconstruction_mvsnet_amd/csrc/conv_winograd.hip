// conv_winograd.hip -- Winograd-along-z convolutions on the fp32 MFMA: conv0 with F(4,3) (below) and the stride-1
// layers conv2 / conv4 with F(2,3) (second half of the file; the F(2,3) form of conv0 -- the transformed planes
// U0 = d0 - d2, U1 = d1 + d2, U2 = d2 - d1, U3 = d1 - d3 take the place of the four halo planes of a two-plane tile,
// G = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2), y0 = m0 + m1 + m2, y1 = m1 - m2 - m3 -- is in attic/conv0_wz_f23.hip).
//
// conv0 (32 -> 8 channels, 68 % of the path's FLOPs; reference models/mvsnet.py:36,
// block models/module.py:26-33) with Winograd F(4,3) along z on the 4x4x1 fp32 MFMA.
//
// conv0 is bound by the matrix pipe AND by the chip's power management (the same kernel runs 17 % faster
// on an all-zero volume: the clock gives way under fp32 MFMA load), so the lever is fewer multiplications
// and fewer bytes moved, not a tighter issue stream.  F(2,3) along z (attic/conv0_wz_f23.hip) issues 2/3 of
// the direct form's MFMAs and re-reads every input plane twice (4 halo planes per 2 output planes).
// F(4,3) computes FOUR output planes from SIX transformed planes:
//     U0 = 4 d0 - 5 d2 + d4            U5 = 4 d1 - 5 d3 + d5
//     U1 = (d4 - 4 d2) + (d3 - 4 d1)   U2 = (d4 - 4 d2) - (d3 - 4 d1)
//     U3 = (d4 - d2) + 2 (d3 - d1)     U4 = (d4 - d2) - 2 (d3 - d1)
//     G0 = g0/4   G1 = -(g0+g1+g2)/6   G2 = -(g0-g1+g2)/6   G3 = g0/24 + g1/12 + g2/6
//     G4 = g0/24 - g1/12 + g2/6        G5 = g2                        m_t = conv2d(U_t, G_t)
//     y0 = m0 + m1 + m2 + m3 + m4      y1 = (m1 - m2) + 2 (m3 - m4)
//     y2 = (m1 + m2) + 4 (m3 + m4)     y3 = (m1 - m2) + 8 (m3 - m4) + m5
// i.e. 6 plane-convolutions per 4 output planes: 1/2 of the direct form's MFMAs (3/4 of F(2,3)'s) and
// 6 halo planes per 4 outputs (3/4 of F(2,3)'s L2 -> LDS traffic).  As with F(2,3) the transformed
// planes simply take the place of the halo planes in LDS; the weights are transformed on the host.
//
//   tile  : 4 (z) x 4 (y) x 32 (x) outputs; LDS: six transformed 6 x 34 halo planes of one 8-channel
//           chunk (48-byte voxel stride, conflict-free ds_read_b128) + the chunk's weights = 72.6 KB,
//           two blocks per CU
//   block : 256 threads = 4 waves; the 12 work units (plane t, M-group of 2 rows x 32 x) are dealt three
//           per wave, so every SIMD carries the same MFMA load; per (tap, half): 3 A reads + 4 broadcast
//           B reads feed 24 MFMAs (v_mfma_f32_4x4x1_16b_f32, N = 4 channels: Cout = 8 is two exact tiles)
//   K     : 4 chunks of 8 input channels (C8 planes), staged + transformed per chunk, chunk c+1's loads
//           in flight during chunk c's MFMAs
//   end   : the six m_t go through LDS once per tile; a thread then owns (position, 4 channels), forms
//           y0..y3 (+bias, ReLU) and writes four 16-byte pieces
// The sums are re-associated and the transform constants are not powers of two, so the result differs
// from an fmaf chain by a few 1e-7 relative (tests bound it per layer against the oracle).
#include "mvs_internal.h"

// `make ablate49`: phase clocks (s_memtime) of the four waves of one block, printed to stderr
// (profiles/r02_conv0_clocks.txt).  Product builds leave MVS_ABLATE at 0.
#ifndef MVS_ABLATE
#define MVS_ABLATE 0
#endif
#include "storage.h"

namespace mvs {

namespace c43 {
constexpr int TZ = 4, TY = 4, TX = 32;
constexpr int HY = TY + 2, HX = TX + 2;
constexpr int NT_PLANES = 6;
constexpr int VS = 12;                          // floats per voxel in LDS (8 data + 4 pad)
constexpr int PLANE = HY * HX * VS;             // 2448 floats
constexpr int TILE_FLOATS = NT_PLANES * PLANE;  // 58,752 B
constexpr int NCOL = HY * HX * 2;               // z-columns of 16-byte pieces (y, x, half): 408
constexpr int CPT = (NCOL + 255) / 256;         // 2
constexpr int WPLANE = 9 * 2 * 2 * 16;          // weight floats per transformed plane per chunk
constexpr int BW_FLOATS = NT_PLANES * WPLANE;   // per chunk [t][tap][half][nt][j][k]: 3456 floats
constexpr int WPIECES = BW_FLOATS / 4;          // 864 16-byte pieces
constexpr int WPT = (WPIECES + 255) / 256;      // 4
constexpr int NPOS = TY * TX;                   // 128 (y, x) positions
constexpr int EXS = 12;                         // floats per position in the exchange tile
static_assert(NT_PLANES * NPOS * EXS <= TILE_FLOATS, "exchange tile must fit in the input tile");
static_assert(PLANE % 4 == 0, "planes are addressed in 16-byte units");
}  // namespace c43

// PAR = wave parity: which of the wave's three units share a plane (even waves: units on planes
// p, p, p+1; odd waves: p, p+1, p+1) -- a template so that the B-operand registers are picked at
// compile time
template <int PAR>
__device__ __forceinline__ void c43_chunk_mfmas(const float* __restrict__ a0, const float* __restrict__ a1,
                                                const float* __restrict__ a2, const float* __restrict__ wA,
                                                const float* __restrict__ wB, f32x4 (&acc)[3][2]) {
    using namespace c43;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int koff = ((tap / 3) * HX + tap % 3) * VS;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(a0 + koff + half * 4);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(a1 + koff + half * 4);
            const f32x4 x2 = *reinterpret_cast<const f32x4*>(a2 + koff + half * 4);
            const int wo = (tap * 2 + half) * 2 * 16;
            const f32x4 bA0 = *reinterpret_cast<const f32x4*>(wA + wo);
            const f32x4 bA1 = *reinterpret_cast<const f32x4*>(wA + wo + 16);
            const f32x4 bB0 = *reinterpret_cast<const f32x4*>(wB + wo);
            const f32x4 bB1 = *reinterpret_cast<const f32x4*>(wB + wo + 16);
            const f32x4 m0 = PAR ? bB0 : bA0, m1 = PAR ? bB1 : bA1;   // unit 1's plane
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0[k], bA0[k], acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1[k], m0[k], acc[1][0], 0, 0, 0);
                acc[2][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x2[k], bB0[k], acc[2][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0[k], bA1[k], acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1[k], m1[k], acc[1][1], 0, 0, 0);
                acc[2][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x2[k], bB1[k], acc[2][1], 0, 0, 0);
            }
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256, 2) void conv0_w43_mfma_kernel(
    const void* __restrict__ x,      // [4][D][H][W][8] storage dtype DT
    const float* __restrict__ bw,    // [4 chunks][6 t][9 taps][2 halves][2 nt][4 j][4 k]
    const float* __restrict__ bias,  // [8]
    void* __restrict__ y,            // [D][H][W][8] storage dtype DT
    int D, int H, int W
#if MVS_ABLATE == 49
    , long long* __restrict__ dbg
#endif
    ) {
    using namespace c43;
#if MVS_ABLATE == 49   // phase clocks of the four waves of one block (diagnostic build)
#define MVS_TICK(slot) if (dbg && blockIdx.x == 2000 && (threadIdx.x & 63) == 0) dbg[(threadIdx.x >> 6) * 32 + (slot)] = clock64();
#else
#define MVS_TICK(slot)
#endif
    __shared__ __attribute__((aligned(16))) float tile[TILE_FLOATS + BW_FLOATS];
    float* wlds = tile + TILE_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    // Block -> tile.  Blocks are dealt round-robin over the 8 XCDs (b % 8 names the XCD: speed only,
    // never correctness) and every XCD has its own L2, so halo rows / planes shared by neighbouring
    // tiles are only re-read from L2 when the neighbours run on the SAME XCD, close in time.  Each XCD
    // therefore owns a band of nby/8 tile rows and walks it x-fastest, then row, then z: the XCD's ~64
    // resident blocks form slabs of a few z tiles over the whole band, whose y / x halos are shared while
    // they are in L2 and whose z halo (two of six planes) is re-read by the slab that follows.  Measured
    // FETCH_SIZE per launch: 281 MB against 424 MB for z-fastest (328 MB for row-fastest, then z): with
    // the gfx950 correction 0.69 GB = 1.10x the algorithmic 629 MB (was 1.58x); the kernel's time does not
    // change (0.380 / 0.379 / 0.376 ms minimum in the same session) -- it is not HBM-bound.
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

    // staging bookkeeping: a thread owns up to CPT z-columns (y, x, half) and stages all six planes of
    // each.  Loads are unconditional (masked afterwards) so that the compiler keeps counted waits.
    int goff[CPT], loff[CPT];
    unsigned okxy = 0;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int col = tid + i * 256;
        const int half = col & 1, v = col >> 1;
        const int hx = v % HX, hy = v / HX;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = col < NCOL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? (int)(((size_t)gy * W + gx) * 8 + half * 4) : 0;
        okxy |= ok ? (1u << i) : 0u;
        loff[i] = (col < NCOL) ? (v * VS + half * 4) / 4 : -1;   // in 16-byte units
    }
    size_t zoff[NT_PLANES];
    bool zok[NT_PLANES];
#pragma unroll
    for (int q = 0; q < NT_PLANES; ++q) {
        const int gz = z0 - 1 + q;
        zok[q] = gz >= 0 && gz < D;
        zoff[q] = zok[q] ? (size_t)gz * HW8 : 0;
    }

    // work units of this wave: u = 3 wave + i -> (plane u >> 1, M-group u & 1); lane -> (row, xl) of the
    // M-group's 2 rows x 32 x
    const int row = lane >> 5, xl = lane & 31;
    const int u0 = 3 * wave;
    const float* ab[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = u0 + i, pl = u >> 1, mg = u & 1;
        ab[i] = tile + pl * PLANE + ((2 * mg + row) * HX + xl) * VS;
    }
    const int pA = u0 >> 1;   // the wave's two planes: pA, pA + 1
    const float* wA = wlds + pA * WPLANE + (lane & 3) * 4;
    const float* wB = wA + WPLANE;

    f32x4 acc[3][2];  // [unit][nt]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 stg[CPT][NT_PLANES];
    f32x4 wst[WPT];
    // fp32 storage: raw buffer loads -- a piece outside the volume gets a byte offset beyond the
    // buffer's range and the hardware returns zeros, so the transform needs no select instructions
    unsigned boff[CPT][NT_PLANES];
    if constexpr (DT == MVS_F32) {
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int q = 0; q < NT_PLANES; ++q)
                boff[i][q] = (((okxy >> i) & 1u) && zok[q]) ? (unsigned)((zoff[q] + (size_t)goff[i]) * 4) : 0x80000000u;
    }
    auto load_chunk = [&](int c) {
        const size_t plane = (size_t)c * V8;
        if constexpr (DT == MVS_F32) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(static_cast<const float*>(x) + plane), (short)0, (int)(V8 * 4), 0x00020000);
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int q = 0; q < NT_PLANES; ++q)
                    stg[i][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)boff[i][q], 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int q = 0; q < NT_PLANES; ++q) stg[i][q] = St<DT>::load4(x, plane + zoff[q] + goff[i]);
        }
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(bw) + (size_t)c * WPIECES;
#pragma unroll
        for (int i = 0; i < WPT; ++i) wst[i] = wsrc[min(tid + i * 256, WPIECES - 1)];
    };
    auto store_chunk = [&]() {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            if (loff[i] >= 0) {
                const bool in = (okxy >> i) & 1u;
                constexpr bool hw0 = DT == MVS_F32;  // zeros already delivered by the buffer loads
                f32x4 d[NT_PLANES];
#pragma unroll
                for (int q = 0; q < NT_PLANES; ++q) d[q] = (hw0 || (in && zok[q])) ? stg[i][q] : zero;
                const f32x4 t1 = d[4] - 4.0f * d[2], t2 = d[3] - 4.0f * d[1];
                const f32x4 t3 = d[4] - d[2], t4 = 2.0f * (d[3] - d[1]);
                f32x4* t4p = reinterpret_cast<f32x4*>(tile) + loff[i];
                t4p[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
                t4p[PLANE / 4] = t1 + t2;
                t4p[2 * (PLANE / 4)] = t1 - t2;
                t4p[3 * (PLANE / 4)] = t3 + t4;
                t4p[4 * (PLANE / 4)] = t3 - t4;
                t4p[5 * (PLANE / 4)] = 4.0f * d[1] - 5.0f * d[3] + d[5];
            }
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (tid + i * 256 < WPIECES) reinterpret_cast<f32x4*>(wlds)[tid + i * 256] = wst[i];
    };

    MVS_TICK(0)
    load_chunk(0);
    store_chunk();
    MVS_TICK(1)
    __syncthreads();
    MVS_TICK(2)

#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (c < 3) load_chunk(c + 1);
        MVS_TICK(3 + 5 * c)
        if (wave & 1) c43_chunk_mfmas<1>(ab[0], ab[1], ab[2], wA, wB, acc);
        else c43_chunk_mfmas<0>(ab[0], ab[1], ab[2], wA, wB, acc);
        MVS_TICK(4 + 5 * c)
        if (c < 3) {
            __syncthreads();  // every wave is done reading chunk c's planes and weights
            MVS_TICK(5 + 5 * c)
            store_chunk();
            MVS_TICK(6 + 5 * c)
            __syncthreads();
            MVS_TICK(7 + 5 * c)
        }
    }

    // exchange: D layout lane 4*blk + j, register i -> position 4*blk + i of the M-group, channel
    // 4*nt + j.  ex[t][pos][channel], pos = 64 mg + (row * 32 + xl) = y * 32 + x of the tile.
    __syncthreads();
    {
        const int blk = lane >> 2, j = lane & 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = u0 + i, pl = u >> 1, mg = u & 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* e = tile + ((pl * NPOS + 64 * mg + 4 * blk + r) * EXS);
                e[j] = acc[i][0][r];
                e[4 + j] = acc[i][1][r];
            }
        }
    }
    __syncthreads();
    const int pos = tid >> 1, ch = tid & 1;  // (y, x) of the 4 x 32 tile, channels 4 ch .. 4 ch + 3
    const int gy = y0 + (pos >> 5), gx = x0 + (pos & 31);
    if (gy >= H || gx >= W) return;
    f32x4 M[NT_PLANES];
#pragma unroll
    for (int q = 0; q < NT_PLANES; ++q) M[q] = *reinterpret_cast<const f32x4*>(tile + (q * NPOS + pos) * EXS + ch * 4);
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
        St<DT>::store4(y, (((size_t)(z0 + q) * H + gy) * W + gx) * 8 + ch * 4, v);
    }
    MVS_TICK(24)
#undef MVS_TICK
}

template <int DT>
static int run_conv0_w43(const void* x, void* y, const float* bw, const float* bias, int D, int H, int W,
                         hipStream_t s) {
    using namespace c43;
    const int nb = ((W + TX - 1) / TX) * ((H + TY - 1) / TY) * ((D + TZ - 1) / TZ);
#if MVS_ABLATE == 49
    static long long* dbg = nullptr;
    if (!dbg && hipMalloc(&dbg, 128 * sizeof(long long)) != hipSuccess) dbg = nullptr;
    if (dbg) (void)hipMemsetAsync(dbg, 0, 128 * sizeof(long long), s);
    conv0_w43_mfma_kernel<DT><<<nb, 256, 0, s>>>(x, bw, bias, y, D, H, W, dbg);
    if (dbg && nb > 2000) {
        long long h[128];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 4; ++w) {
            const long long* t = h + w * 32;
            fprintf(stderr, "conv0 wave %d: stage0 %lld bar %lld |", w, t[1] - t[0], t[2] - t[1]);
            for (int c = 0; c < 4; ++c)
                fprintf(stderr, " c%d: ld %lld mfma %lld bar %lld st %lld bar %lld |", c, t[3 + 5 * c] - (c ? t[2 + 5 * c] : t[2]),
                        t[4 + 5 * c] - t[3 + 5 * c], c < 3 ? t[5 + 5 * c] - t[4 + 5 * c] : 0, c < 3 ? t[6 + 5 * c] - t[5 + 5 * c] : 0,
                        c < 3 ? t[7 + 5 * c] - t[6 + 5 * c] : 0);
            fprintf(stderr, " end %lld total %lld\n", t[24] - t[19], t[24] - t[0]);
        }
    }
#else
    conv0_w43_mfma_kernel<DT><<<nb, 256, 0, s>>>(x, bw, bias, y, D, H, W);
#endif
    return check_hip(hipGetLastError(), "conv0_w43_mfma launch");
}

int launch_conv0_wino43(const void* x, void* y, const float* bw, const float* bias, int D, int H, int W,
                        int dtype, hipStream_t s) {
    if ((size_t)D * H * W * 8 * 4 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "conv0_wino43: plane of %zu bytes exceeds 31-bit buffer offsets",
                    (size_t)D * H * W * 8 * 4);
    MVS_DISPATCH_DTYPE(dtype, (run_conv0_w43<DT>(x, y, bw, bias, D, H, W, s)))
}

// wfold [27][32][8] (tap = kz*9 + ky*3 + kx) -> bw [4 chunks][6 t][9][2 halves][2 nt][4 j][4 k] with the
// z taps transformed by G of F(4,3) (double arithmetic, rounded once)
void pack_conv0_wino43_weights(const float* wfold, float* bw) {
    for (int c = 0; c < 4; ++c)
        for (int t = 0; t < 6; ++t)
            for (int tap = 0; tap < 9; ++tap)
                for (int half = 0; half < 2; ++half)
                    for (int nt = 0; nt < 2; ++nt)
                        for (int j = 0; j < 4; ++j)
                            for (int k = 0; k < 4; ++k) {
                                const int ci = 8 * c + 4 * half + k, co = 4 * nt + j;
                                const double g0 = wfold[((size_t)(0 * 9 + tap) * 32 + ci) * 8 + co];
                                const double g1 = wfold[((size_t)(1 * 9 + tap) * 32 + ci) * 8 + co];
                                const double g2 = wfold[((size_t)(2 * 9 + tap) * 32 + ci) * 8 + co];
                                double g;
                                switch (t) {
                                    case 0: g = g0 / 4.0; break;
                                    case 1: g = -(g0 + g1 + g2) / 6.0; break;
                                    case 2: g = -(g0 - g1 + g2) / 6.0; break;
                                    case 3: g = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
                                    case 4: g = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
                                    default: g = g2; break;
                                }
                                bw[((((((size_t)c * 6 + t) * 9 + tap) * 2 + half) * 2 + nt) * 4 + j) * 4 + k] = (float)g;
                            }
}


// =============================================================================================
// Winograd F(2,3) along z for the stride-1 layers conv2 16->16 and conv4 32->32 (models/mvsnet.py:39,42) on
// v_mfma_f32_16x16x4_f32, built like convg_mfma_kernel (conv3d_mfma.hip):
// M-tile = 2(y) x 8(x) voxels, N-tile = 16 channels, K-step = 2 taps x 8 channels -- but the block's
// tile is always two output planes deep, its four halo planes are stored TRANSFORMED, wave t runs the
// 3x3 (ky,kx) convolution of plane t for all BY x BX M-tiles and all N-tiles (9 taps -> 5 K-steps
// instead of 27 -> 14 for two planes: 20 vs 28 MFMA quads per M-tile pair), and the four partial
// results are combined through LDS once per tile.
// =============================================================================================
template <int CIN, int COUT, int BY, int BX>
struct ConvWZ {
    static constexpr int NT = COUT / 16;
    static constexpr int MT = BY * BX;
    static constexpr int NCH = CIN / 8;
    static constexpr int KS = 5;  // 9 taps padded to 10
    static constexpr int VS = 8;
    static constexpr int HY = 2 * BY + 2, HX = 8 * BX + 2;
    static constexpr int HXP = (HX + 7) / 8 * 8;
    static constexpr int PLANE = HY * HXP * VS;
    static constexpr int IN_FLOATS = 4 * PLANE;
    static constexpr int NCOL = HY * HX * 2;
    static constexpr int CPT = (NCOL + 255) / 256;
    static constexpr int NPOS = MT * 16;                  // (y, x) positions of the tile
    static constexpr int EXS = COUT + 4;                  // floats per position in the exchange tile
    static constexpr int EX_FLOATS = 4 * NPOS * EXS;
    static constexpr int TILE_FLOATS = IN_FLOATS > EX_FLOATS ? IN_FLOATS : EX_FLOATS;
    static constexpr int NUNIT = NPOS * (COUT / 8);       // (position, 8-channel plane) output units
    static constexpr int UPT = (NUNIT + 255) / 256;
    static_assert(CPT * 4 <= 32, "too many staged pieces per thread");
    static constexpr int tap_off(int tap) {
        const int t = tap > 8 ? 8 : tap;
        return ((t / 3) * HXP + t % 3) * VS;
    }
};

template <int DT, int CIN, int COUT, int BY, int BX>
__global__ __launch_bounds__(256) void convwz_mfma_kernel(
    const void* __restrict__ x,      // [CIN/8][D][H][W][8] storage dtype DT
    const float* __restrict__ bp,    // [NCH][4 t][NT][5][64][4]
    const float* __restrict__ bias,  // [COUT]
    void* __restrict__ y,            // [COUT/8][D][H][W][8]
    int D, int H, int W) {
    using G = ConvWZ<CIN, COUT, BY, BX>;
    __shared__ __attribute__((aligned(16))) float tile[G::TILE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, t = tid >> 6;  // wave t -> transformed plane t
    const int nbx = (W + 8 * BX - 1) / (8 * BX), nby = (H + 2 * BY - 1) / (2 * BY);
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int x0 = bx * 8 * BX, y0 = by * 2 * BY, z0 = bz * 2;
    const size_t HW8 = (size_t)H * W * 8, V8 = (size_t)D * HW8;

    int goff[G::CPT], loff[G::CPT];
    unsigned okxy = 0;
#pragma unroll
    for (int i = 0; i < G::CPT; ++i) {
        const int col = tid + i * 256;
        const int half = col & 1, v = col >> 1;
        const int hx = v % G::HX, hy = v / G::HX;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = col < G::NCOL && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? (int)(((size_t)gy * W + gx) * 8 + half * 4) : 0;
        okxy |= ok ? (1u << i) : 0u;
        loff[i] = (col < G::NCOL) ? (hy * G::HXP + hx) * G::VS + half * 4 : -1;
    }
    size_t zoff[4];
    bool zok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gz = z0 - 1 + q;
        zok[q] = gz >= 0 && gz < D;
        zoff[q] = zok[q] ? (size_t)gz * HW8 : 0;
    }

    // A fragment: lane (r -> voxel (ry, rx) of the M-tile, g): k-step ks covers taps 2ks (g>>1 == 0)
    // and 2ks+1 (g>>1 == 1), channels 4(g&1)..+3 of the chunk
    const int r = lane & 15, g = lane >> 4, gh = g >> 1;
    const int ry = r >> 3, rx = r & 7;
    int abase[G::MT];
#pragma unroll
    for (int i = 0; i < G::MT; ++i) {
        const int tx = i % BX, ty = i / BX;
        abase[i] = t * G::PLANE + ((2 * ty + ry) * G::HXP + 8 * tx + rx) * G::VS + (g & 1) * 4;
    }
    f32x4 acc[G::NT][G::MT];
#pragma unroll
    for (int n = 0; n < G::NT; ++n)
#pragma unroll
        for (int i = 0; i < G::MT; ++i) acc[n][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 breg[G::NT][G::KS];
    f32x4 stg[G::CPT][4];

#define WZ_LOAD_B(C)                                                                                 \
    {                                                                                                \
        _Pragma("unroll") for (int n = 0; n < G::NT; ++n) {                                          \
            const f32x4* bsrc = reinterpret_cast<const f32x4*>(bp) +                                 \
                                ((size_t)(((C) * 4 + t) * G::NT + n) * G::KS) * 64 + lane;           \
            _Pragma("unroll") for (int ks = 0; ks < G::KS; ++ks) breg[n][ks] = bsrc[ks * 64];        \
        }                                                                                            \
    }
#define WZ_LOAD_A(C)                                                                                 \
    {                                                                                                \
        const size_t plane = (size_t)(C) * V8;                                                       \
        _Pragma("unroll") for (int i = 0; i < G::CPT; ++i)                                           \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                            \
                stg[i][q] = St<DT>::load4(x, plane + zoff[q] + goff[i]);                             \
    }
#define WZ_STORE_A()                                                                                 \
    {                                                                                                \
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};                                              \
        _Pragma("unroll") for (int i = 0; i < G::CPT; ++i)                                           \
            if (loff[i] >= 0) {                                                                      \
                const bool in = (okxy >> i) & 1u;                                                    \
                const f32x4 d0 = (in && zok[0]) ? stg[i][0] : zero;                                  \
                const f32x4 d1 = (in && zok[1]) ? stg[i][1] : zero;                                  \
                const f32x4 d2 = (in && zok[2]) ? stg[i][2] : zero;                                  \
                const f32x4 d3 = (in && zok[3]) ? stg[i][3] : zero;                                  \
                *reinterpret_cast<f32x4*>(tile + loff[i]) = d0 - d2;                                 \
                *reinterpret_cast<f32x4*>(tile + G::PLANE + loff[i]) = d1 + d2;                      \
                *reinterpret_cast<f32x4*>(tile + 2 * G::PLANE + loff[i]) = d2 - d1;                  \
                *reinterpret_cast<f32x4*>(tile + 3 * G::PLANE + loff[i]) = d1 - d3;                  \
            }                                                                                        \
    }

    WZ_LOAD_B(0)
    WZ_LOAD_A(0)
    WZ_STORE_A()
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < G::NCH; ++c) {
        if (c + 1 < G::NCH) WZ_LOAD_A(c + 1)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const int koff = gh ? G::tap_off(2 * ks + 1) : G::tap_off(2 * ks);
            f32x4 a[G::MT];
#pragma unroll
            for (int i = 0; i < G::MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(tile + abase[i] + koff);
#pragma unroll
            for (int n = 0; n < G::NT; ++n) {
                const f32x4 bq = breg[n][ks];
#pragma unroll
                for (int i = 0; i < G::MT; ++i) acc[n][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[n][i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < G::MT; ++i) acc[n][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[n][i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < G::MT; ++i) acc[n][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[n][i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < G::MT; ++i) acc[n][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[n][i], 0, 0, 0);
            }
        }
        if (c + 1 < G::NCH) {
            WZ_LOAD_B(c + 1)
            __syncthreads();
            WZ_STORE_A()
            __syncthreads();
        }
    }
#undef WZ_LOAD_B
#undef WZ_LOAD_A
#undef WZ_STORE_A

    // exchange: D layout col n = lane&15 -> channel 16 nt + n; row m = 4*(lane>>4) + e -> voxel of the
    // M-tile.  ex[t][pos][channel], pos = 16 * mtile + m.
    __syncthreads();
#pragma unroll
    for (int n = 0; n < G::NT; ++n)
#pragma unroll
        for (int i = 0; i < G::MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * (lane >> 4) + e;
                tile[(t * G::NPOS + 16 * i + m) * G::EXS + 16 * n + (lane & 15)] = acc[n][i][e];
            }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < G::UPT; ++j) {
        const int u = tid + j * 256;
        if (u >= G::NUNIT) break;
        const int pos = u % G::NPOS, pl = u / G::NPOS;
        const int mt = pos >> 4, m = pos & 15;
        const int gy = y0 + 2 * (mt / BX) + (m >> 3), gx = x0 + 8 * (mt % BX) + (m & 7);
        if (gy >= H || gx >= W) continue;
        f32x4 M[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* e = tile + (q * G::NPOS + pos) * G::EXS + pl * 8;
            M[q][0] = *reinterpret_cast<const f32x4*>(e);
            M[q][1] = *reinterpret_cast<const f32x4*>(e + 4);
        }
        const f32x4 bv0 = *reinterpret_cast<const f32x4*>(bias + pl * 8);
        const f32x4 bv1 = *reinterpret_cast<const f32x4*>(bias + pl * 8 + 4);
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            if (z0 + o >= D) break;
            f32x4 lo, hi;
            if (o == 0) {
                lo = (M[0][0] + M[1][0]) + M[2][0] + bv0;
                hi = (M[0][1] + M[1][1]) + M[2][1] + bv1;
            } else {
                lo = (M[1][0] - M[2][0]) - M[3][0] + bv0;
                hi = (M[1][1] - M[2][1]) - M[3][1] + bv1;
            }
            lo = __builtin_elementwise_max(lo, zero);
            hi = __builtin_elementwise_max(hi, zero);
            const size_t off = (size_t)pl * V8 + (((size_t)(z0 + o) * H + gy) * W + gx) * 8;
            if constexpr (DT == MVS_F32) {
                St<DT>::store4(y, off, lo);
                St<DT>::store4(y, off + 4, hi);
            } else {
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                store8_16<DT>(y, off, v);
            }
        }
    }
}

template <int DT, int CIN, int COUT, int BY, int BX>
static int run_convwz(const void* x, void* y, const float* bp, const float* bias, int D, int H, int W,
                      hipStream_t s) {
    if ((size_t)D * H * W * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "convwz_mfma: plane exceeds 31-bit offsets");
    const int nb = ((W + 8 * BX - 1) / (8 * BX)) * ((H + 2 * BY - 1) / (2 * BY)) * ((D + 1) / 2);
    convwz_mfma_kernel<DT, CIN, COUT, BY, BX><<<nb, 256, 0, s>>>(x, bp, bias, y, D, H, W);
    return check_hip(hipGetLastError(), "convwz_mfma launch");
}

template <int DT>
static int launch_convwz_dt(int layer, const void* x, void* y, const float* bp, const float* bias, int D,
                            int H, int W, hipStream_t s) {
    switch (layer) {
        case 2: return run_convwz<DT, 16, 16, 4, 2>(x, y, bp, bias, D, H, W, s);
        case 4: return run_convwz<DT, 32, 32, 2, 2>(x, y, bp, bias, D, H, W, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "convwz_mfma: layer %d not covered", layer);
    }
}

// layers 2, 4 (conv2, conv4)
int launch_convwz_mfma(int layer, const void* x, void* y, const float* bp, const float* bias, int D, int H,
                       int W, int dtype, hipStream_t s) {
    MVS_DISPATCH_DTYPE(dtype, (launch_convwz_dt<DT>(layer, x, y, bp, bias, D, H, W, s)))
}

// wfold [27][cin][cout] -> bp [cin/8][4 t][cout/16][5][64][4], z taps transformed as for conv0
void pack_convwz_weights(const float* wfold, int cin, int cout, float* bp) {
    const int nch = cin / 8, nt = cout / 16;
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < 4; ++t)
            for (int n = 0; n < nt; ++n)
                for (int ks = 0; ks < 5; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j4 = 0; j4 < 4; ++j4) {
                            const int g = lane >> 4, col = lane & 15;
                            const int tap = 2 * ks + (g >> 1);
                            const int ci = 8 * c + 4 * (g & 1) + j4, co = 16 * n + col;
                            float v = 0.0f;
                            if (tap < 9) {
                                const float g0 = wfold[((size_t)(0 * 9 + tap) * cin + ci) * cout + co];
                                const float g1 = wfold[((size_t)(1 * 9 + tap) * cin + ci) * cout + co];
                                const float g2 = wfold[((size_t)(2 * 9 + tap) * cin + ci) * cout + co];
                                v = t == 0 ? g0 : t == 1 ? ((g0 + g1) + g2) * 0.5f
                                  : t == 2 ? ((g0 - g1) + g2) * 0.5f : g2;
                            }
                            bp[(((((size_t)c * 4 + t) * nt + n) * 5 + ks) * 64 + lane) * 4 + j4] = v;
                        }
}

size_t convwz_panel_floats(int cin, int cout) { return (size_t)(cin / 8) * 4 * (cout / 16) * 5 * 64 * 4; }

// wfold [27][32][8] (tap = kz*9 + ky*3 + kx) -> bw [4 chunks][4 t][9][2 halves][2 nt][4 j][4 k] with

}  // namespace mvs
