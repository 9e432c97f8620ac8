// attic: conv0 with Winograd F(2,3) along z on the 4x4x1 MFMA (the round-1 default, 0.416-0.433 ms at cfg2;
// superseded by F(4,3), 0.33-0.35 ms: conv_winograd.hip).  Not built.
// conv0_winograd.hip -- conv0 (32 -> 8 channels, 68 % of the path's FLOPs; reference
// models/mvsnet.py:36, block models/module.py:26-33) with Winograd F(2,3) along z on the 4x4x1 MFMA.
//
// The direct 4x4x1 kernel (conv3d_mfma.hip) is matrix-pipe-bound: 13,824 MFMAs per 2x8x32 tile and no
// wasted columns.  The only way below that floor is fewer multiplications.  F(2,3) along ONE axis
// computes two outputs from four transformed inputs with 4 instead of 6 multiplications:
//     U0 = d0 - d2   U1 = d1 + d2   U2 = d2 - d1   U3 = d1 - d3
//     G0 = g0        G1 = (g0+g1+g2)/2   G2 = (g0-g1+g2)/2   G3 = g2
//     m_t = U_t * G_t          y0 = m0 + m1 + m2        y1 = m1 - m2 - m3
// Taking z as that axis costs nothing in LDS: the block's tile is two output planes deep, so its halo
// is exactly the four input planes d0..d3, and the four TRANSFORMED planes U0..U3 take their place
// (same 65 KB, same 48-byte voxel stride).  The transform is 4 vector adds per staged column, done
// when the chunk is written to LDS; the weights are transformed on the host.  What remains is four
// independent 3x3 (ky,kx) convolutions m_t = conv2d(U_t, G_t) -- one per wave -- and a final
// combination: 9,216 MFMAs per tile, 2/3 of the direct form.
//
//   block: 256 threads = 4 waves; wave t owns transformed plane t: 4 M-groups (2 rows x 32 x each)
//          x 2 N-tiles (4 channels each) = 8 accumulators; per (tap, half): 4 A reads + 2 broadcast B
//          reads feed 32 MFMAs (the direct kernel: 4 reads per 16)
//   K    : 4 chunks of 8 input channels (C8 planes), staged + transformed per chunk, chunk c+1's loads
//          in flight during chunk c's MFMAs (unconditional loads: counted vmcnt)
//   end  : the four m_t go through LDS once per tile; every thread then owns one (y,x) position,
//          forms y0 / y1 (+bias, ReLU) and writes the two voxels with 16-byte stores
// Results are not bit-equal to an fmaf chain (the sums are re-associated); the difference is a few
// fp32 ulp per output (tests/test_gpu_parity.py bounds it against the oracle).
#include "mvs_internal.h"
namespace c0w {
constexpr int TY = 8, TX = 32;             // output tile: 2 (the z pair) x 8 x 32
constexpr int HY = TY + 2, HX = TX + 2;
constexpr int VS = 12;                     // floats per voxel in LDS (8 data + 4 pad)
constexpr int PLANE = HY * HX * VS;        // one transformed plane: 4080 floats
static_assert(PLANE % 4 == 0, "planes are addressed in 16-byte units");
constexpr int TILE_FLOATS = 4 * PLANE;     // 65,280 B
constexpr int NCOL = HY * HX * 2;          // z-columns of 16-byte pieces: (y, x, half)
constexpr int CPT = (NCOL + 255) / 256;    // 3 per thread
constexpr int BW_FLOATS = 4 * 9 * 2 * 2 * 4 * 4;  // per chunk [t][tap][half][nt][j][k]
constexpr int WPIECES = BW_FLOATS / 4;     // 576 16-byte pieces
constexpr int WPT = (WPIECES + 255) / 256;
constexpr int EXS = 12;                    // floats per position in the exchange tile
static_assert(4 * 256 * EXS <= TILE_FLOATS, "exchange tile must fit in the input tile");
}  // namespace c0w

template <int DT>
__global__ __launch_bounds__(256, 2) void conv0_wz_mfma_kernel(
    const void* __restrict__ x,      // [4][D][H][W][8] storage dtype DT
    const float* __restrict__ bw,    // [4 chunks][4 t][9 taps][2 halves][2 nt][4 j][4 k]
    const float* __restrict__ bias,  // [8]
    void* __restrict__ y,            // [D][H][W][8] storage dtype DT
    int D, int H, int W) {
    using namespace c0w;
    __shared__ __attribute__((aligned(16))) float tile[TILE_FLOATS + BW_FLOATS];
    float* wlds = tile + TILE_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * 2;
    const size_t HW8 = (size_t)H * W * 8, V8 = (size_t)D * HW8;

    // staging bookkeeping: a thread owns up to CPT z-columns (y, x, half) and stages all four planes
    // of each.  Loads are unconditional (masked afterwards) so that the compiler keeps counted waits.
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
        // in 16-byte units: VS = 12 floats and half * 4 are multiples of 4, and saying so lets hipcc emit
        // one ds_write_b128 per piece instead of two ds_write2_b32 (it cannot prove the alignment of a float index)
        loff[i] = (col < NCOL) ? (v * VS + half * 4) / 4 : -1;
    }
    size_t zoff[4];
    bool zok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gz = z0 - 1 + q;
        zok[q] = gz >= 0 && gz < D;
        zoff[q] = zok[q] ? (size_t)gz * HW8 : 0;
    }

    // wave t -> transformed plane t; lane -> (row, xl) of an M-group (2 rows x 32 x)
    const int t = wave;
    const int row = lane >> 5, xl = lane & 31;
    const float* abase = tile + t * PLANE + (row * HX + xl) * VS;
    const float* wbase = wlds + t * (9 * 2 * 2 * 16) + (lane & 3) * 4;

    f32x4 acc[4][2];  // [M-group][nt]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 stg[CPT][4];
    f32x4 wst[WPT];
    // fp32 storage: raw buffer loads -- a piece outside the volume gets a byte offset beyond the
    // buffer's range and the hardware returns zeros, so the transform needs no select instructions
    unsigned boff[CPT][4];
    if constexpr (DT == MVS_F32) {
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                boff[i][q] = (((okxy >> i) & 1u) && zok[q]) ? (unsigned)((zoff[q] + (size_t)goff[i]) * 4) : 0x80000000u;
    }
#define C0W_LOAD(C)                                                                            \
    {                                                                                          \
        const size_t plane = (size_t)(C) * V8;                                                 \
        if constexpr (DT == MVS_F32) {                                                         \
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(               \
                const_cast<float*>(static_cast<const float*>(x) + plane), (short)0, (int)(V8 * 4), 0x00020000); \
            _Pragma("unroll") for (int i = 0; i < CPT; ++i)                                    \
                _Pragma("unroll") for (int q = 0; q < 4; ++q)                                  \
                    stg[i][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)boff[i][q], 0, 0)); \
        } else {                                                                               \
            _Pragma("unroll") for (int i = 0; i < CPT; ++i)                                    \
                _Pragma("unroll") for (int q = 0; q < 4; ++q)                                  \
                    stg[i][q] = St<DT>::load4(x, plane + zoff[q] + goff[i]);                   \
        }                                                                                      \
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(bw) + (size_t)(C) * WPIECES;        \
        _Pragma("unroll") for (int i = 0; i < WPT; ++i)                                        \
            wst[i] = wsrc[min(tid + i * 256, WPIECES - 1)];                                    \
    }
#define C0W_STORE()                                                                            \
    {                                                                                          \
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};                                        \
        _Pragma("unroll") for (int i = 0; i < CPT; ++i)                                        \
            if (loff[i] >= 0) {                                                                \
                const bool in = (okxy >> i) & 1u;                                              \
                const bool hw0 = DT == MVS_F32;  /* zeros already delivered by the buffer loads */ \
                const f32x4 d0 = (hw0 || (in && zok[0])) ? stg[i][0] : zero;                    \
                const f32x4 d1 = (hw0 || (in && zok[1])) ? stg[i][1] : zero;                    \
                const f32x4 d2 = (hw0 || (in && zok[2])) ? stg[i][2] : zero;                    \
                const f32x4 d3 = (hw0 || (in && zok[3])) ? stg[i][3] : zero;                    \
                f32x4* t4 = reinterpret_cast<f32x4*>(tile) + loff[i];                          \
                t4[0] = d0 - d2;                                                               \
                t4[PLANE / 4] = d1 + d2;                                                       \
                t4[2 * (PLANE / 4)] = d2 - d1;                                                 \
                t4[3 * (PLANE / 4)] = d1 - d3;                                                 \
            }                                                                                  \
        _Pragma("unroll") for (int i = 0; i < WPT; ++i)                                        \
            if (tid + i * 256 < WPIECES) reinterpret_cast<f32x4*>(wlds)[tid + i * 256] = wst[i]; \
    }

    C0W_LOAD(0)
    C0W_STORE()
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (c < 3) C0W_LOAD(c + 1)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int koff = ((tap / 3) * HX + tap % 3) * VS;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 a[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    a[m] = *reinterpret_cast<const f32x4*>(abase + 2 * m * HX * VS + koff + half * 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(wbase + ((tap * 2 + half) * 2 + 0) * 16);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(wbase + ((tap * 2 + half) * 2 + 1) * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        acc[m][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m][k], b0[k], acc[m][0], 0, 0, 0);
                        acc[m][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m][k], b1[k], acc[m][1], 0, 0, 0);
                    }
            }
        }
        if (c < 3) {
            __syncthreads();  // every wave is done reading chunk c's planes and weights
            C0W_STORE()
            __syncthreads();
        }
    }
#undef C0W_LOAD
#undef C0W_STORE

    // exchange: D layout lane 4*blk + j, register i -> position 4*blk + i of the M-group, channel
    // 4*nt + j.  ex[t][pos][channel], pos = 64 m + (row_in_group * 32 + xl) = row * 32 + xl of the tile.
    __syncthreads();
    {
        const int blk = lane >> 2, j = lane & 3;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* e = tile + ((t * 256 + 64 * m + 4 * blk + i) * EXS);
                e[j] = acc[m][0][i];
                e[4 + j] = acc[m][1][i];
            }
    }
    __syncthreads();
    const int pos = tid;  // (row, xl) of the 8 x 32 tile
    const int gy = y0 + (pos >> 5), gx = x0 + (pos & 31);
    if (gy >= H || gx >= W) return;
    f32x4 M[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* e = tile + (q * 256 + pos) * EXS;
        M[q][0] = *reinterpret_cast<const f32x4*>(e);
        M[q][1] = *reinterpret_cast<const f32x4*>(e + 4);
    }
    const f32x4 bv0 = *reinterpret_cast<const f32x4*>(bias), bv1 = *reinterpret_cast<const f32x4*>(bias + 4);
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
        const size_t off = (((size_t)(z0 + o) * H + gy) * W + gx) * 8;
        if constexpr (DT == MVS_F32) {
            St<DT>::store4(y, off, lo);
            St<DT>::store4(y, off + 4, hi);
        } else {
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            store8_16<DT>(y, off, v);
        }
    }
}

template <int DT>
static int run_conv0_wz(const void* x, void* y, const float* bw, const float* bias, int D, int H, int W,
                        hipStream_t s) {
    const int nb = ((W + c0w::TX - 1) / c0w::TX) * ((H + c0w::TY - 1) / c0w::TY) * ((D + 1) / 2);
    conv0_wz_mfma_kernel<DT><<<nb, 256, 0, s>>>(x, bw, bias, y, D, H, W);
    return check_hip(hipGetLastError(), "conv0_wz_mfma launch");
}

int launch_conv0_winograd(const void* x, void* y, const float* bw, const float* bias, int D, int H, int W,
                          int dtype, hipStream_t s) {
    if ((size_t)D * H * W * 8 * 4 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "conv0_winograd: plane of %zu bytes exceeds 31-bit buffer offsets",
                    (size_t)D * H * W * 8 * 4);
    MVS_DISPATCH_DTYPE(dtype, (run_conv0_wz<DT>(x, y, bw, bias, D, H, W, s)))
}
// the z taps transformed: G0 = g0, G1 = (g0+g1+g2)/2, G2 = (g0-g1+g2)/2, G3 = g2
void pack_conv0_winograd_weights(const float* wfold, float* bw) {
    for (int c = 0; c < 4; ++c)
        for (int t = 0; t < 4; ++t)
            for (int tap = 0; tap < 9; ++tap)
                for (int half = 0; half < 2; ++half)
                    for (int nt = 0; nt < 2; ++nt)
                        for (int j = 0; j < 4; ++j)
                            for (int k = 0; k < 4; ++k) {
                                const int ci = 8 * c + 4 * half + k, co = 4 * nt + j;
                                const float g0 = wfold[((size_t)(0 * 9 + tap) * 32 + ci) * 8 + co];
                                const float g1 = wfold[((size_t)(1 * 9 + tap) * 32 + ci) * 8 + co];
                                const float g2 = wfold[((size_t)(2 * 9 + tap) * 32 + ci) * 8 + co];
                                const float g = t == 0 ? g0 : t == 1 ? ((g0 + g1) + g2) * 0.5f
                                              : t == 2 ? ((g0 - g1) + g2) * 0.5f : g2;
                                bw[((((((size_t)c * 4 + t) * 9 + tap) * 2 + half) * 2 + nt) * 4 + j) * 4 + k] = g;
                            }
}

