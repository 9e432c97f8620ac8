// conv3d_mfma.hip -- fp32-MFMA implicit-GEMM 3x3x3 convolutions for gfx950 (MI355X).
//
// Every convolution of CostRegNet (reference models/mvsnet.py:35-73, ConvBnReLU3D of
// models/module.py:26-33) except the 8 -> 1 `prob` layer runs here on exact-fp32 MFMA
// (64 FLOP/clk/SIMD, bit-equal to an fmaf chain).  Kernels in this file:
//   conv0_4x4_mfma_kernel    conv0 (32 -> 8, 68 % of the path's FLOPs) on v_mfma_f32_4x4x1_16b_f32, direct form:
//                            N = 4 channels per tile, so Cout = 8 is two exact N-tiles (MVS_CONV0_WINO=0
//                            and volumes beyond the Winograd kernel's 31-bit offsets; cross-check of conv0_w43)
//   convg_mfma_kernel        conv1..conv6 (stride 1 / 2, Cout multiple of 16) on 16x16x4
//   deconvg_mfma_kernel      conv7 / conv9 / conv11 (+ skip) on 16x16x4, gather form over the input
// plus the host-side packers that lay the BN-folded weights out in per-lane fragment order.
// All kernels share one scheme: K is processed in chunks of 8 input channels = one plane of the
// C8-planar input; per chunk the block's halo tile of that plane is staged in LDS (voxel stride /
// row pitch chosen so the A-fragment ds_read_b128 are conflict-free) while the previous chunk's
// MFMAs run; the global loads are unconditional so hipcc keeps counted vmcnt waits.
//
#include <cstdlib>

#include "mvs_internal.h"
#include "storage.h"

namespace mvs {

typedef unsigned u32x4r __attribute__((__vector_size__(4 * sizeof(unsigned))));

// Diagnostic builds only (make ABLATE=n -> libmvs_hip_ablate<n>.so, wrong results by design):
//   1 = conv0 without epilogue stores, 2 = without chunk re-staging, 3 = also without the
//   staging loads.  Product builds leave MVS_ABLATE at 0.
#ifndef MVS_ABLATE
#define MVS_ABLATE 0
#endif

// ---------------------------------------------------------------------------------------------
// conv0 on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per instruction (512 FLOP
// in 8 cycles = the same 64 FLOP/clk/SIMD as the 16x16x4 form), so N = 4 output channels per
// tile and Cout = 8 is two exact N-tiles: no padded or Toeplitz-wasted columns (a 16x16x4 form with
// N = 2 x-adjacent outputs x 8 channels spends 4/3 of the useful MFMA cycles: attic/conv0_pair_mfma.hip).
//   lane l = 4*blk + i supplies A = in[voxel l of the 64-voxel M-group][k] and B = w[k][4*nt + i];
//   D: lane 4*blk + j, register i' = out[voxel 4*blk + i'][4*nt + j]   (probed on gfx950:
//   tools/probes/mfma4x4_probe.hip)
// 2 x 8 x 32 output tile, halo staging with a 48-byte voxel stride (keeps the 64 x 16 B A reads
// conflict-free), chunk c+1 staged under chunk c's MFMAs; a wave owns two M-groups (2 rows x
// 32 x each).  The chunk's weights (27 taps x 8 ci x 8 co = 6.9 KB, [tap][half][nt][j][4 k])
// live in LDS next to the tile: every (tap, half) step is 2 A reads + 2 broadcast B reads
// (ds_read_b128) feeding 16 MFMAs.
// ---------------------------------------------------------------------------------------------
// Tile depth / block size of the 4x4x1 kernel (compile-time knobs for A/B builds):
//   C0Q_TZ = 2, 256 threads: 72 KB LDS, 2 blocks per CU            [default]
//   C0Q_TZ = 4, 512 threads: 105 KB LDS, 1 block per CU, halo ratio 2.0 instead of 2.66
#ifndef C0Q_TZ
#define C0Q_TZ 2
#endif
namespace c0q {
constexpr int TZ = C0Q_TZ, TY = 8, TX = 32;
constexpr int THREADS = 128 * TZ;                     // one wave per (z slice, 4 rows)
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int VS = 12;
constexpr int TILE_FLOATS = HZ * HY * HX * VS;
constexpr int NPIECE = HZ * HY * HX * 2;
constexpr int PIECES_PER_THREAD = (NPIECE + THREADS - 1) / THREADS;
constexpr int BQ_FLOATS = 27 * 2 * 2 * 4 * 4;  // per chunk
constexpr int WPIECES = BQ_FLOATS / 4;         // 432 16-byte pieces of weights per chunk
constexpr int WPT = (WPIECES + THREADS - 1) / THREADS;
}  // namespace c0q

template <int DT>
__global__ __launch_bounds__(c0q::THREADS, 2) void conv0_4x4_mfma_kernel(
    const void* __restrict__ x,      // [4][D][H][W][8] storage dtype DT
    const float* __restrict__ bq,    // [4 chunks][27 taps][2 halves][2 nt][4 j][4 k]
    const float* __restrict__ bias,  // [8]
    void* __restrict__ y,            // [D][H][W][8] storage dtype DT
    int D, int H, int W) {
    using namespace c0q;
    __shared__ __attribute__((aligned(16))) float tile[TILE_FLOATS + BQ_FLOATS];
    float* wlds = tile + TILE_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    const size_t V = (size_t)D * H * W;

    int goff[PIECES_PER_THREAD];
    int loff[PIECES_PER_THREAD];
    unsigned inside = 0;
#pragma unroll
    for (int i = 0; i < PIECES_PER_THREAD; ++i) {
        const int p = tid + i * THREADS;
        const int half = p & 1, v = p >> 1;
        const int hx = v % HX, t = v / HX;
        const int hy = t % HY, hz = t / HY;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = p < NPIECE && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? (int)((((size_t)gz * H + gy) * W + gx) * 8 + half * 4) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = (p < NPIECE) ? v * VS + half * 4 : -1;
    }

    // lane -> voxel (row, xl) of an M-group; wave -> z slice and 4 rows (two M-groups of 2 rows)
    const int row = lane >> 5, xl = lane & 31;
    const int zt = wave >> 1, yt0 = 4 * (wave & 1);
    const float* abase0 = tile + ((zt * HY + yt0 + row) * HX + xl) * VS;
    const float* abase1 = abase0 + 2 * HX * VS;
    const float* wbase = wlds + (lane & 3) * 4;

    f32x4 acc[2][2];  // [M-group][nt]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 stg[PIECES_PER_THREAD];
    f32x4 wst[WPT];
#define MVS_LOAD_A(C)                                                                         \
    {                                                                                         \
        const size_t plane = (size_t)(C) * V * 8;                                             \
        _Pragma("unroll") for (int i = 0; i < PIECES_PER_THREAD; ++i)                         \
            stg[i] = St<DT>::load4(x, plane + goff[i]);                                       \
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(bq) + (size_t)(C) * WPIECES;       \
        _Pragma("unroll") for (int i = 0; i < WPT; ++i)                                       \
            wst[i] = wsrc[min(tid + i * THREADS, WPIECES - 1)];                               \
    }
#define MVS_STORE_A()                                                                         \
    {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < PIECES_PER_THREAD; ++i)                         \
            if (loff[i] >= 0)                                                                 \
                *reinterpret_cast<f32x4*>(tile + loff[i]) =                                   \
                    ((inside >> i) & 1u) ? stg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};              \
        _Pragma("unroll") for (int i = 0; i < WPT; ++i)                                       \
            if (tid + i * THREADS < WPIECES)                                                  \
                reinterpret_cast<f32x4*>(wlds)[tid + i * THREADS] = wst[i];                   \
    }

    MVS_LOAD_A(0)
    MVS_STORE_A()
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (c < 3) MVS_LOAD_A(c + 1)
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int koff = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * VS;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(abase0 + koff + half * 4);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(abase1 + koff + half * 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(wbase + ((tap * 2 + half) * 2 + 0) * 16);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(wbase + ((tap * 2 + half) * 2 + 1) * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[k], b0[k], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[k], b1[k], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[k], b0[k], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[k], b1[k], acc[1][1], 0, 0, 0);
                }
            }
        }
        if (c < 3) {
            __syncthreads();  // every wave is done reading chunk c's tile and weights
            MVS_STORE_A()
            __syncthreads();
        }
    }
#undef MVS_LOAD_A
#undef MVS_STORE_A

    // epilogue: lane 4*blk + j, register i -> voxel 4*blk + i of the M-group, channel 4*nt + j
    const int blk = lane >> 2, j = lane & 3;
    const int gz = z0 + zt;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vox = 4 * blk + i;  // 0..63: (row, xl) = (vox >> 5, vox & 31)
            const int gy = y0 + yt0 + 2 * m + (vox >> 5), gx = x0 + (vox & 31);
            if (gz < D && gy < H && gx < W) {
                const size_t o = (((size_t)gz * H + gy) * W + gx) * 8;
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    St<DT>::store1(y, o + 4 * n + j, fmaxf(acc[m][n][i] + bias[4 * n + j], 0.0f));
            }
        }
}

// wfold [27][32][8] -> bq [4 chunks][27][2 halves][2 nt][4 j][4 k]:  w[tap][8c + 4half + k][4nt + j]
void pack_conv0_4x4_weights(const float* wfold, float* bq) {
    for (int c = 0; c < 4; ++c)
        for (int tap = 0; tap < 27; ++tap)
            for (int half = 0; half < 2; ++half)
                for (int nt = 0; nt < 2; ++nt)
                    for (int j = 0; j < 4; ++j)
                        for (int k = 0; k < 4; ++k)
                            bq[(((((size_t)c * 27 + tap) * 2 + half) * 2 + nt) * 4 + j) * 4 + k] =
                                wfold[((size_t)tap * 32 + 8 * c + 4 * half + k) * 8 + 4 * nt + j];
}

template <int DT>
static int run_conv0_4x4(const void* x, void* y, const float* bq, const float* bias, int D, int H, int W,
                         int nb, hipStream_t s) {
    conv0_4x4_mfma_kernel<DT><<<nb, c0q::THREADS, 0, s>>>(x, bq, bias, y, D, H, W);
    return check_hip(hipGetLastError(), "conv0_4x4_mfma launch");
}

int launch_conv0_mfma(const void* x, void* y, const float* bq, const float* bias, int D, int H, int W, int dtype,
                      hipStream_t s) {
    if ((size_t)D * H * W * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "conv0_mfma: plane of %zu elements exceeds 31-bit offsets",
                    (size_t)D * H * W * 8);
    const int nbq = ((W + c0q::TX - 1) / c0q::TX) * ((H + c0q::TY - 1) / c0q::TY) * ((D + c0q::TZ - 1) / c0q::TZ);
    MVS_DISPATCH_DTYPE(dtype, (run_conv0_4x4<DT>(x, y, bq, bias, D, H, W, nbq, s)))
}

// number of CUs of the current device (persistent grids are sized from it); MVS_PERSIST_CUS overrides
static int persistent_blocks_per_cu_scale() {
    static const int cus = [] {
        const char* e = getenv("MVS_PERSIST_CUS");
        const int v = e ? atoi(e) : 0;
        if (v > 0) return v;
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n;
    }();
    return cus;
}

// =============================================================================================
// Generic fp32-MFMA implicit-GEMM 3x3x3 convolution (stride 1 or 2), Cout a multiple of 16:
// conv1..conv6 of CostRegNet (models/mvsnet.py:38-45; ConvBnReLU3D of models/module.py:26-33).
//
//   M : output voxels; one MFMA tile = 2(y) x 8(x) voxels (W, H are multiples of 8 / 2 at every
//       level the ABI admits, so tiles never straddle a row end except on ragged shapes: masked)
//   N : 16 output channels per N-tile, NT = COUT/16 tiles
//   K : (tap, ci): chunks of 8 input channels (= one C8 plane), 27 taps padded to 28 so that a
//       k-step of 16 is exactly 2 taps x 8 channels: 14 k-steps per chunk
//   A[m][k] = in[S*oz+kz-1][S*oy+ky-1][S*ox+kx-1][ci]         B[k][n] = w[tap][ci][co]
// Block = 4 waves = NT N-tiles x MG M-groups of MPW M-tiles each; the block tile is BZ x BY x BX
// M-tiles.  Per chunk the input halo tile is staged in LDS (row pitch padded to a multiple of 8
// voxels, voxel stride 32 B for stride 1 / 48 B for stride 2: conflict-free ds_read_b128), the
// (chunk, N-tile) B panel sits in registers (56 VGPRs), staging of chunk c+1 overlaps the MFMAs
// of chunk c exactly as in the conv0 kernel above.
// =============================================================================================
template <int CIN, int COUT, int S, int BZ, int BY, int BX>
struct ConvG {
    static constexpr int NT = COUT / 16;
    static constexpr int MG = 4 / NT;
    static constexpr int MT = BZ * BY * BX;
    static constexpr int MPW = MT / MG;
    static constexpr int NCH = CIN / 8;
    static constexpr int KS = 14;
    static constexpr int VS = (S == 1) ? 8 : 12;
    static constexpr int HZ = (BZ - 1) * S + 3, HY = (2 * BY - 1) * S + 3, HX = (8 * BX - 1) * S + 3;
    static constexpr int HXP = (HX + 7) / 8 * 8;
    static constexpr int TILE_FLOATS = HZ * HY * HXP * VS;
    static constexpr int NPIECE = HZ * HY * HX * 2;
    static constexpr int PPT = (NPIECE + 255) / 256;
    static_assert(NT == 1 || NT == 2 || NT == 4, "COUT must be 16, 32 or 64");
    static_assert(MT % MG == 0, "block tile must split evenly over the M-groups");
    static_assert(PPT <= 32, "piece mask is 32 bits");
    static constexpr int tap_off(int tap) {  // LDS float offset of tap (kz,ky,kx); tap 27 = padding
        const int t = tap > 26 ? 26 : tap;
        return (((t / 9) * HY + (t / 3) % 3) * HXP + t % 3) * VS;
    }
};

template <int DT, int CIN, int COUT, int S, int BZ, int BY, int BX>
__global__ __launch_bounds__(256) void convg_mfma_kernel(
    const void* __restrict__ x,      // [CIN/8][Di][Hi][Wi][8] storage dtype DT
    const float* __restrict__ bp,    // [NCH][NT][14][64][4]
    const float* __restrict__ bias,  // [COUT]
    void* __restrict__ y,            // [COUT/8][Do][Ho][Wo][8] storage dtype DT
    int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
    using G = ConvG<CIN, COUT, S, BZ, BY, BX>;
    __shared__ __attribute__((aligned(16))) float tile[G::TILE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NT, mg = wave / G::NT;
    const int nbx = (Wo + 8 * BX - 1) / (8 * BX), nby = (Ho + 2 * BY - 1) / (2 * BY);
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ox0 = bx * 8 * BX, oy0 = by * 2 * BY, oz0 = bz * BZ;  // first output voxel of the block
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;  // halo origin (input coords)
    const size_t Vin = (size_t)Di * Hi * Wi, Vout = (size_t)Do * Ho * Wo;

    int goff[G::PPT], loff[G::PPT];
    unsigned inside = 0;
#pragma unroll
    for (int i = 0; i < G::PPT; ++i) {
        const int p = tid + i * 256;
        const int half = p & 1, v = p >> 1;
        const int hx = v % G::HX, t = v / G::HX;
        const int hy = t % G::HY, hz = t / G::HY;
        const int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = p < G::NPIECE && gz >= 0 && gz < Di && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
        goff[i] = ok ? (int)((((size_t)gz * Hi + gy) * Wi + gx) * 8 + half * 4) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = (p < G::NPIECE) ? ((hz * G::HY + hy) * G::HXP + hx) * G::VS + half * 4 : -1;
    }

    // A fragment: lane (r = lane&15 -> voxel (ry, rx) of the M-tile, g = lane>>4): k-step ks covers
    // taps 2ks (g>>1 == 0) and 2ks+1 (g>>1 == 1), channels 4(g&1)..+3 of the chunk
    const int r = lane & 15, g = lane >> 4, gh = g >> 1;
    const int ry = r >> 3, rx = r & 7;
    int abase[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
        abase[i] = (((tz * S) * G::HY + (2 * ty + ry) * S) * G::HXP + (8 * tx + rx) * S) * G::VS + (g & 1) * 4;
    }

    f32x4 acc[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 breg[G::KS];
    f32x4 stg[G::PPT];

#define MVS_LOAD_B(C)                                                                               \
    {                                                                                               \
        const f32x4* bsrc =                                                                         \
            reinterpret_cast<const f32x4*>(bp) + ((size_t)((C) * G::NT + nt) * G::KS) * 64 + lane;  \
        _Pragma("unroll") for (int ks = 0; ks < G::KS; ++ks) breg[ks] = bsrc[ks * 64];              \
    }
#define MVS_LOAD_A(C)                                                                               \
    {                                                                                               \
        const size_t plane = (size_t)(C) * Vin * 8;                                                 \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            stg[i] = St<DT>::load4(x, plane + goff[i]);                                             \
    }
#define MVS_STORE_A()                                                                               \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            if (loff[i] >= 0)                                                                       \
                *reinterpret_cast<f32x4*>(tile + loff[i]) =                                         \
                    ((inside >> i) & 1u) ? stg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};                    \
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

    // epilogue: D layout col n = lane&15 -> co = 16 nt + n; row m = 4*(lane>>4) + e -> voxel of tile
    const int n = lane & 15, co = 16 * nt + n;
    const float bv = bias[co];
    const size_t yplane = (size_t)(co >> 3) * Vout * 8 + (co & 7);
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
        const int gz = oz0 + tz;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 4 * (lane >> 4) + e;
            const int gy = oy0 + 2 * ty + (m >> 3), gx = ox0 + 8 * tx + (m & 7);
            if (gz < Do && gy < Ho && gx < Wo)
                St<DT>::store1(y, yplane + (((size_t)gz * Ho + gy) * Wo + gx) * 8, fmaxf(acc[i][e] + bv, 0.0f));
        }
    }
}

template <int DT, int CIN, int COUT, int S, int BZ, int BY, int BX>
static int run_convg(const void* x, void* y, const float* bp, const float* bias, int Di, int Hi, int Wi,
                     hipStream_t s) {
    const int Do = (Di - 1) / S + 1, Ho = (Hi - 1) / S + 1, Wo = (Wi - 1) / S + 1;
    if ((size_t)Di * Hi * Wi * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "convg_mfma: plane exceeds 31-bit offsets");
    const int nb = ((Wo + 8 * BX - 1) / (8 * BX)) * ((Ho + 2 * BY - 1) / (2 * BY)) * ((Do + BZ - 1) / BZ);
    convg_mfma_kernel<DT, CIN, COUT, S, BZ, BY, BX><<<nb, 256, 0, s>>>(x, bp, bias, y, Di, Hi, Wi, Do,
                                                                       Ho, Wo);
    return check_hip(hipGetLastError(), "convg_mfma launch");
}

// ---------------------------------------------------------------------------------------------
// Persistent form of convg_mfma_kernel for layers with many tiles per CU and few chunks (conv1:
// 7,680 tiles, one chunk of 56 MFMAs per wave -- less than a global-load round trip, which every
// one-tile block paid in the open).  Blocks loop over tiles and the chunk pipeline runs across tile
// boundaries: the next tile's first chunk is requested before this tile's last chunk of MFMAs.
// conv1: 0.072 -> 0.056 ms.
// ---------------------------------------------------------------------------------------------
template <int DT, int CIN, int COUT, int S, int BZ, int BY, int BX>
__global__ __launch_bounds__(256) void convg_persist_mfma_kernel(
    const void* __restrict__ x,      // [CIN/8][Di][Hi][Wi][8] storage dtype DT
    const float* __restrict__ bp,    // [NCH][NT][14][64][4]
    const float* __restrict__ bias,  // [COUT]
    void* __restrict__ y,            // [COUT/8][Do][Ho][Wo][8] storage dtype DT
    int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
    using G = ConvG<CIN, COUT, S, BZ, BY, BX>;
    __shared__ __attribute__((aligned(16))) float tile[G::TILE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NT, mg = wave / G::NT;
    const int nbx = (Wo + 8 * BX - 1) / (8 * BX), nby = (Ho + 2 * BY - 1) / (2 * BY);
    const int ntiles = nbx * nby * ((Do + BZ - 1) / BZ);
    const size_t Vin = (size_t)Di * Hi * Wi, Vout = (size_t)Do * Ho * Wo;

    // Persistent block: tiles blockIdx.x, blockIdx.x + gridDim.x, ...; the chunk pipeline runs across
    // tile boundaries (the next tile's first chunk is requested before this tile's last chunk of
    // MFMAs).  These layers have 1..8 chunks of only 56-224 MFMAs per wave, i.e. less than a
    // global-load round trip: as one-tile blocks every tile paid that latency in the open.
    // Staging bookkeeping: the decomposition of piece p = tid + 256 i into halo coordinates (divisions by
    // the non-power-of-two halo extents) is the same for every tile -- done once; per tile only a scalar
    // base offset and the scalar valid ranges of (hz, hy, hx) are left (interior tiles: no masks at all).
    int goff[G::PPT], loff[G::PPT], rel[G::PPT];
    unsigned pos[G::PPT];   // hz | hy << 8 | hx << 16
    unsigned inside = 0;
    int ox0 = 0, oy0 = 0, oz0 = 0;
#pragma unroll
    for (int i = 0; i < G::PPT; ++i) {
        const int p = tid + i * 256;
        const int half = p & 1, v = p >> 1;
        const int hx = v % G::HX, t = v / G::HX;
        const int hy = t % G::HY, hz = t / G::HY;
        const bool have = p < G::NPIECE;
        loff[i] = have ? ((hz * G::HY + hy) * G::HXP + hx) * G::VS + half * 4 : -1;
        rel[i] = have ? ((hz * Hi + hy) * Wi + hx) * 8 + half * 4 : 0;
        pos[i] = have ? (unsigned)(hz | (hy << 8) | (hx << 16)) : 0x00FFFFFFu;
    }
#define MVS_SETUP(TILE)                                                                             \
    {                                                                                               \
        int b_ = (TILE);                                                                            \
        const int bx_ = b_ % nbx; b_ /= nbx;                                                        \
        const int by_ = b_ % nby;                                                                   \
        const int bz_ = b_ / nby;                                                                   \
        ox0 = bx_ * 8 * BX; oy0 = by_ * 2 * BY; oz0 = bz_ * BZ;                                     \
        const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;                          \
        const int base_ = ((iz0 * Hi + iy0) * Wi + ix0) * 8;   /* may be negative; ok pieces are not */ \
        const int zl = max(0, -iz0), zh = min(G::HZ - 1, Di - 1 - iz0);                             \
        const int yl = max(0, -iy0), yh = min(G::HY - 1, Hi - 1 - iy0);                             \
        const int xl = max(0, -ix0), xh = min(G::HX - 1, Wi - 1 - ix0);                             \
        inside = 0;                                                                                 \
        if (zl == 0 && zh == G::HZ - 1 && yl == 0 && yh == G::HY - 1 && xl == 0 && xh == G::HX - 1) { \
            _Pragma("unroll") for (int i = 0; i < G::PPT; ++i) {                                    \
                const bool ok = pos[i] != 0x00FFFFFFu;                                              \
                goff[i] = ok ? base_ + rel[i] : 0;                                                  \
                inside |= ok ? (1u << i) : 0u;                                                      \
            }                                                                                       \
        } else {                                                                                    \
            _Pragma("unroll") for (int i = 0; i < G::PPT; ++i) {                                    \
                const int hz = pos[i] & 255, hy = (pos[i] >> 8) & 255, hx = (int)(pos[i] >> 16);    \
                const bool ok = hz >= zl && hz <= zh && hy >= yl && hy <= yh && hx >= xl && hx <= xh; \
                goff[i] = ok ? base_ + rel[i] : 0;                                                  \
                inside |= ok ? (1u << i) : 0u;                                                      \
            }                                                                                       \
        }                                                                                           \
    }

    // A fragment: lane (r = lane&15 -> voxel (ry, rx) of the M-tile, g = lane>>4): k-step ks covers
    // taps 2ks (g>>1 == 0) and 2ks+1 (g>>1 == 1), channels 4(g&1)..+3 of the chunk
    const int r = lane & 15, g = lane >> 4, gh = g >> 1;
    const int ry = r >> 3, rx = r & 7;
    int abase[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
        abase[i] = (((tz * S) * G::HY + (2 * ty + ry) * S) * G::HXP + (8 * tx + rx) * S) * G::VS + (g & 1) * 4;
    }

    f32x4 breg[G::KS];
    f32x4 stg[G::PPT];

#define MVS_LOAD_B(C)                                                                               \
    {                                                                                               \
        const f32x4* bsrc =                                                                         \
            reinterpret_cast<const f32x4*>(bp) + ((size_t)((C) * G::NT + nt) * G::KS) * 64 + lane;  \
        _Pragma("unroll") for (int ks = 0; ks < G::KS; ++ks) breg[ks] = bsrc[ks * 64];              \
    }
#define MVS_LOAD_A(C)                                                                               \
    {                                                                                               \
        const size_t plane = (size_t)(C) * Vin * 8;                                                 \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            stg[i] = St<DT>::load4(x, plane + goff[i]);                                             \
    }
#define MVS_STORE_A()                                                                               \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            if (loff[i] >= 0)                                                                       \
                *reinterpret_cast<f32x4*>(tile + loff[i]) =                                         \
                    ((inside >> i) & 1u) ? stg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};                    \
    }

    int tile_id = blockIdx.x;
    if (tile_id >= ntiles) return;
    MVS_SETUP(tile_id)
    MVS_LOAD_B(0)
    MVS_LOAD_A(0)
    MVS_STORE_A()
    __syncthreads();

    const int n = lane & 15, co = 16 * nt + n;
    const float bv = bias[co];
    const size_t yplane = (size_t)(co >> 3) * Vout * 8 + (co & 7);

#pragma unroll 1
    for (; tile_id < ntiles; tile_id += gridDim.x) {
        f32x4 acc[G::MPW];
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int cx0 = ox0, cy0 = oy0, cz0 = oz0;
        const int next_id = tile_id + gridDim.x;
        const bool has_next = next_id < ntiles;

#pragma unroll 1
        for (int c = 0; c < G::NCH; ++c) {
            if (c + 1 < G::NCH) {
                MVS_LOAD_A(c + 1)
            } else if (has_next) {
                MVS_SETUP(next_id)
                MVS_LOAD_A(0)
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
            if (c + 1 < G::NCH) {
                MVS_LOAD_B(c + 1)
                __syncthreads();
                MVS_STORE_A()
                __syncthreads();
            }
        }

        // epilogue: D layout col n = lane&15 -> co = 16 nt + n; row m = 4*(lane>>4) + e -> voxel of tile
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) {
            const int t = mg * G::MPW + i;
            const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
            const int gz = cz0 + tz;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 4 * (lane >> 4) + e;
                const int gy = cy0 + 2 * ty + (m >> 3), gx = cx0 + 8 * tx + (m & 7);
                if (gz < Do && gy < Ho && gx < Wo)
                    St<DT>::store1(y, yplane + (((size_t)gz * Ho + gy) * Wo + gx) * 8, fmaxf(acc[i][e] + bv, 0.0f));
            }
        }
        if (has_next) {
            if (G::NCH > 1) MVS_LOAD_B(0)
            __syncthreads();  // every wave is done reading this tile's last chunk
            MVS_STORE_A()     // next tile's chunk 0 (requested during this tile's last chunk)
            __syncthreads();
        }
    }
#undef MVS_SETUP
#undef MVS_LOAD_B
#undef MVS_LOAD_A
#undef MVS_STORE_A
}

template <int DT, int CIN, int COUT, int S, int BZ, int BY, int BX>
static int run_convg_persist(const void* x, void* y, const float* bp, const float* bias, int Di, int Hi, int Wi,
                     hipStream_t s) {
    using G = ConvG<CIN, COUT, S, BZ, BY, BX>;
    const int Do = (Di - 1) / S + 1, Ho = (Hi - 1) / S + 1, Wo = (Wi - 1) / S + 1;
    if ((size_t)Di * Hi * Wi * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "convg_mfma: plane exceeds 31-bit offsets");
    const int ntiles = ((Wo + 8 * BX - 1) / (8 * BX)) * ((Ho + 2 * BY - 1) / (2 * BY)) * ((Do + BZ - 1) / BZ);
    // persistent blocks: as many per CU as the LDS tile allows (at most 4)
    int per_cu = (int)((160 * 1024) / (sizeof(float) * G::TILE_FLOATS + 512));
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const int nblocks = persistent_blocks_per_cu_scale() * per_cu;
    // only when every block gets several tiles; otherwise the one-tile-per-block kernel balances
    // better and is leaner (conv3: 1,152 tiles measured 0.037 vs 0.032 ms)
    if (ntiles < 4 * nblocks) return run_convg<DT, CIN, COUT, S, BZ, BY, BX>(x, y, bp, bias, Di, Hi, Wi, s);
    const int nb = nblocks;
    convg_persist_mfma_kernel<DT, CIN, COUT, S, BZ, BY, BX><<<nb, 256, 0, s>>>(x, bp, bias, y, Di, Hi, Wi, Do,
                                                                       Ho, Wo);
    return check_hip(hipGetLastError(), "convg_mfma launch");
}

// =============================================================================================
// conv1 (8 -> 16, stride 2: models/mvsnet.py:38) as a z-MARCHING kernel, fp32 storage (round 3).
//
// conv1 reads the largest activation of the net (conv0's output: 126 MB at cfg2) to do 3.4 GFLOP: its floor is
// HBM and MFMA at once (20 / 21.6 us).  The tile kernel above re-stages a 3 x 9 x 33 halo per 1 x 4 x 16 outputs
// (1.74x the input through L2 -> LDS), a single 56-MFMA chunk per wave per tile: 0.052 ms even as persistent
// blocks.  Here a 512-thread block owns an 8 x 16 (y, x) column of the OUTPUT and marches along z: per output
// plane two new input planes enter a ring of five in LDS (17 x 33 voxels each, halo only in y / x: 1.10x), requested
// up to three steps ahead into three rotating register sets (bytes in flight are what the kernel needs, see
// conv0z16 in conv3d_mfma16.hip); wave w owns one 2 x 8 M-tile, the 14-k-step panel (one chunk) stays in registers:
// 56 MFMAs per wave and step, one block barrier per step.  Voxel stride 32 B (not the 48 B the tile kernels use
// for stride 2): with the row pitch = 1 (mod 8) sixteen-byte slots the lane groups of ds_read_b128 cover 16
// distinct slots ({0,4,8,12}, {2,6,10,14} for the second row, +1 for the other channel half).  Every VMEM
// instruction is unconditional (clamped plane, out-of-range store offset): hipcc then keeps counted vmcnt waits.
// Epilogue through a wave-private LDS strip into 16-byte stores (one per lane).  Same panel, same k order as
// convg_mfma_kernel: identical results.
// =============================================================================================
namespace c1z {
constexpr int TYO = 8, TXO = 16;                 // output tile: 4 x 2 M-tiles of 2 x 8
constexpr int HY = 2 * TYO + 1, HX = 2 * TXO + 1;
constexpr int PITCH = 73 * 4;                    // floats per row: 73 sixteen-byte slots (66 used)
constexpr int SLOT = HY * PITCH;                 // one input plane (8 channels): 19.9 KB
constexpr int RING = 5;
constexpr int THREADS = 512;
constexpr int NPIECE = HY * HX * 2;              // 16-byte pieces of one plane
constexpr int PPT = (NPIECE + THREADS - 1) / THREADS;   // 3
constexpr int STRIP = 16 * 20;                   // epilogue strip per wave: [voxel 16][20 floats]
}  // namespace c1z

__global__ __launch_bounds__(c1z::THREADS) void conv1z_mfma_kernel(
    const float* __restrict__ x,     // [1][Di][Hi][Wi][8]
    const float* __restrict__ bp,    // [1][1][14][64][4]
    const float* __restrict__ bias,  // [16]
    float* __restrict__ y,           // [2][Do][Ho][Wo][8]
    int Di, int Hi, int Wi, int Do, int Ho, int Wo, int ZC, int nbx, int nby) {
    using namespace c1z;
    __shared__ __attribute__((aligned(16))) float ring[RING * SLOT];
    __shared__ __attribute__((aligned(16))) float strips[8 * STRIP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b;
    {   // XCD k works through the k-th eighth of the (z chunk, row, column) sequence (blockIdx % 8 names the XCD)
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ox0 = bx * TXO, oy0 = by * TYO;
    const int za = bz * ZC, zb = min(za + ZC, Do);
    const int ix0 = 2 * ox0 - 1, iy0 = 2 * oy0 - 1;
    const size_t HW8 = (size_t)Hi * Wi * 8;

    // staging: piece p = tid + i * THREADS -> (hy, hx, half) of a plane
    size_t goff[PPT];
    int loff[PPT];
    unsigned okxy = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = tid + i * THREADS;
        const int half = p & 1, v = p >> 1;
        const int hx = v % HX, hy = v / HX;
        const int gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = p < NPIECE && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
        goff[i] = ok ? ((size_t)gy * Wi + gx) * 8 + half * 4 : 0;
        okxy |= ok ? (1u << i) : 0u;
        loff[i] = p < NPIECE ? hy * PITCH + hx * 8 + half * 4 : -1;
    }
    auto load_plane = [&](int gz, f32x4 (&st)[PPT]) {   // raw loads (plane clamped into the volume); masked when stored
        const size_t zo = (size_t)min(max(gz, 0), Di - 1) * HW8;
#pragma unroll
        for (int i = 0; i < PPT; ++i) st[i] = *reinterpret_cast<const f32x4*>(x + goff[i] + zo);
    };
    auto slot_of = [&](int gz) { return ((gz + 1) % RING) * SLOT; };   // gz >= -1
    auto store_plane = [&](int gz, const f32x4 (&st)[PPT]) {
        float* slot = ring + slot_of(gz);
        const bool zok = gz >= 0 && gz < Di;
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            if (loff[i] >= 0)
                *reinterpret_cast<f32x4*>(slot + loff[i]) = (zok && ((okxy >> i) & 1u)) ? st[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    // the panel in registers
    f32x4 breg[14];
#pragma unroll
    for (int ks = 0; ks < 14; ++ks) breg[ks] = reinterpret_cast<const f32x4*>(bp)[ks * 64 + lane];

    // A fragment: lane (r -> voxel (ry, rx) of the M-tile, g): taps 2ks (g>>1 == 0) / 2ks+1, channels 4(g&1)..+3
    const int r = lane & 15, g = lane >> 4, gh = g >> 1;
    const int tyw = wave >> 1, txw = wave & 1;
    const int abase = (2 * (2 * tyw + (r >> 3))) * PITCH + (2 * (8 * txw + (r & 7))) * 8 + (g & 1) * 4;
    int kin[14];   // within-plane offset of this lane's tap of k-step ks
#pragma unroll
    for (int ks = 0; ks < 14; ++ks) {
        const int t0 = 2 * ks, t1 = 2 * ks + 1 > 26 ? 26 : 2 * ks + 1;
        const int o0 = ((t0 / 3) % 3) * PITCH + (t0 % 3) * 8, o1 = ((t1 / 3) % 3) * PITCH + (t1 % 3) * 8;
        kin[ks] = abase + (gh ? o1 : o0);
    }
    const int n = lane & 15;
    const float bv = bias[n];
    float* strip = strips + wave * STRIP;
    // epilogue store role: lane -> piece (voxel = lane >> 2, channels 4 (lane & 3) ..)
    const int sv = lane >> 2, sq = lane & 3;
    const int sgy = oy0 + 2 * tyw + (sv >> 3), sgx = ox0 + 8 * txw + (sv & 7);
    const bool st_ok = sgy < Ho && sgx < Wo;
    const size_t Vout8 = (size_t)Do * Ho * Wo * 8;
    const unsigned st_off = (unsigned)((((size_t)(sq >> 1) * Vout8) + ((size_t)sgy * Wo + sgx) * 8 + (sq & 1) * 4) * 4);
    const unsigned zstep_b = (unsigned)((size_t)Ho * Wo * 8 * 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, (int)(unsigned)(2 * Vout8 * 4), 0x00020000);

    // prologue: input planes 2za-1, 2za, 2za+1 into the ring; the planes of steps za (2za+2, 2za+3) and za+1 in flight
    // (all seven planes requested back to back: one round trip in front of the first MFMA, not two)
    f32x4 a0[PPT], a1[PPT], b0[PPT], b1[PPT], c0[PPT], c1[PPT];
    {
        f32x4 t0[PPT];
        load_plane(2 * za - 1, c0);
        load_plane(2 * za, c1);
        load_plane(2 * za + 1, t0);
        load_plane(2 * za + 2, a0);
        load_plane(2 * za + 3, a1);
        load_plane(2 * za + 4, b0);
        load_plane(2 * za + 5, b1);
        store_plane(2 * za - 1, c0);
        store_plane(2 * za, c1);
        store_plane(2 * za + 1, t0);
    }
    __syncthreads();

    // one output plane zo: (n0, n1) receive input planes 2zo+6, 2zo+7 (first read in step zo+3), (w0, w1) hold planes
    // 2zo+2, 2zo+3 (requested two steps ago) and go into the slots of planes 2zo-3, 2zo-2, which no step reads any more
    auto step = [&](int zo, f32x4 (&w0)[PPT], f32x4 (&w1)[PPT], f32x4 (&n0)[PPT], f32x4 (&n1)[PPT]) {
        load_plane(2 * zo + 6, n0);
        load_plane(2 * zo + 7, n1);
        int sb[3];
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) sb[kz] = slot_of(2 * zo - 1 + kz);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            const int t0 = 2 * ks, t1 = 2 * ks + 1 > 26 ? 26 : 2 * ks + 1;
            const int s = (t0 / 9 == t1 / 9) ? sb[t0 / 9] : (gh ? sb[t1 / 9] : sb[t0 / 9]);
            const f32x4 a = *reinterpret_cast<const f32x4*>(ring + s + kin[ks]);
            const f32x4 bq = breg[ks];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq.w, acc, 0, 0, 0);
        }
        // epilogue: row m = 4 g + e of the M-tile, column n -> strip[m][n]; then one 16-byte piece per lane
#pragma unroll
        for (int e = 0; e < 4; ++e) strip[(4 * g + e) * 20 + n] = fmaxf(acc[e] + bv, 0.0f);
        __builtin_amdgcn_wave_barrier();
        const f32x4 o = *reinterpret_cast<const f32x4*>(strip + sv * 20 + sq * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4r, o), yrs,
                                               (int)((st_ok && zo < zb) ? st_off + (unsigned)zo * zstep_b : 0xFFFFFFF0u), 0, 0);
        __builtin_amdgcn_wave_barrier();
        store_plane(2 * zo + 2, w0);
        store_plane(2 * zo + 3, w1);
        __syncthreads();
    };
    int zo = za;
#pragma unroll 1
    for (; zo + 3 <= zb; zo += 3) {
        step(zo, a0, a1, c0, c1);
        step(zo + 1, b0, b1, a0, a1);
        step(zo + 2, c0, c1, b0, b1);
    }
    if (zo < zb) {   // the one or two planes left over (the register sets are back in their first order)
        step(zo, a0, a1, c0, c1);
        if (zo + 1 < zb) step(zo + 1, b0, b1, a0, a1);
    }
}

static int run_conv1z(const void* x, void* y, const float* bp, const float* bias, int Di, int Hi, int Wi, int Do, int Ho,
                      int Wo, hipStream_t s) {
    using namespace c1z;
    const int nbx = (Wo + TXO - 1) / TXO, nby = (Ho + TYO - 1) / TYO, ncol = nbx * nby;
    int cus = persistent_blocks_per_cu_scale();
    // z chunks (>= 4 output planes; each chunk re-reads one input plane and pays a prologue of about two steps): the
    // split that fills the last round of one-block-per-CU best
    int best = 1;
    double best_eff = 0.0;
    for (int nz = 1; nz <= (Do + 3) / 4; ++nz) {
        const int zc = (Do + nz - 1) / nz, nzc = (Do + zc - 1) / zc;
        const long nb = (long)ncol * nzc;
        const double eff = (double)nb / (double)(((nb + cus - 1) / cus) * cus) * zc / (zc + 2.5);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = nz; }
    }
    const int ZC = (Do + best - 1) / best, nzc = (Do + ZC - 1) / ZC;
    conv1z_mfma_kernel<<<ncol * nzc, THREADS, 0, s>>>(static_cast<const float*>(x), bp, bias, static_cast<float*>(y), Di, Hi,
                                                      Wi, Do, Ho, Wo, ZC, nbx, nby);
    return check_hip(hipGetLastError(), "conv1z_mfma launch");
}

template <int DT>
static int launch_convg_dt(int layer, const void* x, void* y, const float* bp, const float* bias, int Di,
                           int Hi, int Wi, hipStream_t s) {
    switch (layer) {
        case 1: {
            // fp32 storage, a volume whose (y, x) columns fill the chip: the z-marching kernel (MVS_CONV1Z=0/1 = never /
            // always); else the persistent tile kernel
            if constexpr (DT == MVS_F32) {
                static const int zm = [] {
                    const char* e = getenv("MVS_CONV1Z");
                    return e ? atoi(e) : -1;
                }();
                const int Do = (Di - 1) / 2 + 1, Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
                const long ncol = (long)((Wo + c1z::TXO - 1) / c1z::TXO) * ((Ho + c1z::TYO - 1) / c1z::TYO);
                if (zm != 0 && (size_t)Do * Ho * Wo * 16 * 4 < ((size_t)1 << 32) - 64 && (size_t)Di * Hi * Wi * 8 < ((size_t)1 << 31) &&
                    (zm == 1 || ncol * ((Do + 3) / 4) >= persistent_blocks_per_cu_scale() / 2))
                    return run_conv1z(x, y, bp, bias, Di, Hi, Wi, Do, Ho, Wo, s);
            }
            return run_convg_persist<DT, 8, 16, 2, 1, 2, 2>(x, y, bp, bias, Di, Hi, Wi, s);
        }
        case 2: return run_convg<DT, 16, 16, 1, 2, 4, 2>(x, y, bp, bias, Di, Hi, Wi, s);
        case 3: return run_convg<DT, 16, 32, 2, 1, 2, 2>(x, y, bp, bias, Di, Hi, Wi, s);
        case 4: return run_convg<DT, 32, 32, 1, 1, 2, 2>(x, y, bp, bias, Di, Hi, Wi, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "convg_mfma: layer %d not covered", layer);
    }
}

// layers 1..4 (conv1..conv4; conv5 / conv6: conv3d_small.hip)
int launch_convg_mfma(int layer, const void* x, void* y, const float* bp, const float* bias, int Di,
                      int Hi, int Wi, int dtype, hipStream_t s) {
    MVS_DISPATCH_DTYPE(dtype, (launch_convg_dt<DT>(layer, x, y, bp, bias, Di, Hi, Wi, s)))
}

// Host-side packing for convg: wfold [27][cin][cout] -> bp [cin/8][cout/16][14][64][4]
void pack_convg_weights(const float* wfold, int cin, int cout, float* bp) {
    const int nch = cin / 8, nt = cout / 16;
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < nt; ++t)
            for (int ks = 0; ks < 14; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const int g = lane >> 4, n = lane & 15;
                        const int tap = 2 * ks + (g >> 1);
                        const int ci = 8 * c + 4 * (g & 1) + j4, co = 16 * t + n;
                        const float v = tap < 27 ? wfold[((size_t)tap * cin + ci) * cout + co] : 0.0f;
                        bp[((((size_t)c * nt + t) * 14 + ks) * 64 + lane) * 4 + j4] = v;
                    }
}

size_t convg_panel_floats(int cin, int cout) { return (size_t)(cin / 8) * (cout / 16) * 14 * 64 * 4; }

// =============================================================================================
// fp32-MFMA transposed convolution: ConvTranspose3d k3 s2 p1 op1 (+folded BN) + ReLU + skip add,
// conv7 / conv9 / conv11 of CostRegNet (models/mvsnet.py:47-60, 69-71).
//
// Gather form per axis:  o = 2i - 1 + k  =>  even o = 2i: (k=1, input i);  odd o = 2i+1:
// (k=2, input i) and (k=0, input i+1).  The GEMM runs over the INPUT grid:
//   M : input voxels i (MFMA tile = 2(y) x 8(x)); each produces the 2x2x2 outputs 2i+p
//   classes (pz, py): 4 accumulator sets; their (z,y) taps: 1x1, 1x2, 2x1, 2x2 input offsets
//   N : (px, co) -- the x parity is folded into N, so both x taps (dx = 0, 1) are always
//       present: B[(dz,dy,dx),ci][(px,co)] = w[kz][ky][kx][ci][co] with kx = 1 / none for px=0
//       (dx = 0 / 1) and kx = 2 / 0 for px=1   -> 3/4 of the MFMA work is useful and Cout = 8
//       (conv11) fills the 16 MFMA columns
//   K : per chunk of 8 input channels 18 (tap, dx) pairs = 9 k-steps of 16 over the 4 classes
// The input halo tile (BZ+1 x 2BY+1 x 8BX+1 voxels) is staged once per chunk and serves all
// 8 output parities.  Epilogue: ReLU(acc + bias) + skip  (skip + relu(bn(deconv)), mvsnet.py:69).
// =============================================================================================
template <int CIN, int COUT, int BZ, int BY, int BX>
struct DeconvG {
    static constexpr int NTT = 2 * COUT / 16;  // N-tiles over (px, co)
    static constexpr int MG = 4 / NTT;
    static constexpr int MT = BZ * BY * BX;
    static constexpr int MPW = MT / MG;
    static constexpr int NCH = CIN / 8;
    static constexpr int KS = 9;
    static constexpr int VS = 8;
    static constexpr int HZ = BZ + 1, HY = 2 * BY + 1, HX = 8 * BX + 1;
    static constexpr int HXP = (HX + 7) / 8 * 8;
    static constexpr int IN_TILE_FLOATS = HZ * HY * HXP * VS;
    static constexpr int NPIECE = HZ * HY * HX * 2;
    static constexpr int PPT = (NPIECE + 255) / 256;
    // epilogue staging tile: one z parity of the block's 2BZ x 4BY x 16BX output voxels x COUT channels, rows padded
    // (16 floats per 8 voxels, row pitch = 16 mod 32) so that the accumulator scatter is conflict-free
    static constexpr int OZ = BZ, OY = 4 * BY, OX = 16 * BX;  // one z parity at a time
    static constexpr int RP0 = OX * COUT + 16 * (OX / 8);
    static constexpr int RP = (RP0 % 32 == 16) ? RP0 : RP0 + 16;
    static constexpr int OUT_TILE_FLOATS = OZ * OY * RP;
    static constexpr int TILE_FLOATS = IN_TILE_FLOATS > OUT_TILE_FLOATS ? IN_TILE_FLOATS : OUT_TILE_FLOATS;
    static constexpr int NUNIT = OZ * OY * OX * (COUT / 8);  // (plane, voxel) units of 8 channels
    static constexpr int UPT = (NUNIT + 255) / 256;
    static_assert(NTT == 1 || NTT == 2 || NTT == 4, "COUT must be 8, 16 or 32");
    static_assert(MT % MG == 0, "block tile must split evenly over the M-groups");
};

// deconv_step(ks) -> (class, kz, dz, ky, dy): mvs_internal.h (shared with conv11_prob.hip and the host packer)

template <int DT, int CIN, int COUT, int BZ, int BY, int BX>
__global__ __launch_bounds__(256) void deconvg_mfma_kernel(
    const void* __restrict__ x,      // [CIN/8][Di][Hi][Wi][8] storage dtype DT
    const float* __restrict__ bp,    // [NCH][NTT][9][64][4]
    const float* __restrict__ bias,  // [COUT]
    const void* __restrict__ skip,   // [COUT/8][2Di][2Hi][2Wi][8]
    void* __restrict__ y,            // [COUT/8][2Di][2Hi][2Wi][8]
    int Di, int Hi, int Wi) {
    using G = DeconvG<CIN, COUT, BZ, BY, BX>;
    __shared__ __attribute__((aligned(16))) float tile[G::TILE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NTT, mg = wave / G::NTT;
    const int nbx = (Wi + 8 * BX - 1) / (8 * BX), nby = (Hi + 2 * BY - 1) / (2 * BY);
    int b = blockIdx.x;
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
        const int p = tid + i * 256;
        const int half = p & 1, v = p >> 1;
        const int hx = v % G::HX, t = v / G::HX;
        const int hy = t % G::HY, hz = t / G::HY;
        const int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = p < G::NPIECE && gz < Di && gy < Hi && gx < Wi;
        goff[i] = ok ? (int)((((size_t)gz * Hi + gy) * Wi + gx) * 8 + half * 4) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = (p < G::NPIECE) ? ((hz * G::HY + hy) * G::HXP + hx) * G::VS + half * 4 : -1;
    }

    // lane (r -> input voxel (ry, rx) of the M-tile, g): dx = g>>1, channels 4(g&1)..+3
    const int r = lane & 15, g = lane >> 4;
    const int ry = r >> 3, rx = r & 7;
    int abase[G::MPW];
#pragma unroll
    for (int i = 0; i < G::MPW; ++i) {
        const int t = mg * G::MPW + i;
        const int tx = t % BX, ty = (t / BX) % BY, tz = t / (BX * BY);
        abase[i] = ((tz * G::HY + 2 * ty + ry) * G::HXP + 8 * tx + rx + (g >> 1)) * G::VS + (g & 1) * 4;
    }

    f32x4 acc[4][G::MPW];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < G::MPW; ++i) acc[c][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 breg[G::KS];
    f32x4 stg[G::PPT];

#define MVS_LOAD_B(C)                                                                               \
    {                                                                                               \
        const f32x4* bsrc =                                                                         \
            reinterpret_cast<const f32x4*>(bp) + ((size_t)((C) * G::NTT + nt) * G::KS) * 64 + lane; \
        _Pragma("unroll") for (int ks = 0; ks < G::KS; ++ks) breg[ks] = bsrc[ks * 64];              \
    }
#define MVS_LOAD_A(C)                                                                               \
    {                                                                                               \
        const size_t plane = (size_t)(C) * Vin * 8;                                                 \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            stg[i] = St<DT>::load4(x, plane + goff[i]);                                             \
    }
#define MVS_STORE_A()                                                                               \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < G::PPT; ++i)                                          \
            if (loff[i] >= 0)                                                                       \
                *reinterpret_cast<f32x4*>(tile + loff[i]) =                                         \
                    ((inside >> i) & 1u) ? stg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};                    \
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
            const DeconvStep st = deconv_step(ks);
            const int koff = (st.dz * G::HY + st.dy) * G::HXP * G::VS;
            f32x4 a[G::MPW];
#pragma unroll
            for (int i = 0; i < G::MPW; ++i) a[i] = *reinterpret_cast<const f32x4*>(tile + abase[i] + koff);
            const f32x4 bq = breg[ks];
#pragma unroll
            for (int i = 0; i < G::MPW; ++i) acc[st.cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[st.cls][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MPW; ++i) acc[st.cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[st.cls][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MPW; ++i) acc[st.cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[st.cls][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MPW; ++i) acc[st.cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[st.cls][i], 0, 0, 0);
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

    // epilogue: col n -> (px, co); row m -> input voxel of the tile; class -> (pz, py).
    // ReLU(acc + bias) is scattered into an LDS tile laid out [oz][oy][ox][co]; then every thread
    // owns whole voxels of one C8 plane (8 channels = 32 B) and does the skip add with 16-byte
    // loads / stores that are contiguous across the wave (the scalar form cost 0.036 of conv11's
    // 0.090 ms in exposed skip-load latency and 4-byte stores).
    // Two passes, one per output z parity (halves the staging tile: conv11 43 -> 21.5 KB, so a fourth
    // block fits per CU); in each pass the skip values are requested first, the scatter and its
    // barrier run while they are in flight.
    const int Do = 2 * Di;
    const int nn = 16 * nt + (lane & 15);
    const int px = nn / COUT, co = nn % COUT;
    const float bv = bias[co];
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
        // units of this pass: (plane, oz_l in [0,BZ), oy, ox), 8 channels each
        size_t uo[G::UPT];
        int usrc[G::UPT];
        f32x4 sk0[G::UPT], sk1[G::UPT];
        float sk16[G::UPT][8];
#pragma unroll
        for (int j = 0; j < G::UPT; ++j) {
            const int u = tid + j * 256;
            const int ox = u % G::OX;
            int t = u / G::OX;
            const int oy = t % G::OY; t /= G::OY;
            const int ozl = t % BZ, pl = t / BZ;
            const int gz = 2 * (iz0 + ozl) + pz, gy = 2 * iy0 + oy, gx = 2 * ix0 + ox;
            const bool ok = u < G::NUNIT && gz < Do && gy < Ho && gx < Wo;
            usrc[j] = ok ? (ozl * G::OY + oy) * G::RP + ox * COUT + (ox >> 3) * 16 + pl * 8 : -1;
            uo[j] = ok ? ((size_t)pl * Vout + ((size_t)gz * Ho + gy) * Wo + gx) * 8 : 0;
            if constexpr (DT == MVS_F32) {
                sk0[j] = St<DT>::load4(skip, uo[j]);
                sk1[j] = St<DT>::load4(skip, uo[j] + 4);
            } else {
                load8_16<DT>(skip, uo[j], sk16[j]);
            }
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
                    tile[(tz * G::OY + 2 * ly + py) * G::RP + ox * COUT + (ox >> 3) * 16 + co] =
                        fmaxf(acc[2 * pz + py][i][e] + bv, 0.0f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G::UPT; ++j) {
            if (usrc[j] < 0) continue;
            const float* src = tile + usrc[j];
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
            if constexpr (DT == MVS_F32) {
                St<DT>::store4(y, uo[j], lo + sk0[j]);
                St<DT>::store4(y, uo[j] + 4, hi + sk1[j]);
            } else {
                const float v[8] = {lo.x + sk16[j][0], lo.y + sk16[j][1], lo.z + sk16[j][2], lo.w + sk16[j][3],
                                    hi.x + sk16[j][4], hi.y + sk16[j][5], hi.z + sk16[j][6], hi.w + sk16[j][7]};
                store8_16<DT>(y, uo[j], v);
            }
        }
    }
}

template <int DT, int CIN, int COUT, int BZ, int BY, int BX>
static int run_deconvg(const void* x, const void* skip, void* y, const float* bp, const float* bias,
                       int Di, int Hi, int Wi, hipStream_t s) {
    if ((size_t)Di * Hi * Wi * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "deconvg_mfma: plane exceeds 31-bit offsets");
    const int nb = ((Wi + 8 * BX - 1) / (8 * BX)) * ((Hi + 2 * BY - 1) / (2 * BY)) * ((Di + BZ - 1) / BZ);
    deconvg_mfma_kernel<DT, CIN, COUT, BZ, BY, BX><<<nb, 256, 0, s>>>(x, bp, bias, skip, y, Di, Hi, Wi);
    return check_hip(hipGetLastError(), "deconvg_mfma launch");
}

template <int DT>
static int launch_deconvg_dt(int layer, const void* x, const void* skip, void* y, const float* bp,
                             const float* bias, int Di, int Hi, int Wi, hipStream_t s) {
    switch (layer) {
        case 8: return run_deconvg<DT, 32, 16, 1, 4, 1>(x, skip, y, bp, bias, Di, Hi, Wi, s);
        case 9: return run_deconvg<DT, 16, 8, 1, 4, 2>(x, skip, y, bp, bias, Di, Hi, Wi, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "deconvg_mfma: layer %d not covered", layer);
    }
}

// layers 8, 9 (conv9, conv11; conv7: conv3d_small.hip)
int launch_deconvg_mfma(int layer, const void* x, const void* skip, void* y, const float* bp,
                        const float* bias, int Di, int Hi, int Wi, int dtype, hipStream_t s) {
    MVS_DISPATCH_DTYPE(dtype, (launch_deconvg_dt<DT>(layer, x, skip, y, bp, bias, Di, Hi, Wi, s)))
}

// Host-side packing for deconvg: wfold [27][cin][cout] -> bp [cin/8][2*cout/16][9][64][4]
void pack_deconvg_weights(const float* wfold, int cin, int cout, float* bp) {
    const int nch = cin / 8, ntt = 2 * cout / 16;
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < ntt; ++t)
            for (int ks = 0; ks < 9; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const DeconvStep st = deconv_step(ks);
                        const int g = lane >> 4, n = lane & 15, dx = g >> 1;
                        const int ci = 8 * c + 4 * (g & 1) + j4;
                        const int nn = 16 * t + n, px = nn / cout, co = nn % cout;
                        const int kx = px == 0 ? (dx == 0 ? 1 : -1) : (dx == 0 ? 2 : 0);
                        const float v = kx < 0 ? 0.0f
                                               : wfold[((size_t)(st.kz * 9 + st.ky * 3 + kx) * cin + ci) * cout + co];
                        bp[((((size_t)c * ntt + t) * 9 + ks) * 64 + lane) * 4 + j4] = v;
                    }
}

}  // namespace mvs
