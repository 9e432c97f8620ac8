// conv0_wino44.hip -- conv0 (32 -> 8, models/mvsnet.py:36) with Winograd F(4,3) along z AND y on the fp32
// 4x4x1 MFMA (gfx950): a quarter of the direct form's multiplications, half of conv0_wino43.hip's.
//
// conv0 sits at the power limit (profiles/r02_conv0_clocks.txt): taking stalls out of its instruction stream
// buys nothing, fewer MFMAs do.  The 4 x 4 (z, y) outputs of a column come from a 6 x 6 block of transformed
// values  U = B^T d B  (B^T of F(4,3), the matrix conv0_wino43.hip applies along z), multiplied elementwise --
// per x tap and input channel -- with the weights  W[t][v] = sum_kz sum_ky G[t][kz] G[v][ky] w[kz][ky][kx]
// (host, double arithmetic), and  Y = A^T M A.  Per 4 x 4 x 32 tile and chunk of 8 channels: 6 x 6
// transformed rows of 32 positions x 3 x-taps = 864 MFMAs instead of 1,728.
//
//   block : 384 threads = 6 waves on a 4 x 4 x 32 tile; wave t owns transformed z-plane t: three units of
//           2 transformed rows x 32 x (each row v has its own weights: the B operand address depends on
//           the lane's row)
//   LDS   : the 6 x 6 x 34 transformed halo of one chunk at a 32-byte voxel stride (the two 16-byte halves
//           of a voxel swap places in every other group of 8 x: conflict-free ds_read_b128, checked by
//           brute force) = 39.2 KB, + the chunk's 36 x 3 weight slices = 27.6 KB: 66.8 KB, two blocks per CU
//   stage : pass 1 as in conv0_wino43.hip -- a thread owns a z-column (row, x, half), loads its 6 raw planes,
//           writes the 6 z-transformed values; barrier; pass 2 IN PLACE -- the same thread index now owns the
//           y-column (plane, x, half), reads its 6 rows and writes the 6 y-transformed rows back (6 rows in,
//           6 rows out: that is why F(4,3) and not F(2,3) along y); barrier
//   end   : the 36 partial results of a column go through LDS once per tile; a thread then owns
//           (output row, x, 4 channels), applies A^T along y per plane, then along z, + bias, ReLU, stores.
// Error against an fp64 convolution (numpy model of this arithmetic on variance-like inputs): 4e-6 of
// max |y| (F(4,3) along z alone: 8e-7, fmaf chain: 3e-7).
#include "mvs_internal.h"
#include "storage.h"

namespace mvs {

namespace c44 {
constexpr int TZ = 4, TY = 4, TX = 32, NTHR = 384;
constexpr int HY = TY + 2, HX = TX + 2;
constexpr int NP = 6;                            // transformed planes (z) = transformed rows (y)
constexpr int VS = 8;                            // floats per voxel in LDS
constexpr int PLANE = HY * HX * VS;              // 1632 floats
constexpr int TILE_FLOATS = NP * PLANE;          // 39,168 B
constexpr int NCOL = HY * HX * 2;                // 16-byte pieces of one plane: (row | plane, x, half) = 408
constexpr int CPT = (NCOL + NTHR - 1) / NTHR;    // 2
constexpr int WROW = 3 * 2 * 2 * 16;             // weight floats of one (t, v): [kx][half][nt][j][k]
constexpr int WPLANE = NP * WROW;                // per transformed plane t
constexpr int BW_FLOATS = NP * WPLANE;           // per chunk: 6,912 floats
constexpr int WPIECES = BW_FLOATS / 4;           // 1,728 16-byte pieces
constexpr int WPT = (WPIECES + NTHR - 1) / NTHR; // 5
constexpr int EXS = 8;                           // floats per position in the exchange tile [t][v][x]
static_assert(NP * NP * TX * EXS <= TILE_FLOATS, "exchange tile must fit in the input tile");

// float offset of the 16-byte half `half` of halo voxel (row-major index rowbase, column hx)
__device__ __forceinline__ int voxel_off(int rowbase, int hx, int half) {
    return (rowbase * HX + hx) * VS + ((half ^ ((hx >> 3) & 1)) << 2);
}
// B^T of F(4,3) on six values (f32x4 = 4 channels at once)
__device__ __forceinline__ void bt43(const f32x4 (&d)[6], f32x4 (&u)[6]) {
    const f32x4 t1 = d[4] - 4.0f * d[2], t2 = d[3] - 4.0f * d[1];
    const f32x4 t3 = d[4] - d[2], t4 = 2.0f * (d[3] - d[1]);
    u[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
    u[1] = t1 + t2;
    u[2] = t1 - t2;
    u[3] = t3 + t4;
    u[4] = t3 - t4;
    u[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
// row q (0..3) of A^T of F(4,3)
__device__ __forceinline__ f32x4 at43(const f32x4 (&m)[6], int q) {
    const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    switch (q) {
        case 0: return (m[0] + s12) + s34;
        case 1: return d12 + 2.0f * d34;
        case 2: return s12 + 4.0f * s34;
        default: return (d12 + 8.0f * d34) + m[5];
    }
}
}  // namespace c44

template <int DT>
__global__ __launch_bounds__(c44::NTHR, 3) void conv0_w44_mfma_kernel(
    const void* __restrict__ x,      // [4][D][H][W][8] storage dtype DT
    const float* __restrict__ bw,    // [4 chunks][6 t][6 v][3 kx][2 halves][2 nt][4 j][4 k]
    const float* __restrict__ bias,  // [8]
    void* __restrict__ y,            // [D][H][W][8] storage dtype DT
    int D, int H, int W) {
    using namespace c44;
    __shared__ __attribute__((aligned(16))) float tile[TILE_FLOATS + BW_FLOATS];
    float* wlds = tile + TILE_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY, nbz = (D + TZ - 1) / TZ;
    // block -> tile: as conv0_wino43.hip (each XCD owns a band of tile rows and walks it z-fastest)
    int bx, by, bz;
    if (nby % 8 == 0 && gridDim.x % 8 == 0) {
        const int xcd = blockIdx.x & 7, rows = nby >> 3;
        int i = blockIdx.x >> 3;
        bz = i % nbz; i /= nbz;
        by = xcd * rows + i % rows;
        bx = i / rows;
    } else {
        int b = blockIdx.x;
        bx = b % nbx; b /= nbx;
        by = b % nby;
        bz = b / nby;
    }
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    const size_t HW8 = (size_t)H * W * 8, V8 = (size_t)D * HW8;

    // staging bookkeeping: piece col = tid + i * NTHR -> (a, hx, half); a = halo row in pass 1, plane in pass 2
    int goff[CPT], l1[CPT], l2[CPT];
    unsigned okxy = 0, have = 0;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int col = tid + i * NTHR;
        const int half = col & 1, v = col >> 1;
        const int hx = v % HX, a = v / HX;
        const int gy = y0 + a - 1, gx = x0 + hx - 1;
        const bool in = col < NCOL;
        const bool ok = in && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? (int)(((size_t)gy * W + gx) * 8 + half * 4) : 0;
        okxy |= ok ? (1u << i) : 0u;
        have |= in ? (1u << i) : 0u;
        l1[i] = in ? voxel_off(a, hx, half) : 0;             // + t * PLANE : row a of plane t
        l2[i] = in ? a * PLANE + voxel_off(0, hx, half) : 0; // + r * HX * VS : row r of plane a
    }
    size_t zoff[NP];
    bool zok[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int gz = z0 - 1 + q;
        zok[q] = gz >= 0 && gz < D;
        zoff[q] = zok[q] ? (size_t)gz * HW8 : 0;
    }
    // fp32 storage: raw buffer loads; the plane offset travels in the scalar offset, a piece outside the volume
    // gets a vector offset beyond the buffer's range (hardware zeros, no selects in the transform)
    unsigned gbyte[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) gbyte[i] = ((okxy >> i) & 1u) ? (unsigned)goff[i] * 4u : 0x80000000u;

    // MFMA roles: wave = transformed plane t; unit i = transformed rows 2i, 2i+1; lane -> (row, xl)
    const int row = lane >> 5, xl = lane & 31;
    int aoff[3];                                   // x-tap kx, half 0 (half 1: ^ 4)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aoff[kx] = voxel_off(0, xl + kx, 0);
    const float* ab[3];
    const float* wb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ab[i] = tile + wave * PLANE + (2 * i + row) * HX * VS;
        wb[i] = wlds + wave * WPLANE + (2 * i + row) * WROW + (lane & 3) * 4;
    }

    f32x4 acc[3][2];  // [unit][nt]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 stg[CPT][NP];
    f32x4 wst[WPT];
    auto load_chunk = [&](int c) {
        const size_t plane = (size_t)c * V8;
        if constexpr (DT == MVS_F32) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(static_cast<const float*>(x) + plane), (short)0, (int)(V8 * 4), 0x00020000);
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    stg[i][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rs, (int)(zok[q] ? gbyte[i] : 0x80000000u), (int)(unsigned)(zoff[q] * 4), 0));
        } else {
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int q = 0; q < NP; ++q) stg[i][q] = St<DT>::load4(x, plane + zoff[q] + goff[i]);
        }
        const f32x4* wsrc = reinterpret_cast<const f32x4*>(bw) + (size_t)c * WPIECES;
#pragma unroll
        for (int i = 0; i < WPT; ++i) wst[i] = wsrc[min(tid + i * NTHR, WPIECES - 1)];
    };
    auto store_z = [&]() {      // pass 1: z transform of the staged columns, weights
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            if ((have >> i) & 1u) {
                const bool in = (okxy >> i) & 1u;
                constexpr bool hw0 = DT == MVS_F32;  // zeros already delivered by the buffer loads
                f32x4 d[NP], u[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) d[q] = (hw0 || (in && zok[q])) ? stg[i][q] : zero;
                bt43(d, u);
#pragma unroll
                for (int t = 0; t < NP; ++t) *reinterpret_cast<f32x4*>(tile + t * PLANE + l1[i]) = u[t];
            }
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (tid + i * NTHR < WPIECES) reinterpret_cast<f32x4*>(wlds)[tid + i * NTHR] = wst[i];
    };
    auto transform_y = [&]() {  // pass 2, in place: the same piece index now names (plane, x, half)
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            if ((have >> i) & 1u) {
                f32x4 d[NP], u[NP];
#pragma unroll
                for (int r = 0; r < NP; ++r) d[r] = *reinterpret_cast<const f32x4*>(tile + l2[i] + r * HX * VS);
                bt43(d, u);
#pragma unroll
                for (int r = 0; r < NP; ++r) *reinterpret_cast<f32x4*>(tile + l2[i] + r * HX * VS) = u[r];
            }
    };
    auto chunk_mfmas = [&]() {
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ao = (kx == 0 ? aoff[0] : kx == 1 ? aoff[1] : aoff[2]) ^ (half << 2);
                const int wo = (kx * 2 + half) * 32;
                f32x4 xa[3], b0[3], b1[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    xa[i] = *reinterpret_cast<const f32x4*>(ab[i] + ao);
                    b0[i] = *reinterpret_cast<const f32x4*>(wb[i] + wo);
                    b1[i] = *reinterpret_cast<const f32x4*>(wb[i] + wo + 16);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[i][k], b0[i][k], acc[i][0], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[i][k], b1[i][k], acc[i][1], 0, 0, 0);
                }
            }
    };

    load_chunk(0);
    store_z();
    __syncthreads();
    transform_y();
    __syncthreads();

#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (c < 3) load_chunk(c + 1);
        chunk_mfmas();
        if (c < 3) {
            __syncthreads();  // every wave is done reading chunk c's planes and weights
            store_z();
            __syncthreads();
            transform_y();
            __syncthreads();
        }
    }

    // exchange: D layout lane 4*blk + j, register r -> position 4*blk + r of the unit (row * 32 + x), channel
    // 4*nt + j.  ex[t][v][x][channel]
    __syncthreads();
    {
        const int blk = lane >> 2, j = lane & 3;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pos = 4 * blk + r;
                float* e = tile + ((wave * NP + 2 * i + (pos >> 5)) * TX + (pos & 31)) * EXS;
                e[j] = acc[i][0][r];
                e[4 + j] = acc[i][1][r];
            }
    }
    __syncthreads();
    if (tid >= 256) return;
    const int oy = tid >> 6, ox = (tid >> 1) & 31, ch = tid & 1;   // output row, x, channels 4 ch .. + 3
    const int gy = y0 + oy, gx = x0 + ox;
    if (gy >= H || gx >= W) return;
    f32x4 Y[NP];
#pragma unroll
    for (int t = 0; t < NP; ++t) {     // A^T along y, per transformed plane
        f32x4 m[NP];
#pragma unroll
        for (int v = 0; v < NP; ++v) m[v] = *reinterpret_cast<const f32x4*>(tile + ((t * NP + v) * TX + ox) * EXS + ch * 4);
        Y[t] = oy == 0 ? at43(m, 0) : oy == 1 ? at43(m, 1) : oy == 2 ? at43(m, 2) : at43(m, 3);
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + ch * 4);
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < TZ; ++q) {     // A^T along z
        if (z0 + q >= D) break;
        const f32x4 v = __builtin_elementwise_max(at43(Y, q) + bv, zero);
        St<DT>::store4(y, (((size_t)(z0 + q) * H + gy) * W + gx) * 8 + ch * 4, v);
    }
}

template <int DT>
static int run_conv0_w44(const void* x, void* y, const float* bw, const float* bias, int D, int H, int W,
                         hipStream_t s) {
    using namespace c44;
    const int nb = ((W + TX - 1) / TX) * ((H + TY - 1) / TY) * ((D + TZ - 1) / TZ);
    conv0_w44_mfma_kernel<DT><<<nb, NTHR, 0, s>>>(x, bw, bias, y, D, H, W);
    return check_hip(hipGetLastError(), "conv0_w44_mfma launch");
}

int launch_conv0_wino44(const void* x, void* y, const float* bw, const float* bias, int D, int H, int W,
                        int dtype, hipStream_t s) {
    if ((size_t)D * H * W * 8 * 4 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "conv0_wino44: plane of %zu bytes exceeds 31-bit buffer offsets",
                    (size_t)D * H * W * 8 * 4);
    MVS_DISPATCH_DTYPE(dtype, (run_conv0_w44<DT>(x, y, bw, bias, D, H, W, s)))
}

// wfold [27][32][8] (tap = kz*9 + ky*3 + kx) -> bw [4 chunks][6 t][6 v][3 kx][2 halves][2 nt][4 j][4 k] with the
// z and y taps transformed by G of F(4,3) (double arithmetic, rounded once)
void pack_conv0_wino44_weights(const float* wfold, float* bw) {
    static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int c = 0; c < 4; ++c)
        for (int t = 0; t < 6; ++t)
            for (int v = 0; v < 6; ++v)
                for (int kx = 0; kx < 3; ++kx)
                    for (int half = 0; half < 2; ++half)
                        for (int nt = 0; nt < 2; ++nt)
                            for (int j = 0; j < 4; ++j)
                                for (int k = 0; k < 4; ++k) {
                                    const int ci = 8 * c + 4 * half + k, co = 4 * nt + j;
                                    double g = 0.0;
                                    for (int kz = 0; kz < 3; ++kz)
                                        for (int ky = 0; ky < 3; ++ky)
                                            g += G[t][kz] * G[v][ky] *
                                                 (double)wfold[((size_t)(kz * 9 + ky * 3 + kx) * 32 + ci) * 8 + co];
                                    bw[(((((((size_t)c * 6 + t) * 6 + v) * 3 + kx) * 2 + half) * 2 + nt) * 4 + j) * 4 + k] = (float)g;
                                }
}

}  // namespace mvs
