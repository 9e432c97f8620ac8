// attic: conv0 on v_mfma_f32_16x16x4_f32 in the Toeplitz 'pair' form and its 8-wave split-K variant
// (measured 0.64 / 0.69 ms vs 0.57 ms for the 4x4x1 form at cfg2).  Not built.
namespace c0 {
constexpr int TZ = 2, TY = 8, TX = 32;              // output tile
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;  // halo tile
constexpr int VS = 12;                              // floats per voxel in LDS (8 data + 4 pad)
constexpr int TILE_FLOATS = HZ * HY * HX * VS;      // 16320 floats = 65280 B
constexpr int NPIECE = HZ * HY * HX * 2;            // 16-byte pieces per chunk (2720)
constexpr int PIECES_PER_THREAD = (NPIECE + 255) / 256;  // 11
constexpr int KS = 18;                              // k-steps of 16 per chunk (36 taps x 8 ci / 16)
}  // namespace c0

__global__ __launch_bounds__(256, 2) void conv0_pair_mfma_kernel(
    const float* __restrict__ x,     // [4][D][H][W][8]
    const float* __restrict__ bp,    // [4 chunks][18 k-steps][64 lanes][4]  packed Toeplitz B
    const float* __restrict__ bias,  // [8]
    float* __restrict__ y,           // [D][H][W][8]
    int D, int H, int W) {
    using namespace c0;
    __shared__ __attribute__((aligned(16))) float tile[TILE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    const size_t V = (size_t)D * H * W;

    // fill bookkeeping: piece -> (halo voxel, half).  Loads are unconditional (out-of-volume
    // pieces read offset 0 and are zeroed by `inside` when written to LDS): a predicated load
    // would make the compiler fall back to s_waitcnt vmcnt(0) and serialise the pipeline below.
    int goff[PIECES_PER_THREAD];
    int loff[PIECES_PER_THREAD];
    unsigned inside = 0;
#pragma unroll
    for (int i = 0; i < PIECES_PER_THREAD; ++i) {
        const int p = tid + i * 256;
        const int half = p & 1, v = p >> 1;
        const int hx = v % HX, t = v / HX;
        const int hy = t % HY, hz = t / HY;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = p < NPIECE && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? (int)((((size_t)gz * H + gy) * W + gx) * 8 + half * 4) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = (p < NPIECE) ? v * VS + half * 4 : -1;
    }

    // A-fragment addressing: lane (r = lane&15: pair index, g = lane>>4): k = (tap 2ks+(g>>1),
    // ci 4(g&1)+j); halo x of the tap = 2r + kx' with kx' = kx0(ks) + (g>>1)
    const int r = lane & 15, g = lane >> 4;
    const int zt = wave >> 1, yt0 = 4 * (wave & 1);
    const int lane_off = ((zt * HY + yt0) * HX + 2 * r + (g >> 1)) * VS + (g & 1) * 4;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline over the 4 K-chunks: the global loads of chunk c+1's halo tile are issued
    // before chunk c's MFMAs and written to LDS after them; chunk c+1's B panel is fetched while
    // the block drains through the barriers.
    f32x4 breg[KS];
    f32x4 stg[PIECES_PER_THREAD];
#define MVS_LOAD_B(C)                                                                         \
    {                                                                                         \
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(bp) + (size_t)(C) * KS * 64 + lane; \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) breg[ks] = bsrc[ks * 64];           \
    }
#define MVS_LOAD_A(C)                                                                         \
    {                                                                                         \
        const float* plane = x + (size_t)(C) * V * 8;                                         \
        _Pragma("unroll") for (int i = 0; i < PIECES_PER_THREAD; ++i)                         \
            stg[i] = *reinterpret_cast<const f32x4*>(plane + goff[i]);                        \
    }
#define MVS_STORE_A()                                                                         \
    {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < PIECES_PER_THREAD; ++i)                         \
            if (loff[i] >= 0)                                                                 \
                *reinterpret_cast<f32x4*>(tile + loff[i]) =                                   \
                    ((inside >> i) & 1u) ? stg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};              \
    }

    MVS_LOAD_B(0)
    MVS_LOAD_A(0)
    MVS_STORE_A()
    __syncthreads();

    const float* abase = tile + lane_off;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
#if MVS_ABLATE != 3
        if (c < 3) MVS_LOAD_A(c + 1)
#endif
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int tap0 = 2 * ks;  // taps (kz, ky, kx') with kx' = tap % 4
            const int kz = tap0 / 12, ky = (tap0 / 4) % 3, kx0 = tap0 % 4;
            const int koff = ((kz * HY + ky) * HX + kx0) * VS;
            f32x4 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(abase + koff + i * HX * VS);
            const f32x4 bq = breg[ks];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[i], 0, 0, 0);
        }
#if MVS_ABLATE != 2 && MVS_ABLATE != 3
        if (c < 3) {
            MVS_LOAD_B(c + 1)
            __syncthreads();  // every wave is done reading chunk c's tile
            MVS_STORE_A()
            __syncthreads();
        }
#endif
    }
#undef MVS_LOAD_B
#undef MVS_LOAD_A
#undef MVS_STORE_A

    // epilogue: D layout col n = lane&15 = (j, co), row m = 4*(lane>>4) + e = pair index
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
#if MVS_ABLATE == 1
            if (gz < D && gy < H && gx < W && acc[i][e] == 12345.678f) {
#else
            if (gz < D && gy < H && gx < W) {
#endif
                const float v = fmaxf(acc[i][e] + bv, 0.0f);  // ReLU(BN(conv)) with BN folded
                y[(((size_t)gz * H + gy) * W + gx) * 8 + co] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 8-wave variant: same tile, staging and GEMM as above, but the block has 512 threads and the K
// range of every chunk is split between two wave groups (waves 0-3: k-steps 0..8, waves 4-7:
// k-steps 9..17) that work on the same 16 M-tiles; the partial accumulators of the second group
// are added through LDS at the end.  Halving the per-wave B panel (36 VGPRs) brings the kernel
// under 128 VGPRs, so 2 blocks x 8 waves = 4 waves per SIMD keep the MFMA pipe fed while other
// waves stage, wait at barriers or run the epilogue.
// ---------------------------------------------------------------------------------------------
namespace c0 {
constexpr int KSH = KS / 2;                               // k-steps per wave group (9)
constexpr int PIECES_PER_THREAD8 = (NPIECE + 511) / 512;  // 6
}  // namespace c0

__global__ __launch_bounds__(512, 4) void conv0_pair_mfma8_kernel(
    const float* __restrict__ x, const float* __restrict__ bp, const float* __restrict__ bias,
    float* __restrict__ y, int D, int H, int W) {
    using namespace c0;
    __shared__ __attribute__((aligned(16))) float tile[TILE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = wave >> 2, mg = wave & 3;
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    const size_t V = (size_t)D * H * W;

    int goff[PIECES_PER_THREAD8];
    int loff[PIECES_PER_THREAD8];
    unsigned inside = 0;
#pragma unroll
    for (int i = 0; i < PIECES_PER_THREAD8; ++i) {
        const int p = tid + i * 512;
        const int half = p & 1, v = p >> 1;
        const int hx = v % HX, t = v / HX;
        const int hy = t % HY, hz = t / HY;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = p < NPIECE && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? (int)((((size_t)gz * H + gy) * W + gx) * 8 + half * 4) : 0;
        inside |= ok ? (1u << i) : 0u;
        loff[i] = (p < NPIECE) ? v * VS + half * 4 : -1;
    }

    const int r = lane & 15, g = lane >> 4;
    const int zt = mg >> 1, yt0 = 4 * (mg & 1);
    const int lane_off = ((zt * HY + yt0) * HX + 2 * r + (g >> 1)) * VS + (g & 1) * 4;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 breg[KSH];
    f32x4 stg[PIECES_PER_THREAD8];
#define MVS_LOAD_B(C)                                                                              \
    {                                                                                              \
        const f32x4* bsrc =                                                                        \
            reinterpret_cast<const f32x4*>(bp) + ((size_t)(C) * KS + kh * KSH) * 64 + lane;        \
        _Pragma("unroll") for (int ks = 0; ks < KSH; ++ks) breg[ks] = bsrc[ks * 64];               \
    }
#define MVS_LOAD_A(C)                                                                              \
    {                                                                                              \
        const float* plane = x + (size_t)(C) * V * 8;                                              \
        _Pragma("unroll") for (int i = 0; i < PIECES_PER_THREAD8; ++i)                             \
            stg[i] = *reinterpret_cast<const f32x4*>(plane + goff[i]);                             \
    }
#define MVS_STORE_A()                                                                              \
    {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < PIECES_PER_THREAD8; ++i)                             \
            if (loff[i] >= 0)                                                                      \
                *reinterpret_cast<f32x4*>(tile + loff[i]) =                                        \
                    ((inside >> i) & 1u) ? stg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};                   \
    }

    MVS_LOAD_B(0)
    MVS_LOAD_A(0)
    MVS_STORE_A()
    __syncthreads();

    const float* abase = tile + lane_off;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (c < 3) MVS_LOAD_A(c + 1)
        if (kh == 0) {
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) {
                const int tap0 = 2 * ks;
                const int kz = tap0 / 12, ky = (tap0 / 4) % 3, kx0 = tap0 % 4;
                const int koff = ((kz * HY + ky) * HX + kx0) * VS;
                f32x4 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(abase + koff + i * HX * VS);
                const f32x4 bq = breg[ks];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[i], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) {
                const int tap0 = 2 * (ks + KSH);
                const int kz = tap0 / 12, ky = (tap0 / 4) % 3, kx0 = tap0 % 4;
                const int koff = ((kz * HY + ky) * HX + kx0) * VS;
                f32x4 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(abase + koff + i * HX * VS);
                const f32x4 bq = breg[ks];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[i], 0, 0, 0);
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

    // cross-group reduction through LDS (the halo tile is dead now): group 1 publishes its
    // partial accumulators, group 0 adds them and runs the epilogue
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(tile);
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(mg * 4 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 o = red[(mg * 4 + i) * 64 + lane];
        acc[i][0] += o[0]; acc[i][1] += o[1]; acc[i][2] += o[2]; acc[i][3] += o[3];
    }

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
            if (gz < D && gy < H && gx < W) {
                const float v = fmaxf(acc[i][e] + bv, 0.0f);
                y[(((size_t)gz * H + gy) * W + gx) * 8 + co] = v;
            }
        }
    }
}


// Host-side packing of the Toeplitz-expanded conv0 weights in per-lane fragment order.
//   wfold [27][32][8]  BN-folded tap-major weights  ->  bp [4][18][64][4]
void pack_conv0_pair_weights(const float* wfold, float* bp) {
    using namespace c0;
    for (int c = 0; c < 4; ++c)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j4 = 0; j4 < 4; ++j4) {
                    const int g = lane >> 4, n = lane & 15;
                    const int tap = 2 * ks + (g >> 1);
                    const int kz = tap / 12, ky = (tap / 4) % 3, kxp = tap % 4;
                    const int ci = 8 * c + 4 * (g & 1) + j4;
                    const int jj = n >> 3, co = n & 7;
                    const int kx = kxp - jj;
                    float v = 0.0f;
                    if (kx >= 0 && kx <= 2) v = wfold[((size_t)(kz * 9 + ky * 3 + kx) * 32 + ci) * 8 + co];
                    bp[(((size_t)c * KS + ks) * 64 + lane) * 4 + j4] = v;
                }
}

