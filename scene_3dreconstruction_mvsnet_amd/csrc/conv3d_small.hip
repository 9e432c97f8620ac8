// conv3d_small.hip -- fp32-MFMA 3x3x3 convolutions for the SMALL levels of CostRegNet (gfx950).
//
// conv5 32->64 s2, conv6 64->64 and conv7 (transposed) 64->32 run on the 1/8-resolution grid (cfg2: 24 x 16 x 20
// = 7,680 voxels = 480 MFMA M-tiles for 256 CUs; models/mvsnet.py:44-49,66-69; block ConvBnReLU3D,
// models/module.py:26-33).  With the chunk-pipelined kernels of
// conv3d_mfma.hip these layers are one or two rounds of 4-wave blocks whose life is a serial chain -- per chunk of
// 8 input channels: global halo load -> barrier -> LDS store -> barrier -> 56..224 MFMAs -> the next chunk's 14 KB
// B panel from L2 -- at one or two waves per SIMD, so nothing hides the chain: 20-33 us per layer against a
// 5-11 us MFMA floor (16-bit builds of the same layers, where the MFMAs are free, still take 10-13 us).
// Measured at cfg2: conv5 0.0200 -> 0.0175 ms, conv6 0.0329 -> 0.0236, conv7 0.0277 -> 0.0225.
//
// Here the chain is cut across, not pipelined:
//   * ALL input channels of the block's halo tile are staged in LDS at once (one barrier per block);
//   * K is split over the waves: a block has NT x KW waves (up to 16 = 4 per SIMD), wave (nt, kq) owns N-tile nt
//     and the chunks kq*CPW .. +CPW-1; every wave streams only its own slice of the weight panel, all slices are in
//     flight at once, and each B register feeds MT M-tiles;
//   * the KW partial accumulators are summed through LDS in a fixed order (bit-reproducible), + bias, ReLU.
//   * M-tile = 4(y) x 4(x) voxels (the 2 x 8 tile of conv3d_mfma.hip wastes 1/6 of the MFMAs on a 20-voxel row);
//     conflict-free ds_read_b128 A fragments need the row pitch = 4 (mod 8) voxels: the read's lane groups
//     {0-3,12-15,20-27} / {4-11,16-19,28-31} then cover 16 distinct 16-byte slots (voxel stride 32 B for stride 1,
//     48 B for stride 2).
// Same B panels as conv3d_mfma.hip (pack_convg_weights / pack_deconvg_weights), same k order inside a chunk;
// only the order in which the chunks' partial sums are added differs (fp32 re-association).
#include "mvs_internal.h"
#include "storage.h"

namespace mvs {

namespace {

constexpr int pitch4mod8(int hx) { return hx + ((4 - hx % 8) + 8) % 8; }

template <int CIN, int COUT, int S, int BZ, int BY, int BX, int KW>
struct ConvS {
    static constexpr int NT = COUT / 16;
    static constexpr int NCH = CIN / 8;
    static constexpr int CPW = NCH / KW;              // chunks per wave
    static constexpr int MT = BZ * BY * BX;           // M-tiles (4 x 4 output voxels) per block, all on every wave
    static constexpr int WAVES = NT * KW;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int KS = 14;                     // 27 taps padded to 28 = 14 k-steps of 2 taps x 8 channels
    static constexpr int VS = (S == 1) ? 8 : 12;      // floats per voxel in LDS
    static constexpr int HZ = (BZ - 1) * S + 3, HY = (4 * BY - 1) * S + 3, HX = (4 * BX - 1) * S + 3;
    static constexpr int HXP = pitch4mod8(HX);
    static constexpr int CH_FLOATS = HZ * HY * HXP * VS;
    static constexpr int IN_FLOATS = NCH * CH_FLOATS;
    static constexpr int RED_FLOATS = (KW - 1) * NT * MT * 256;
    static constexpr int LDS_FLOATS = IN_FLOATS + RED_FLOATS;
    static constexpr int NPIECE = NCH * HZ * HY * HX * 2;   // 16-byte pieces of the whole halo tile
    static constexpr int PPT = (NPIECE + THREADS - 1) / THREADS;
    static_assert(NCH % KW == 0, "chunks must split evenly over the K groups");
    static_assert(THREADS <= 1024, "at most 16 waves per block");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "tile exceeds the CU's LDS");
    static_assert(PPT <= 32, "piece mask is 32 bits");
    static constexpr int tap_off(int tap) {  // LDS float offset of tap (kz,ky,kx); tap 27 = padding (zero weights)
        const int t = tap > 26 ? 26 : tap;
        return (((t / 9) * HY + (t / 3) % 3) * HXP + t % 3) * VS;
    }
};

template <int DT, int CIN, int COUT, int S, int BZ, int BY, int BX, int KW>
__global__ __launch_bounds__(64 * (COUT / 16) * KW) void convs_mfma_kernel(
    const void* __restrict__ x,      // [CIN/8][Di][Hi][Wi][8] storage dtype DT
    const float* __restrict__ bp,    // [NCH][NT][14][64][4]   (pack_convg_weights)
    const float* __restrict__ bias,  // [COUT]
    void* __restrict__ y,            // [COUT/8][Do][Ho][Wo][8] storage dtype DT
    int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
    using G = ConvS<CIN, COUT, S, BZ, BY, BX, KW>;
    __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS];
    float* const tile = lds;
    float* const red = lds + G::IN_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NT, kq = wave / G::NT;
    const int nbx = (Wo + 4 * BX - 1) / (4 * BX), nby = (Ho + 4 * BY - 1) / (4 * BY);
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ox0 = bx * 4 * BX, oy0 = by * 4 * BY, oz0 = bz * BZ;      // first output voxel of the block
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;  // halo origin (input coordinates)
    const size_t Vin = (size_t)Di * Hi * Wi, Vout = (size_t)Do * Ho * Wo;

    // this wave's first B panel: in flight while the tile is staged
    f32x4 breg[G::KS];
    auto load_b = [&](int c) {
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(bp) + ((size_t)(c * G::NT + nt) * G::KS) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) breg[ks] = bsrc[ks * 64];
    };
    load_b(kq * G::CPW);

    // ---- stage the halo tile of ALL chunks (zero outside the volume) ----
    {
        f32x4 stg[G::PPT];
        int loff[G::PPT];
        unsigned inside = 0;
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) {
            const int p = tid + i * G::THREADS;
            const int half = p & 1;
            int v = p >> 1;
            const int hx = v % G::HX; v /= G::HX;
            const int hy = v % G::HY; v /= G::HY;
            const int hz = v % G::HZ;
            const int c = v / G::HZ;
            const int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
            const bool have = p < G::NPIECE;
            const bool ok = have && gz >= 0 && gz < Di && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
            const size_t goff = ok ? ((size_t)c * Vin + ((size_t)gz * Hi + gy) * Wi + gx) * 8 + half * 4 : 0;
            stg[i] = St<DT>::load4(x, goff);
            inside |= ok ? (1u << i) : 0u;
            loff[i] = have ? c * G::CH_FLOATS + ((hz * G::HY + hy) * G::HXP + hx) * G::VS + half * 4 : -1;
        }
#pragma unroll
        for (int i = 0; i < G::PPT; ++i)
            if (loff[i] >= 0)
                *reinterpret_cast<f32x4*>(tile + loff[i]) = ((inside >> i) & 1u) ? stg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    // A fragment: lane (r = lane&15 -> voxel (ry, rx) of the 4 x 4 M-tile, g = lane>>4): k-step ks covers taps 2ks
    // (g>>1 == 0) and 2ks+1 (g>>1 == 1), channels 4(g&1)..+3 of the chunk
    const int r = lane & 15, g = lane >> 4, gh = g >> 1;
    const int ry = r >> 2, rx = r & 3;
    int abase[G::MT];
#pragma unroll
    for (int i = 0; i < G::MT; ++i) {
        const int tx = i % BX, ty = (i / BX) % BY, tz = i / (BX * BY);
        abase[i] = (((tz * S) * G::HY + (4 * ty + ry) * S) * G::HXP + (4 * tx + rx) * S) * G::VS + (g & 1) * 4;
    }
    f32x4 acc[G::MT];
#pragma unroll
    for (int i = 0; i < G::MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int cc = 0; cc < G::CPW; ++cc) {
        const int c = kq * G::CPW + cc;
        if (cc > 0) load_b(c);
        const float* tc = tile + c * G::CH_FLOATS;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const int koff = gh ? G::tap_off(2 * ks + 1) : G::tap_off(2 * ks);
            f32x4 a[G::MT];
#pragma unroll
            for (int i = 0; i < G::MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(tc + abase[i] + koff);
            const f32x4 bq = breg[ks];
#pragma unroll
            for (int i = 0; i < G::MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[i], 0, 0, 0);
        }
    }

    // ---- sum the KW partial results (fixed order kq = 0, 1, 2, ...), + bias, ReLU, store ----
    if constexpr (KW > 1) {
        if (kq > 0) {
#pragma unroll
            for (int i = 0; i < G::MT; ++i)
                *reinterpret_cast<f32x4*>(red + ((((kq - 1) * G::NT + nt) * G::MT + i) * 64 + lane) * 4) = acc[i];
        }
        __syncthreads();
        if (kq > 0) return;
#pragma unroll
        for (int q = 1; q < KW; ++q)
#pragma unroll
            for (int i = 0; i < G::MT; ++i)
                acc[i] += *reinterpret_cast<const f32x4*>(red + ((((q - 1) * G::NT + nt) * G::MT + i) * 64 + lane) * 4);
    }
    // D layout: col n = lane&15 -> co = 16 nt + n; row m = 4*(lane>>4) + e -> voxel (ry = lane>>4, rx = e) of the tile
    const int n = lane & 15, co = 16 * nt + n;
    const float bv = bias[co];
    const size_t yplane = (size_t)(co >> 3) * Vout * 8 + (co & 7);
#pragma unroll
    for (int i = 0; i < G::MT; ++i) {
        const int tx = i % BX, ty = (i / BX) % BY, tz = i / (BX * BY);
        const int gz = oz0 + tz, gy = oy0 + 4 * ty + (lane >> 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gx = ox0 + 4 * tx + e;
            if (gz < Do && gy < Ho && gx < Wo)
                St<DT>::store1(y, yplane + (((size_t)gz * Ho + gy) * Wo + gx) * 8, fmaxf(acc[i][e] + bv, 0.0f));
        }
    }
}

template <int DT, int CIN, int COUT, int S, int BZ, int BY, int BX, int KW>
int run_convs(const void* x, void* y, const float* bp, const float* bias, int Di, int Hi, int Wi, hipStream_t s) {
    using G = ConvS<CIN, COUT, S, BZ, BY, BX, KW>;
    const int Do = (Di - 1) / S + 1, Ho = (Hi - 1) / S + 1, Wo = (Wi - 1) / S + 1;
    const int nb = ((Wo + 4 * BX - 1) / (4 * BX)) * ((Ho + 4 * BY - 1) / (4 * BY)) * ((Do + BZ - 1) / BZ);
    convs_mfma_kernel<DT, CIN, COUT, S, BZ, BY, BX, KW><<<nb, G::THREADS, 0, s>>>(x, bp, bias, y, Di, Hi, Wi, Do, Ho, Wo);
    return check_hip(hipGetLastError(), "convs_mfma launch");
}

template <int DT>
int launch_convs_dt(int layer, const void* x, void* y, const float* bp, const float* bias, int Di, int Hi, int Wi,
                    hipStream_t s) {
    switch (layer) {
        // block tile 1 x 8 x 4 output voxels (2 M-tiles), 16 waves = 4 N-tiles x 4 K groups
        case 5: return run_convs<DT, 32, 64, 2, 1, 2, 1, 4>(x, y, bp, bias, Di, Hi, Wi, s);
        case 6: return run_convs<DT, 64, 64, 1, 1, 2, 1, 4>(x, y, bp, bias, Di, Hi, Wi, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "convs_mfma: layer %d not covered", layer);
    }
}


// ---------------------------------------------------------------------------------------------
// Transposed convolution (ConvTranspose3d k3 s2 p1 op1 + folded BN + ReLU, + skip: models/mvsnet.py:47-60,69-70)
// in the same scheme, gather form over the INPUT grid exactly as deconvg_mfma_kernel (conv3d_mfma.hip): 4 (pz, py)
// output-parity classes = 4 accumulator sets, the x parity folded into N = (px, co), 9 k-steps per chunk
// (deconv_step, mvs_internal.h), same panel (pack_deconvg_weights).  K split over KW wave groups; the exchange
// doubles as the epilogue's work split: class c is finished by group c % KW, which receives the other groups'
// partial sums of that class through LDS, adds them in group order, applies bias + ReLU, adds the skip value and
// stores -- every wave takes part, straight from registers (4-byte accesses in 32-byte runs; these outputs are a
// few MB).
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int BZ, int BY, int BX, int KW>
struct DeconvS {
    static constexpr int NTT = 2 * COUT / 16;         // N-tiles over (px, co)
    static constexpr int NCH = CIN / 8;
    static constexpr int CPW = NCH / KW;
    static constexpr int MT = BZ * BY * BX;           // M-tiles of 4 x 4 INPUT voxels per block
    static constexpr int WAVES = NTT * KW;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int KS = 9;
    static constexpr int VS = 8;
    static constexpr int HZ = BZ + 1, HY = 4 * BY + 1, HX = 4 * BX + 1;
    static constexpr int HXP = pitch4mod8(HX);
    static constexpr int CH_FLOATS = HZ * HY * HXP * VS;
    static constexpr int IN_FLOATS = NCH * CH_FLOATS;
    static constexpr int RED_FLOATS = 4 * (KW - 1) * NTT * MT * 256;   // [class][writer slot][nt][i][lane][4]
    static constexpr int LDS_FLOATS = IN_FLOATS + RED_FLOATS;
    static constexpr int NPIECE = NCH * HZ * HY * HX * 2;
    static constexpr int PPT = (NPIECE + THREADS - 1) / THREADS;
    static_assert(KW == 1 || KW == 2 || KW == 4, "the 4 parity classes are dealt over the K groups");
    static_assert(NCH % KW == 0, "chunks must split evenly over the K groups");
    static_assert(THREADS <= 1024, "at most 16 waves per block");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "tile exceeds the CU's LDS");
    static_assert(PPT <= 32, "piece mask is 32 bits");
};

template <int DT, int CIN, int COUT, int BZ, int BY, int BX, int KW>
__global__ __launch_bounds__(64 * (2 * COUT / 16) * KW) void deconvs_mfma_kernel(
    const void* __restrict__ x,      // [CIN/8][Di][Hi][Wi][8] storage dtype DT
    const float* __restrict__ bp,    // [NCH][NTT][9][64][4]   (pack_deconvg_weights)
    const float* __restrict__ bias,  // [COUT]
    const void* __restrict__ skip,   // [COUT/8][2Di][2Hi][2Wi][8]
    void* __restrict__ y,            // [COUT/8][2Di][2Hi][2Wi][8]
    int Di, int Hi, int Wi) {
    using G = DeconvS<CIN, COUT, BZ, BY, BX, KW>;
    __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS];
    float* const tile = lds;
    float* const red = lds + G::IN_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave % G::NTT, kq = wave / G::NTT;
    const int nbx = (Wi + 4 * BX - 1) / (4 * BX), nby = (Hi + 4 * BY - 1) / (4 * BY);
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ix0 = bx * 4 * BX, iy0 = by * 4 * BY, iz0 = bz * BZ;
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t Vin = (size_t)Di * Hi * Wi, Vout = Vin * 8;

    f32x4 breg[G::KS];
    auto load_b = [&](int c) {
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(bp) + ((size_t)(c * G::NTT + nt) * G::KS) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) breg[ks] = bsrc[ks * 64];
    };
    load_b(kq * G::CPW);

    {
        f32x4 stg[G::PPT];
        int loff[G::PPT];
        unsigned inside = 0;
#pragma unroll
        for (int i = 0; i < G::PPT; ++i) {
            const int p = tid + i * G::THREADS;
            const int half = p & 1;
            int v = p >> 1;
            const int hx = v % G::HX; v /= G::HX;
            const int hy = v % G::HY; v /= G::HY;
            const int hz = v % G::HZ;
            const int c = v / G::HZ;
            const int gz = iz0 + hz, gy = iy0 + hy, gx = ix0 + hx;
            const bool have = p < G::NPIECE;
            const bool ok = have && gz < Di && gy < Hi && gx < Wi;
            const size_t goff = ok ? ((size_t)c * Vin + ((size_t)gz * Hi + gy) * Wi + gx) * 8 + half * 4 : 0;
            stg[i] = St<DT>::load4(x, goff);
            inside |= ok ? (1u << i) : 0u;
            loff[i] = have ? c * G::CH_FLOATS + ((hz * G::HY + hy) * G::HXP + hx) * G::VS + half * 4 : -1;
        }
#pragma unroll
        for (int i = 0; i < G::PPT; ++i)
            if (loff[i] >= 0)
                *reinterpret_cast<f32x4*>(tile + loff[i]) = ((inside >> i) & 1u) ? stg[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    // lane (r -> input voxel (ry, rx) of the 4 x 4 M-tile, g): dx = g>>1, channels 4(g&1)..+3
    const int r = lane & 15, g = lane >> 4;
    const int ry = r >> 2, rx = r & 3;
    int abase[G::MT];
#pragma unroll
    for (int i = 0; i < G::MT; ++i) {
        const int tx = i % BX, ty = (i / BX) % BY, tz = i / (BX * BY);
        abase[i] = ((tz * G::HY + 4 * ty + ry) * G::HXP + 4 * tx + rx + (g >> 1)) * G::VS + (g & 1) * 4;
    }
    f32x4 acc[4][G::MT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < G::MT; ++i) acc[c][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int cc = 0; cc < G::CPW; ++cc) {
        const int c = kq * G::CPW + cc;
        if (cc > 0) load_b(c);
        const float* tc = tile + c * G::CH_FLOATS;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const DeconvStep st = deconv_step(ks);
            const int koff = (st.dz * G::HY + st.dy) * G::HXP * G::VS;
            f32x4 a[G::MT];
#pragma unroll
            for (int i = 0; i < G::MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(tc + abase[i] + koff);
            const f32x4 bq = breg[ks];
#pragma unroll
            for (int i = 0; i < G::MT; ++i) acc[st.cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[st.cls][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MT; ++i) acc[st.cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[st.cls][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MT; ++i) acc[st.cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[st.cls][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < G::MT; ++i) acc[st.cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[st.cls][i], 0, 0, 0);
        }
    }

    // ---- exchange: class c goes to group c % KW (writer slot = the writer's rank among the other groups) ----
    auto red_at = [&](int c, int slot, int i) {
        return reinterpret_cast<f32x4*>(red + ((((c * (KW - 1) + slot) * G::NTT + nt) * G::MT + i) * 64 + lane) * 4);
    };
    if constexpr (KW > 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int owner = c % KW;
            if (owner != kq) {
                const int slot = kq > owner ? kq - 1 : kq;
#pragma unroll
                for (int i = 0; i < G::MT; ++i) *red_at(c, slot, i) = acc[c][i];
            }
        }
        __syncthreads();
    }
    // col nn = 16 nt + (lane&15) -> (px, co); row m = 4*(lane>>4) + e -> input voxel (ry = lane>>4, rx = e)
    const int nn = 16 * nt + (lane & 15);
    const int px = nn / COUT, co = nn % COUT;
    const float bv = bias[co];
    const size_t yplane = (size_t)(co >> 3) * Vout * 8 + (co & 7);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c % KW != kq) continue;
        const int pz = c >> 1, py = c & 1;
        f32x4 sum[G::MT];
#pragma unroll
        for (int i = 0; i < G::MT; ++i) {
            // fixed order over the K groups q = 0 .. KW-1 (this group's own partial at position kq)
            f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KW; ++q) {
                if (q == kq) t = (q == 0) ? acc[c][i] : t + acc[c][i];
                else {
                    const f32x4 o = *red_at(c, q > kq ? q - 1 : q, i);
                    t = (q == 0) ? o : t + o;
                }
            }
            sum[i] = t;
        }
#pragma unroll
        for (int i = 0; i < G::MT; ++i) {
            const int tx = i % BX, ty = (i / BX) % BY, tz = i / (BX * BY);
            const int gz = 2 * (iz0 + tz) + pz, gy = 2 * (iy0 + 4 * ty + (lane >> 4)) + py;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gx = 2 * (ix0 + 4 * tx + e) + px;
                if (gz < Do && gy < Ho && gx < Wo) {
                    const size_t o = yplane + (((size_t)gz * Ho + gy) * Wo + gx) * 8;
                    St<DT>::store1(y, o, fmaxf(sum[i][e] + bv, 0.0f) + St<DT>::load1(skip, o));
                }
            }
        }
    }
}

template <int DT, int CIN, int COUT, int BZ, int BY, int BX, int KW>
int run_deconvs(const void* x, const void* skip, void* y, const float* bp, const float* bias, int Di, int Hi, int Wi,
                hipStream_t s) {
    using G = DeconvS<CIN, COUT, BZ, BY, BX, KW>;
    const int nb = ((Wi + 4 * BX - 1) / (4 * BX)) * ((Hi + 4 * BY - 1) / (4 * BY)) * ((Di + BZ - 1) / BZ);
    deconvs_mfma_kernel<DT, CIN, COUT, BZ, BY, BX, KW><<<nb, G::THREADS, 0, s>>>(x, bp, bias, skip, y, Di, Hi, Wi);
    return check_hip(hipGetLastError(), "deconvs_mfma launch");
}

template <int DT>
int launch_deconvs_dt(int layer, const void* x, const void* skip, void* y, const float* bp, const float* bias, int Di,
                      int Hi, int Wi, hipStream_t s) {
    switch (layer) {
        // block tile 1 x 8 x 4 input voxels (2 M-tiles), 16 waves = 4 N-tiles x 4 K groups
        case 7: return run_deconvs<DT, 64, 32, 1, 2, 1, 4>(x, skip, y, bp, bias, Di, Hi, Wi, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "deconvs_mfma: layer %d not covered", layer);
    }
}

}  // namespace

// conv5, conv6 (launch_convs_mfma) and conv7 (launch_deconvs_mfma).  The 1/4-resolution layers measured SLOWER in
// this scheme at cfg2 (3,840 M-tiles = 3.75 rounds of blocks; conv3 0.034 vs 0.031 ms, conv4 direct 0.045 vs 0.039
// Winograd, conv9 0.041 vs 0.035): with several rounds per CU the chunk pipeline of co-resident blocks overlaps
// well enough, and the wider blocks only lose occupancy.
bool convs_covers(int layer) { return layer >= 5 && layer <= 7; }

int launch_convs_mfma(int layer, const void* x, void* y, const float* bp, const float* bias, int Di, int Hi, int Wi,
                      int dtype, hipStream_t s) {
    if ((size_t)Di * Hi * Wi * 8 >= ((size_t)1 << 31)) return fail(MVS_ERR_BAD_SHAPE, "convs_mfma: plane exceeds 31-bit offsets");
    MVS_DISPATCH_DTYPE(dtype, (launch_convs_dt<DT>(layer, x, y, bp, bias, Di, Hi, Wi, s)))
}

int launch_deconvs_mfma(int layer, const void* x, const void* skip, void* y, const float* bp, const float* bias, int Di,
                        int Hi, int Wi, int dtype, hipStream_t s) {
    MVS_DISPATCH_DTYPE(dtype, (launch_deconvs_dt<DT>(layer, x, skip, y, bp, bias, Di, Hi, Wi, s)))
}

}  // namespace mvs
