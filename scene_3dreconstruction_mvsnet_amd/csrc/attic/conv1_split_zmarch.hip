// attic/conv1_split_zmarch.hip -- NOT BUILT.  conv1 (8 -> 16, stride 2) for fp32 volumes with split bf16 operands as a
// z-marching kernel (round 4).  A fragment of csrc/conv3d_mfma16.hip (it uses that file's gs_split8 / gs_mfma / u32x4 and
// the three-piece panels of pack_split_panels); it was selected by MVS_SPLIT_LAYERS=2 + the rule of conv1z_mfma.
// Correct: tests/layer_check.py at 24x24x40 and 16x16x32 and the full-size per-layer test passed with the fp32 bounds.
// Slower: 0.054 ms at cfg2 against 0.046 for the fp32-MFMA z-marching kernel (conv1z_mfma) -- three rings in LDS (147 KB,
// one block per CU), twelve 16-byte LDS writes per staged plane pair, 21 A-fragment reads per step: the step got longer
// than the 56 fp32 MFMAs it replaced with 42 bf16 ones.
//
// ---------------------------------------------------------------------------------------------
// conv1zs: conv1 (8 -> 16, stride 2, models/mvsnet.py:38) for fp32 volumes with split operands as a z-MARCHING kernel --
// convz16's scheme (a 512-thread block owns an 8 x 16 column of the OUTPUT, a ring of five input planes in LDS, two
// new planes per output plane requested two steps ahead into rotating register sets, the panel in registers, one block
// barrier per output plane, branch-free VMEM) with fp32 planes in, three bf16 rings (a staged voxel is split once on its
// way into LDS), three weight panels, six MFMAs per (k-step, M-tile), and fp32 results through a wave-private strip
// into 16-byte stores.  conv1 is bound by the bytes it moves (157 MB: 20 us) as soon as its matrix work is cheap: the
// fp32-MFMA z-marching kernel (conv1z_mfma, 0.046 ms) spends 56 fp32 MFMAs per wave and step, this one 42 bf16 ones.
// MEASURED SLOWER (0.054 ms at cfg2): selected only by MVS_SPLIT_LAYERS=2 (tests); conv1 stays on conv1z_mfma.
// ---------------------------------------------------------------------------------------------
namespace c1zs {
constexpr int S = 2, TY = 8, TX = 16, HY = (TY - 1) * S + 3, HX = (TX - 1) * S + 3;   // 17 x 33 input voxels
constexpr int HXP = 34;                               // row pitch (voxels)
constexpr int PLANE = HY * HXP * 8;                   // bf16 elements of one piece of one plane
constexpr int RING = 5;
constexpr int PIECE = RING * PLANE;                   // one piece's ring
constexpr int THREADS = 512;
constexpr int NVOX = HY * HX;                         // 561 staged voxels per plane
constexpr int PPT = (NVOX + THREADS - 1) / THREADS;   // 2
constexpr int STRIP = 16 * 16;                        // floats: [voxel 16][channel 16] of the wave's M-tile
static_assert(3 * PIECE * 2 + 8 * STRIP * 4 <= 160 * 1024, "rings exceed the CU's LDS");
}  // namespace c1zs

__global__ __launch_bounds__(c1zs::THREADS) void conv1zs_mfma_kernel(
    const float* __restrict__ x,            // [1][Di][Hi][Wi][8] fp32
    const unsigned short* __restrict__ bp,  // [3 pieces][1][1][7][64][8] bf16 (pack_split_panels)
    const float* __restrict__ bias,         // [16]
    float* __restrict__ y,                  // [2][Do][Ho][Wo][8] fp32
    int Di, int Hi, int Wi, int Do, int Ho, int Wo, int ZC, int nbx, int nby) {
    using namespace c1zs;
    __shared__ __attribute__((aligned(16))) unsigned short ring[3 * PIECE];
    __shared__ __attribute__((aligned(16))) float strips[8 * STRIP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b;
    {   // XCD k works through the k-th eighth of the (z chunk, row, column) sequence
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ox0 = bx * TX, oy0 = by * TY;
    const int za = bz * ZC, zb = min(za + ZC, Do);
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1;
    const size_t HW8 = (size_t)Hi * Wi * 8;

    // staging: voxel v = tid + i * THREADS of the 17 x 33 halo (threads beyond it shadow the last voxel)
    unsigned goff[PPT];   // byte offset inside a plane; beyond the descriptor when outside the image (zero fill)
    int loff[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int v = min(tid + i * THREADS, NVOX - 1);
        const int hx = v % HX, hy = v / HX;
        const int gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
        goff[i] = ok ? (unsigned)(((size_t)gy * Wi + gx) * 32) : 0x80000000u;
        loff[i] = (hy * HXP + hx) * 8;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), (short)0,
                                                                          (int)((size_t)Di * HW8 * 4), 0x00020000);
    // raw buffer loads; the plane in the scalar offset (clamped into the volume: a plane outside it is zeroed when stored)
    auto load_plane = [&](int gz, f32x4 (&st)[PPT][2]) {
        const unsigned zo = (unsigned)((size_t)min(max(gz, 0), Di - 1) * HW8 * 4);
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            st[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)goff[i], (int)zo, 0));
            st[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(goff[i] + 16u), (int)zo, 0));
        }
    };
    auto slot_of = [&](int gz) { return ((gz + 1) % RING) * PLANE; };   // gz >= -1
    auto store_plane = [&](int gz, const f32x4 (&st)[PPT][2]) {
        unsigned short* slot = ring + slot_of(gz);
        const bool zok = gz >= 0 && gz < Di;
        const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            u32x4 p1, p2, p3;
            gs_split8(zok ? st[i][0] : z4, zok ? st[i][1] : z4, p1, p2, p3);
            *reinterpret_cast<u32x4*>(slot + loff[i]) = p1;
            *reinterpret_cast<u32x4*>(slot + PIECE + loff[i]) = p2;
            *reinterpret_cast<u32x4*>(slot + 2 * PIECE + loff[i]) = p3;
        }
    };

    // the three panels in registers: breg[piece][ks]
    u32x4 breg[3][7];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) breg[q][ks] = reinterpret_cast<const u32x4*>(bp)[(q * 7 + ks) * 64 + lane];

    // A fragment: lane (r -> voxel (ry, rx) of the wave's M-tile, g -> tap 4 ks + g); wave -> M-tile (ty, tx)
    const int r = lane & 15, g = lane >> 4;
    const int mtx = wave % (TX / 8), mty = wave / (TX / 8);
    const int mt_off = ((S * (2 * mty + (r >> 3))) * HXP + S * (8 * mtx + (r & 7))) * 8;
    int kin[7], kzl[7];
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        const int tap = min(4 * ks + g, 26);   // tap 27 is padding (zero weights): any valid address
        kzl[ks] = tap / 9;
        kin[ks] = (((tap / 3) % 3) * HXP + tap % 3) * 8;
    }
    const int n = lane & 15;
    const float bv = bias[n];
    float* strip = strips + wave * STRIP;
    // epilogue store role: lane -> (voxel sv of the M-tile, 4-channel quarter sq): 16 voxels x 4 quarters
    const int sv = lane >> 2, sq = lane & 3;
    const int sgy = oy0 + 2 * mty + (sv >> 3), sgx = ox0 + 8 * mtx + (sv & 7);
    const bool st_ok = sgy < Ho && sgx < Wo;
    const size_t Vout8 = (size_t)Do * Ho * Wo * 8;
    const unsigned st_off = (unsigned)(((size_t)(sq >> 1) * Vout8 + ((size_t)sgy * Wo + sgx) * 8 + (sq & 1) * 4) * 4);
    const unsigned zstep_b = (unsigned)((size_t)Ho * Wo * 8 * 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, (int)(unsigned)(2 * Vout8 * 4), 0x00020000);

    // prologue: input planes 2 za - 1 .. 2 za + 1 into the rings; the planes of steps za and za + 1 in flight
    f32x4 w[3][S][PPT][2];   // [register set][plane of the step][voxel][half]
    {
        f32x4 t0[PPT][2], t1[PPT][2], t2[PPT][2];
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

    auto step = [&](int zo, f32x4 (&cur)[S][PPT][2], f32x4 (&nx)[S][PPT][2]) {
#pragma unroll
        for (int q = 0; q < S; ++q) load_plane(S * (zo + 2) + 2 + q, nx[q]);
        int sb[3];
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) sb[kz] = slot_of(S * zo - 1 + kz);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const unsigned short* ap = ring + (kzl[ks] == 0 ? sb[0] : kzl[ks] == 1 ? sb[1] : sb[2]) + kin[ks] + mt_off;
            const u32x4 a1 = *reinterpret_cast<const u32x4*>(ap);
            const u32x4 a2 = *reinterpret_cast<const u32x4*>(ap + PIECE);
            const u32x4 a3 = *reinterpret_cast<const u32x4*>(ap + 2 * PIECE);
            acc = gs_mfma(a3, breg[0][ks], acc);   // the small terms first
            acc = gs_mfma(a1, breg[2][ks], acc);
            acc = gs_mfma(a2, breg[1][ks], acc);
            acc = gs_mfma(a2, breg[0][ks], acc);
            acc = gs_mfma(a1, breg[1][ks], acc);
            acc = gs_mfma(a1, breg[0][ks], acc);
        }
        // epilogue: element e of acc = voxel m = 4 g + e of the M-tile, channel n -> strip[m][n]
#pragma unroll
        for (int e = 0; e < 4; ++e) strip[(4 * g + e) * 16 + n] = fmaxf(acc[e] + bv, 0.0f);
        __builtin_amdgcn_wave_barrier();
        const f32x4 o = *reinterpret_cast<const f32x4*>(strip + sv * 16 + sq * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrs,
                                               (int)((st_ok && zo < zb) ? st_off + (unsigned)zo * zstep_b : 0xFFFFFFF0u), 0, 0);
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

// conv1 through the split z-marching kernel when its columns fill the chip (the rule of conv1z_mfma); else *taken = false
static int try_conv1zs(const void* x, void* y, const unsigned short* bp, const float* bias, int Di, int Hi, int Wi,
                       hipStream_t s, bool* taken) {
    using namespace c1zs;
    *taken = false;
    const int Do = (Di - 1) / 2 + 1, Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    static const int zm = [] {   // MVS_CONV1Z=0/1 = never / always a z-marching conv1 (shared with conv1z_mfma)
        const char* e = getenv("MVS_CONV1Z");
        return e ? atoi(e) : -1;
    }();
    const int nbx = (Wo + TX - 1) / TX, nby = (Ho + TY - 1) / TY;
    const long ncol = (long)nbx * nby;
    if (zm == 0 || Do < 4 || (size_t)Do * Ho * Wo * 16 * 4 >= ((size_t)1 << 32) - 64 || (size_t)Di * Hi * Wi * 32 >= ((size_t)1 << 31))
        return MVS_OK;
    if (zm != 1 && ncol * (Do / 8) * 4 < (long)cus * 3) return MVS_OK;
    int best = 1;
    double best_eff = 0.0;
    for (int nz = 1; nz <= (Do + 3) / 4; ++nz) {   // z chunks of >= 4 output planes: best fill of the last round
        const int zc = (Do + nz - 1) / nz, nzc = (Do + zc - 1) / zc;
        const long nb = ncol * nzc;
        const double eff = (double)nb / (double)(((nb + cus - 1) / cus) * cus) * zc / (zc + 2.5);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = nz; }
    }
    const int ZC = (Do + best - 1) / best, nzc = (Do + ZC - 1) / ZC;
    *taken = true;
    conv1zs_mfma_kernel<<<(unsigned)(ncol * nzc), THREADS, 0, s>>>(static_cast<const float*>(x), bp, bias,
                                                                   static_cast<float*>(y), Di, Hi, Wi, Do, Ho, Wo, ZC, nbx, nby);
    return check_hip(hipGetLastError(), "conv1zs_mfma launch");
}

