// conv11_prob.hip -- the last two CostRegNet layers in one kernel (gfx950, fp32 storage).
//
// Replaces models/mvsnet.py:71-72
//     x = conv0 + self.conv11(x)        ConvTranspose3d(16, 8, k3 s2 p1 op1) + BN + ReLU, + skip
//     x = self.prob(x)                  Conv3d(8, 1, k3 p1) with bias
// without the full-resolution 8-channel tensor in between ever reaching HBM: as two launches the
// 126 MB (cfg2) are written by conv11 and read back by prob, and the two kernels took 0.077 + 0.061 ms of
// the 0.95 ms depth path.  Algorithmic traffic of the fused form: the 16-channel half-resolution input
// (31 MB), conv0's output as skip (126 MB), the fp32 logits (16 MB).
//
// A block owns a column of the volume -- 8 x 16 input voxels, i.e. a 16 x 32 tile of conv11 outputs, of which
// the inner 14 x 30 (plus the volume's own border rows / columns) become logits -- and marches along z over
// a chunk of input planes.  Per input plane i:
//   1. the two input planes i, i+1 are staged per chunk of 8 channels (requested one phase ahead); the skip
//      values of conv11's planes 2i, 2i+1 are requested one step ahead;
//   2. the transposed convolution runs on the 16x16x4 fp32 MFMA exactly as in deconvg_mfma_kernel
//      (gather form over the input grid, x parity folded into N, 4 (pz, py) classes, 9 k-steps per chunk
//      of 8 input channels; same panel, same accumulation order);
//   3. ReLU(acc + bias) is scattered into the LDS tile, then the skip values are added 16 bytes at a time
//      (LDS float atomics onto pre-stored skip values cost 0.2 ms: measured, dropped);
//   4. the 3x3x3 prob stencil runs in scatter form over z: plane p of conv11 feeds the running sums of
//      the logits p-1, p, p+1 (taps kz = 2, 1, 0), so only the two planes just produced have to be in
//      LDS and each of them is read once instead of three times; logit p-1 is complete after plane p.
// Tiles overlap by one input voxel in y and x and chunks by one input plane in z (the halo is recomputed,
// 1.35 x 1.1 of the MFMA work at cfg2).  The stencil's sums are kept as (even, odd channel) pairs so that
// its FMAs are v_pk_fma_f32 with the weights as SGPR pairs; the summation order differs from prob_lds_kernel
// (7e-7 of the logit range at cfg2, tests/test_gpu_parity.py::test_fused_conv11_prob_matches_the_two_launches).
//
// Measured (profiles/r02_conv11_prob.md): 0.117-0.125 ms against 0.135-0.138 ms for the two launches.  The
// kernel is bound by the SUM of its fp32-MFMA cycles (32 % of its SIMD time) and its vector-instruction
// issue (23 %): on gfx950 the fp32 MFMA and the (packed) fp32 VALU share hardware -- equal peak rates -- and
// do not overlap on a SIMD.  A producer / consumer split of the waves (MFMA waves feeding stencil waves
// through a second tile buffer, one barrier per step) measured the same 0.120 ms and was dropped; so were
// LDS float atomics for the skip add (+0.2 ms).
#include <cstdlib>

#include "mvs_internal.h"
#include "storage.h"
#include "split_ops.h"

// Diagnostic builds (`make ablate21` .. `ablate24`, wrong results, timing only): 21 = no stencil, 22 = no
// scatter of the accumulators, 23 = no MFMAs, 24 = no skip loads; the split-operand kernel: 81 = no stencil, 82 = no
// MFMAs, 83 = no split arithmetic, 84 = no scatter / skip add, 85 / 86 = one B / A fragment read per chunk and piece.  Product builds leave MVS_ABLATE at 0.
#ifndef MVS_ABLATE
#define MVS_ABLATE 0
#endif
// 87: the split-operand kernel without its four block barriers per step, 88: without the two around the odd plane's scatter
#if MVS_ABLATE == 87
#define SPLIT_SYNC_A() __builtin_amdgcn_wave_barrier()
#define SPLIT_SYNC_B() __builtin_amdgcn_wave_barrier()
#elif MVS_ABLATE == 88
#define SPLIT_SYNC_A() __syncthreads()
#define SPLIT_SYNC_B() __builtin_amdgcn_wave_barrier()
#else
#define SPLIT_SYNC_A() __syncthreads()
#define SPLIT_SYNC_B() __syncthreads()
#endif

namespace mvs {

namespace cp {
constexpr int BY = 4, BX = 2;                      // MFMA M-tiles (2 rows x 8 columns of input voxels) per block
constexpr int IY = 2 * BY, IX = 8 * BX;            // input tile: 8 x 16 voxels
constexpr int OY = 2 * IY, OX = 2 * IX;            // conv11 tile: 16 x 32 voxels
constexpr int PY = OY - 2, PX = OX - 2;            // logits per tile: 14 x 30

// conv11 tile in LDS: [plane 2][row OY + 2][half 2][x XP][4 channels], a zero border of one voxel all around;
// x-adjacent lanes read adjacent 16-byte pieces of one half-row: conflict-free ds_read_b128
constexpr int XP = 36;
constexpr int HS = XP * 4, RS = 2 * HS, PS = (OY + 2) * RS;
constexpr int C_FLOATS = 2 * PS;

// A wave's private input window in the 16-bit MFMA tap scheme (conv11_prob16 / conv11_prob_split): 6 rows ([plane 2][row 3])
// of 9 voxels, one 16-byte slot per voxel (8 channels x 16 bit).  An A-fragment ds_read_b128 touches rows rho, rho + 1
// (rho = 0, 1, 3, 4): in one of its 16-lane groups columns {0-3, 5-8} of one row and {1-7} of the other, 15 distinct
// slots.  With rows of 10 slots back to back two of them share banks (8 LDS cycles per read instead of 4: measured as
// SQ_LDS_BANK_CONFLICT = 26 % of the kernel's LDS cycles); conflict-free needs consecutive rows 8 slots apart mod 16,
// so columns 0..7 are packed 8 to a row and the ninth column lives in side slots chosen to fall on the bank quad its
// group leaves free (rows 0, 2, 4: 4 or 8 mod 16; rows 1, 3, 5: 12 or 0 mod 16).
constexpr int WSLOTS = 69;
__host__ __device__ constexpr int win_slot(int rho, int c) {
    return c < 8 ? 8 * rho + c : rho == 0 ? 52 : rho == 1 ? 48 : rho == 2 ? 56 : rho == 3 ? 60 : rho == 4 ? 68 : 64;
}
}  // namespace cp

typedef float f32x2v __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// 512 threads per block; everything before the stencil is private to a wave (the first form, with a shared
// input halo and six block barriers per step, is in attic/conv11_prob_form1.hip).  Wave w owns M-tile w (2 x 8 input voxels -> a 4 x 16 strip of both conv11 planes): it
// stages its own 3 x 9 voxel input halo per chunk (1.4x the loads of the shared halo, L1 hits), runs its
// 72 MFMAs, scatters and adds the skip values of its own strip -- LDS operations of one wave execute in
// order, so none of this needs a barrier.  Two block barriers per step (tile complete / tile consumed)
// instead of six, and the eight waves drift apart inside the producing phase instead of marching in step.
// ---------------------------------------------------------------------------------------------
namespace cpv {
using namespace cp;
constexpr int RPI = 76;                      // floats per row of a private input tile: 9 voxels + 4 (second row lands 4 banks on)
constexpr int WIN = 2 * 3 * RPI;             // one chunk: [plane 2][row 3]
constexpr int NPC = 2 * 27 * 2;              // 16-byte pieces of it: [plane][3 x 9 voxels][half]
}  // namespace cpv

template <int DT>
__global__ __launch_bounds__(512, 4) void conv11_prob_priv_kernel(
    const void* __restrict__ x, const float* __restrict__ bp, const float* __restrict__ bias,
    const void* __restrict__ skip, const float* __restrict__ pw, const float* __restrict__ pbias,
    float* __restrict__ cost, int Di, int Hi, int Wi, int ZC, int nbx, int nby) {
    using namespace cpv;
    __shared__ __attribute__((aligned(16))) float ct[C_FLOATS];
    __shared__ __attribute__((aligned(16))) float win[8 * WIN];
    __shared__ __attribute__((aligned(16))) float bpan[2 * 9 * 64 * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // blocks are dealt round-robin over the 8 XCDs (blockIdx.x % 8 names the XCD: speed only, never correctness) and
    // every XCD has its own L2: XCD k works through the k-th eighth of the (z chunk, row, column) sequence, so that
    // tiles sharing a halo column / row (and the skip planes both read) run on the same L2 at about the same time
    int b;
    {
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;   // XCD k runs q (+1 if k < rem) blocks
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ix0 = bx * (IX - 1), iy0 = by * (IY - 1);
    const int za = bz * ZC, zb = min(za + ZC, Di);
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t Vin = (size_t)Di * Hi * Wi, HWi = (size_t)Hi * Wi, HWo = (size_t)Ho * Wo;
    const int i_first = za > 0 ? za - 1 : 0, i_last = zb < Di ? zb : Di - 1;

    // ---- producer roles (wave-private) ----
    const int tx = wave & 1, ty = wave >> 1;            // M-tile: input rows 2 ty .. +1, columns 8 tx .. +7
    float* my = win + wave * WIN;
    int prel[2], ploff[2];
    unsigned pin = 0, phz = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = lane + 64 * j;
        const int half = p & 1, v = p >> 1, hz = v / 27, vv = v % 27, hr = vv / 9, hc = vv % 9;
        const int gy = iy0 + 2 * ty + hr, gx = ix0 + 8 * tx + hc;
        const bool ok = p < NPC && gy < Hi && gx < Wi;
        prel[j] = ok ? (int)((((size_t)hz * Hi + gy) * Wi + gx) * 8 + half * 4) : 0;
        pin |= ok ? (1u << j) : 0u;
        phz |= (hz & 1) ? (1u << j) : 0u;
        ploff[j] = p < NPC ? (hz * 3 + hr) * RPI + hc * 8 + half * 4 : -1;
    }
    f32x4 pre[2];
    bool pre_z1 = true;
    auto load_a = [&](int i, int c) {    // raw loads; masked when they go to LDS
        pre_z1 = i + 1 < Di;
        const size_t base = ((size_t)c * Vin + (size_t)i * HWi) * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = ((pin >> j) & 1u) && (pre_z1 || !((phz >> j) & 1u));
            pre[j] = St<DT>::load4(x, ok ? base + prel[j] : (size_t)0);
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = ((pin >> j) & 1u) && (pre_z1 || !((phz >> j) & 1u));
            if (ploff[j] >= 0) *reinterpret_cast<f32x4*>(my + ploff[j]) = ok ? pre[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    // MFMA lane roles (deconvg_mfma_kernel): row r -> input voxel (r >> 3, r & 7) of the M-tile, g: dx = g >> 1,
    // channels 4 (g & 1) .. + 3; column n = r -> (px, co)
    const int r = lane & 15, g = lane >> 4;
    const int abase = (r >> 3) * RPI + ((r & 7) + (g >> 1)) * 8 + (g & 1) * 4;
    const int px = r >> 3, co = r & 7;
    const float bv = bias[co];
    // element e of the accumulators: input voxel m = 4 g + e -> tile column 8 tx + (m & 7) = 8 tx + 4 (g & 1) + e
    const int sbase0 = (2 * (2 * ty + (g >> 1)) + 1) * RS + (co >> 2) * HS + (2 * (8 * tx + 4 * (g & 1)) + px + 1) * 4 + (co & 3);
    // skip pieces of the wave's strip: lane -> column xx, half, row parity; j -> plane j >> 1, row 2 (j & 1) + parity
    int srel[4], sl[4];
    unsigned sok = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xx = lane & 15, half = (lane >> 4) & 1, rr = 2 * (j & 1) + (lane >> 5), pz = j >> 1;
        const int oy = 4 * ty + rr, ox = 16 * tx + xx;
        const int gy = 2 * iy0 + oy, gx = 2 * ix0 + ox;
        const bool ok = gy < Ho && gx < Wo;
        srel[j] = ok ? (int)((((size_t)pz * Ho + gy) * Wo + gx) * 8 + half * 4) : 0;
        sok |= ok ? (1u << j) : 0u;
        sl[j] = pz * PS + (oy + 1) * RS + half * HS + (ox + 1) * 4;
    }
    f32x4 sk[4];
    auto load_skip = [&](int i, bool de, bool dodd) {
        const size_t base = (size_t)(2 * i) * HWo * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = ((sok >> j) & 1u) && ((j >> 1) ? dodd : de);
            sk[j] = St<DT>::load4(skip, ok ? base + srel[j] : (size_t)0);   // masked when added
        }
    };
    const f32x4* bsrc = reinterpret_cast<const f32x4*>(bpan) + lane;
    f32x4 acc[4];
    auto fetch = [&](int c, int ks, f32x4& a, f32x4& bq) {
        const DeconvStep st = deconv_step(ks);
        a = *reinterpret_cast<const f32x4*>(my + (st.dz * 3 + st.dy) * RPI + abase);
        bq = bsrc[(c * 9 + ks) * 64];
    };
    auto mfma_step = [&](int ks, const f32x4& a, const f32x4& bq) {
        const int cls = deconv_step(ks).cls;
        acc[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq.x, acc[cls], 0, 0, 0);
        acc[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq.y, acc[cls], 0, 0, 0);
        acc[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq.z, acc[cls], 0, 0, 0);
        acc[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq.w, acc[cls], 0, 0, 0);
    };
    auto mfma_chunk = [&](int c, bool de, bool dodd) {
        if (de && dodd) {   // the usual step: operands of k-step ks + 1 requested before the MFMAs of ks
            f32x4 a[2], bq[2];
            fetch(c, 0, a[0], bq[0]);
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                if (ks + 1 < 9) fetch(c, ks + 1, a[(ks + 1) & 1], bq[(ks + 1) & 1]);
                mfma_step(ks, a[ks & 1], bq[ks & 1]);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                if (deconv_step(ks).cls < 2 ? !de : !dodd) continue;   // wave-uniform
                f32x4 a, bq;
                fetch(c, ks, a, bq);
                mfma_step(ks, a, bq);
            }
        }
    };

    // ---- stencil roles: thread -> one position of the 16 x 32 tile ----
    const int lx = tid & 31, ly = tid >> 5;
    const int gyo = 2 * iy0 + ly, gxo = 2 * ix0 + lx;
    const bool pvalid = gyo < Ho && gxo < Wo && (ly >= 1 || iy0 == 0) && (ly <= OY - 2 || gyo == Ho - 1) &&
                        (lx >= 1 || ix0 == 0) && (lx <= OX - 2 || gxo == Wo - 1);
    const size_t pout = (size_t)gyo * Wo + gxo;
    const int cbase = ly * RS + lx * 4;   // LDS row r = tile row r - 1, x index lx = tile column lx - 1
    auto stencil = [&](int plane, f32x2v& k2, f32x2v& k1, f32x2v& k0) {
        const float* base = ct + plane * PS + cbase;
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx) {
            f32x4 v[3][2];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    v[rr][hf] = *reinterpret_cast<const f32x4*>(base + rr * RS + hf * HS + kx * 4);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // uniform addresses: scalar loads, the weights are SGPR-pair operands of v_pk_fma_f32
                const float* w2 = pw + ((2 * 3 + ky) * 3 + kx) * 8;
                const float* w1 = pw + ((1 * 3 + ky) * 3 + kx) * 8;
                const float* w0 = pw + ((0 * 3 + ky) * 3 + kx) * 8;
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
                    const f32x4 q = v[ky][c2 >> 1];
                    const f32x2v d = (c2 & 1) ? (f32x2v){q.z, q.w} : (f32x2v){q.x, q.y};
                    k2 = __builtin_elementwise_fma(d, (f32x2v){w2[2 * c2], w2[2 * c2 + 1]}, k2);
                    k1 = __builtin_elementwise_fma(d, (f32x2v){w1[2 * c2], w1[2 * c2 + 1]}, k1);
                    k0 = __builtin_elementwise_fma(d, (f32x2v){w0[2 * c2], w0[2 * c2 + 1]}, k0);
                }
            }
        }
    };
    // logits leave by raw buffer stores: a position that is not this block's to write gets an offset beyond the
    // descriptor (dropped by the hardware) -- no branch around a VMEM instruction (see conv11_prob16_kernel)
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(cost, (short)0, (int)((size_t)Do * HWo * 4), 0x00020000);
    auto emit = [&](int oz, const f32x2v& sv, bool live) {
        const unsigned off = (pvalid && live) ? (unsigned)(((size_t)oz * HWo + pout) * 4) : 0xFFFFFFFCu;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sv.x + sv.y), crs, (int)off, 0, 0);
    };

    const float pb = pbias[0];
    const f32x2v fresh = {pb, 0.0f};
    f32x2v A = fresh, B = fresh, C = fresh;

    for (int k = tid; k < C_FLOATS / 4; k += 512) reinterpret_cast<f32x4*>(ct)[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < 2 * 9 * 64; k += 512) reinterpret_cast<f32x4*>(bpan)[k] = reinterpret_cast<const f32x4*>(bp)[k];
    // VMEM order of the prologue = the order a step leaves behind (input chunk 0 oldest, then the skip values): the wait
    // in front of the first store_a() of a step is then a counted vmcnt on both paths into the loop
    load_a(i_first, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_skip(i_first, i_first >= za, i_first < zb);
    __syncthreads();

#pragma unroll 1
    for (int i = i_first; i <= i_last; ++i) {
        const bool de = i >= za, dodd = i < zb;       // even / odd conv11 plane of this step wanted
        // ---- wave-private: stage, multiply, scatter, add the skip values ----
        // (lanes of a wave read what other lanes of the same wave wrote: the LDS executes one wave's
        // operations in order; wave_barrier() keeps the compiler from reordering across these points)
        // (sched_barrier: hipcc otherwise hoists the next loads above the use of the previous ones, into fresh
        // registers, and then waits for vmcnt(0) -- the loads it has just issued)
        store_a();
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        load_a(i, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_chunk(0, de, dodd);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        store_a();                                   // after chunk 0's reads in this wave's LDS order
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        load_a(min(i + 1, i_last), 0);               // unconditional (after the last step: a re-read nobody uses)
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(1, de, dodd);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 2 ? !de : !dodd) continue;
            float* dst = ct + (c >> 1) * PS + (c & 1) * RS + sbase0;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[8 * e] = fmaxf(acc[c][e] + bv, 0.0f);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pz = 0; pz < 2; ++pz) {
            if (pz ? !dodd : !de) continue;
            f32x4 t[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) t[j] = *reinterpret_cast<const f32x4*>(ct + sl[2 * pz + j]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(ct + sl[2 * pz + j]) =
                    ((sok >> (2 * pz + j)) & 1u) ? t[j] + sk[2 * pz + j] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        load_skip(min(i + 1, i_last), true, i + 1 < zb);   // unconditional
        __syncthreads();
        // ---- the whole tile: stencil ----
        if (de) stencil(0, A, B, C);           // even plane 2i: completes logit 2i - 1
        emit(max(2 * i - 1, 0), A, de && i > za);
        A = fresh;
        if (dodd) stencil(1, B, C, A);         // odd plane 2i + 1: completes logit 2i
        emit(2 * i, B, dodd && de);
        const f32x2v t = A;  // (A, B, C) <- logits (2i+1, 2i+2, 2i+3)
        A = C;
        B = t;
        C = fresh;
        __syncthreads();
    }
    emit(Do - 1, A, zb == Di);   // the volume's last plane has no successor to complete it
}

// ---------------------------------------------------------------------------------------------
// The same kernel for 16-bit storage (fp16 / bf16 volumes; logits stay fp32): the transposed convolution runs on
// v_mfma_f32_16x16x32_{f16,bf16} exactly as deconvg16_mfma_kernel (conv3d_mfma16.hip: same panel, 5 k-steps per
// 8-channel chunk, a lane's A fragment = the 8 channels of one voxel) -- 10 MFMAs per wave and step instead of 72
// fp32 ones, the panel in registers, both chunks of the wave's 3 x 9 voxel halo staged at once (one 16-byte piece per
// lane and chunk); skip values arrive as 16-byte pieces of 8 channels.  Scatter, skip add, the z-scatter stencil
// and its (even, odd) packed sums are the fp32 kernel's.  Replaces deconvg16 (conv11) + prob_lds: the 8-channel
// full-resolution tensor between them (cfg3: 485 MB written, then read) never reaches HBM.
// ---------------------------------------------------------------------------------------------
namespace cpv16 {
using namespace cp;
constexpr int WIN = WSLOTS * 8;              // one chunk, 16-bit elements: [plane 2][row 3][9 voxels] in the win_slot() layout
constexpr int NPC = 2 * 27;                  // 16-byte pieces of it: [plane][3 x 9 voxels]
}  // namespace cpv16

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
template <int DT>
__device__ __forceinline__ f32x4 mfma16v(u32x4v a, u32x4v b, f32x4 c) {
    if (DT == MVS_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8v, a), __builtin_bit_cast(f16x8v, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, a), __builtin_bit_cast(bf16x8v, b), c, 0, 0, 0);
}
template <int DT>
__device__ __forceinline__ void unpack8(u32x4v v, float (&o)[8]) {
    if (DT == MVS_F16) {
        const f16x8v h = __builtin_bit_cast(f16x8v, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
    } else {
        const bf16x8v h = __builtin_bit_cast(bf16x8v, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)h[i];
    }
}
template <int DT>
__global__ __launch_bounds__(512, 4) void conv11_prob16_kernel(
    const void* __restrict__ x, const unsigned short* __restrict__ bp, const float* __restrict__ bias,
    const void* __restrict__ skip, const float* __restrict__ pw, const float* __restrict__ pbias,
    float* __restrict__ cost, int Di, int Hi, int Wi, int ZC, int nbx, int nby) {
    using namespace cpv16;
    static_assert(DT == MVS_F16 || DT == MVS_BF16, "16-bit storage");
    const unsigned short* xs = static_cast<const unsigned short*>(x);
    const unsigned short* sks = static_cast<const unsigned short*>(skip);
    __shared__ __attribute__((aligned(16))) float ct[C_FLOATS];
    __shared__ __attribute__((aligned(16))) unsigned short win[8 * 2 * WIN];   // [wave][chunk][win_slot(plane, row, voxel)]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // blocks are dealt round-robin over the 8 XCDs (blockIdx.x % 8 names the XCD: speed only, never correctness) and
    // every XCD has its own L2: XCD k works through the k-th eighth of the (z chunk, row, column) sequence, so that
    // tiles sharing a halo column / row (and the skip planes both read) run on the same L2 at about the same time
    int b;
    {
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;   // XCD k runs q (+1 if k < rem) blocks
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ix0 = bx * (IX - 1), iy0 = by * (IY - 1);
    const int za = bz * ZC, zb = min(za + ZC, Di);
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t Vin = (size_t)Di * Hi * Wi, HWi = (size_t)Hi * Wi, HWo = (size_t)Ho * Wo;
    const int i_first = za > 0 ? za - 1 : 0, i_last = zb < Di ? zb : Di - 1;

    // ---- producer roles (wave-private) ----
    const int tx = wave & 1, ty = wave >> 1;            // M-tile: input rows 2 ty .. +1, columns 8 tx .. +7
    unsigned short* my = win + wave * 2 * WIN;
    // staging: lane -> one 16-byte piece (8 channels of one voxel) of each chunk: [plane hz][row hr][column hc]
    const int pv = lane < NPC ? lane : NPC - 1;
    const int phz = pv / 27, pvv = pv % 27, phr = pvv / 9, phc = pvv % 9;
    const int pgy = iy0 + 2 * ty + phr, pgx = ix0 + 8 * tx + phc;
    const bool pin = lane < NPC && pgy < Hi && pgx < Wi;
    const size_t prel = pin ? (((size_t)phz * Hi + pgy) * Wi + pgx) * 8 : 0;
    const int ploff = win_slot(phz * 3 + phr, phc) * 8;
    u32x4v pre[2];
    bool pre_z1 = true;
    auto load_a = [&](int i) {    // both chunks of input planes i, i + 1; raw loads, masked when they go to LDS
        pre_z1 = i + 1 < Di;
        const bool ok = pin && (pre_z1 || phz == 0);
#pragma unroll
        for (int c = 0; c < 2; ++c)
            pre[c] = *reinterpret_cast<const u32x4v*>(xs + (ok ? ((size_t)c * Vin + (size_t)i * HWi) * 8 + prel : (size_t)0));
    };
    auto store_a = [&]() {
        const bool ok = pin && (pre_z1 || phz == 0);
        if (lane < NPC) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                *reinterpret_cast<u32x4v*>(my + c * WIN + ploff) = ok ? pre[c] : (u32x4v){0u, 0u, 0u, 0u};
        }
    };
    // MFMA lane roles (deconvg16_mfma_kernel): row r -> input voxel (r >> 3, r & 7) of the M-tile; g: dx = g & 1,
    // (z, y)-tap combo g >> 1 (deconv16_tap); a lane's A fragment = the 8 channels of one voxel; column n = r -> (px, co)
    const int r = lane & 15, g = lane >> 4;
    int koff[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const Deconv16Tap t0 = deconv16_tap(ks, 0), t1 = deconv16_tap(ks, 1);
        koff[ks] = win_slot(((g >> 1) ? (t1.dz * 3 + t1.dy) : (t0.dz * 3 + t0.dy)) + (r >> 3), (r & 7) + (g & 1)) * 8;
    }
    const int px = r >> 3, co = r & 7;
    const float bv = bias[co];
    // element e of the accumulators: input voxel m = 4 g + e -> tile column 8 tx + (m & 7) = 8 tx + 4 (g & 1) + e
    const int sbase0 = (2 * (2 * ty + (g >> 1)) + 1) * RS + (co >> 2) * HS + (2 * (8 * tx + 4 * (g & 1)) + px + 1) * 4 + (co & 3);
    // skip pieces of the wave's strip (8 channels = 16 bytes each): lane -> column xx, row rr; j -> plane
    const int sxx = lane & 15, srr = lane >> 4;
    const int soy = 4 * ty + srr, sox = 16 * tx + sxx;
    const bool sok = 2 * iy0 + soy < Ho && 2 * ix0 + sox < Wo;
    const size_t srel = sok ? ((size_t)(2 * iy0 + soy) * Wo + 2 * ix0 + sox) * 8 : 0;
    const int sl = (soy + 1) * RS + (sox + 1) * 4;
    u32x4v sk[2];
    auto load_skip = [&](int i, bool de, bool dodd) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = sok && (j ? dodd : de);
            sk[j] = *reinterpret_cast<const u32x4v*>(sks + (ok ? (size_t)(2 * i + j) * HWo * 8 + srel : (size_t)0));   // masked when added
        }
    };
    // the whole panel (2 chunks x 5 k-steps) in registers
    u32x4v breg[2][5];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) breg[c][ks] = reinterpret_cast<const u32x4v*>(bp)[(c * 5 + ks) * 64 + lane];
    f32x4 acc[4];
    auto mfma_all = [&]() {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                constexpr int dummy = 0;
                (void)dummy;
                const int cls = deconv16_tap(ks, 0).cls;
                const u32x4v a = *reinterpret_cast<const u32x4v*>(my + c * WIN + koff[ks]);
                acc[cls] = mfma16v<DT>(a, breg[c][ks], acc[cls]);
            }
    };

    // ---- stencil roles: thread -> one position of the 16 x 32 tile ----
    const int lx = tid & 31, ly = tid >> 5;
    const int gyo = 2 * iy0 + ly, gxo = 2 * ix0 + lx;
    const bool pvalid = gyo < Ho && gxo < Wo && (ly >= 1 || iy0 == 0) && (ly <= OY - 2 || gyo == Ho - 1) &&
                        (lx >= 1 || ix0 == 0) && (lx <= OX - 2 || gxo == Wo - 1);
    const size_t pout = (size_t)gyo * Wo + gxo;
    const int cbase = ly * RS + lx * 4;   // LDS row r = tile row r - 1, x index lx = tile column lx - 1
    auto stencil = [&](int plane, f32x2v& k2, f32x2v& k1, f32x2v& k0) {
        const float* base = ct + plane * PS + cbase;
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx) {
            f32x4 v[3][2];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    v[rr][hf] = *reinterpret_cast<const f32x4*>(base + rr * RS + hf * HS + kx * 4);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // uniform addresses: scalar loads, the weights are SGPR-pair operands of v_pk_fma_f32
                const float* w2 = pw + ((2 * 3 + ky) * 3 + kx) * 8;
                const float* w1 = pw + ((1 * 3 + ky) * 3 + kx) * 8;
                const float* w0 = pw + ((0 * 3 + ky) * 3 + kx) * 8;
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
                    const f32x4 q = v[ky][c2 >> 1];
                    const f32x2v d = (c2 & 1) ? (f32x2v){q.z, q.w} : (f32x2v){q.x, q.y};
                    k2 = __builtin_elementwise_fma(d, (f32x2v){w2[2 * c2], w2[2 * c2 + 1]}, k2);
                    k1 = __builtin_elementwise_fma(d, (f32x2v){w1[2 * c2], w1[2 * c2 + 1]}, k1);
                    k0 = __builtin_elementwise_fma(d, (f32x2v){w0[2 * c2], w0[2 * c2 + 1]}, k0);
                }
            }
        }
    };
    // logits leave by raw buffer stores: a position that is not this block's to write gets an offset beyond the
    // descriptor (dropped by the hardware) -- no branch around a VMEM instruction, so hipcc keeps COUNTED vmcnt waits
    // for the loads requested a step ahead (round 4: with conditional loads / stores the ISA had vmcnt(1) / vmcnt(0)
    // right behind the prefetch of the NEXT step)
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(cost, (short)0, (int)((size_t)Do * HWo * 4), 0x00020000);
    auto emit = [&](int oz, const f32x2v& sv, bool live) {
        const unsigned off = (pvalid && live) ? (unsigned)(((size_t)oz * HWo + pout) * 4) : 0xFFFFFFFCu;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sv.x + sv.y), crs, (int)off, 0, 0);
    };

    const float pb = pbias[0];
    const f32x2v fresh = {pb, 0.0f};
    f32x2v A = fresh, B = fresh, C = fresh;

    for (int k = tid; k < C_FLOATS / 4; k += 512) reinterpret_cast<f32x4*>(ct)[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // VMEM order of the prologue = the order a step leaves behind (inputs oldest, then the skip values): the wait in
    // front of store_a() is then a counted vmcnt on both paths into the loop
    load_a(i_first);
    __builtin_amdgcn_sched_barrier(0);
    load_skip(i_first, i_first >= za, i_first < zb);
    __syncthreads();

#pragma unroll 1
    for (int i = i_first; i <= i_last; ++i) {
        const bool de = i >= za, dodd = i < zb;       // even / odd conv11 plane of this step wanted
        // ---- wave-private: stage, multiply, scatter, add the skip values ----
        // (lanes of a wave read what other lanes of the same wave wrote: the LDS executes one wave's
        // operations in order; wave_barrier() keeps the compiler from reordering across these points)
        store_a();                                   // both chunks of planes i, i + 1 (requested a step ago)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);           // (hipcc otherwise hoists the next loads above the wait for these)
        load_a(min(i + 1, i_last));                  // unconditional (after the last step: a re-read nobody uses)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_all();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 2 ? !de : !dodd) continue;
            float* dst = ct + (c >> 1) * PS + (c & 1) * RS + sbase0;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[8 * e] = fmaxf(acc[c][e] + bv, 0.0f);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pz = 0; pz < 2; ++pz) {
            if (pz ? !dodd : !de) continue;
            float sv[8];
            unpack8<DT>(sk[pz], sv);
            float* d = ct + pz * PS + sl;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(d), t1 = *reinterpret_cast<const f32x4*>(d + HS);
            const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(d) = sok ? t0 + (f32x4){sv[0], sv[1], sv[2], sv[3]} : z4;
            *reinterpret_cast<f32x4*>(d + HS) = sok ? t1 + (f32x4){sv[4], sv[5], sv[6], sv[7]} : z4;
        }
        load_skip(min(i + 1, i_last), true, i + 1 < zb);   // unconditional
        __syncthreads();
        // ---- the whole tile: stencil ----
        if (de) stencil(0, A, B, C);           // even plane 2i: completes logit 2i - 1
        emit(max(2 * i - 1, 0), A, de && i > za);
        A = fresh;
        if (dodd) stencil(1, B, C, A);         // odd plane 2i + 1: completes logit 2i
        emit(2 * i, B, dodd && de);
        const f32x2v t = A;  // (A, B, C) <- logits (2i+1, 2i+2, 2i+3)
        A = C;
        B = t;
        C = fresh;
        __syncthreads();
    }
    emit(Do - 1, A, zb == Di);   // the volume's last plane has no successor to complete it
}

// ---------------------------------------------------------------------------------------------
// fp32 storage, SPLIT OPERANDS (round 4): the fp32 kernel's data flow with the transposed convolution on
// v_mfma_f32_16x16x32_bf16 in the tap scheme of the 16-bit kernel.  Every fp32 input value is written, on its way into
// the wave's private LDS window, as the sum of three bf16 numbers (split_ops.h), the folded weights likewise on the
// host (pack_split_panels(9): three panels in the bf16 layout), and a product is the six leading cross terms, small
// ones first, accumulated in fp32 -- fp32-equivalent (dropped terms <= 2^-26 of a product; csrc/conv0_split.hip), the
// bounds of the fp32 kernel hold unchanged.  60 bf16 MFMAs (960 cycles of a pipe that leaves half its issue slots to
// the vector units) instead of 72 fp32 ones (2,304 cycles of the ALUs the stencil needs too) per wave and step.
// LDS: the three panels (30 KB) are shared by the block, the windows hold three pieces (23 KB) -- so that two blocks
// still fit a CU, the conv11 tile holds ONE plane: even plane -> stencil -> odd plane -> stencil, four block barriers
// per step instead of two; the odd plane's accumulators wait in registers.
// ---------------------------------------------------------------------------------------------
namespace cpvs {
using namespace cp;
constexpr int WIN = WSLOTS * 8;              // one piece of one chunk, bf16 elements: [plane 2][row 3][9 voxels] in the win_slot() layout
constexpr int NPC = 2 * 27;                  // voxels of it
constexpr int PAN = 2 * 5 * 64 * 8;          // one piece of the panel: [chunk 2][k-step 5][lane 64][8] 16-bit elements
}  // namespace cpvs

__global__ __launch_bounds__(512, 4) void conv11_prob_split_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ bp, const float* __restrict__ bias,
    const float* __restrict__ skip, const float* __restrict__ pw, const float* __restrict__ pbias,
    float* __restrict__ cost, int Di, int Hi, int Wi, int ZC, int nbx, int nby) {
    using namespace cpvs;
    __shared__ __attribute__((aligned(16))) float ct[PS];                          // one conv11 plane
    __shared__ __attribute__((aligned(16))) unsigned short win[8 * 3 * WIN];       // [wave][piece][win_slot(plane, row, voxel)]
    __shared__ __attribute__((aligned(16))) unsigned short bpan[3 * PAN];          // [piece][chunk][k-step][lane][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b;   // XCD k works through the k-th eighth of the (z chunk, row, column) sequence (conv11_prob_priv_kernel)
    {
        const int k = blockIdx.x & 7, q = gridDim.x >> 3, rem = gridDim.x & 7;
        b = k * q + min(k, rem) + (blockIdx.x >> 3);
    }
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ix0 = bx * (IX - 1), iy0 = by * (IY - 1);
    const int za = bz * ZC, zb = min(za + ZC, Di);
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t Vin = (size_t)Di * Hi * Wi, HWi = (size_t)Hi * Wi, HWo = (size_t)Ho * Wo;
    const int i_first = za > 0 ? za - 1 : 0, i_last = zb < Di ? zb : Di - 1;

    // ---- producer roles (wave-private) ----
    const int tx = wave & 1, ty = wave >> 1;            // M-tile: input rows 2 ty .. +1, columns 8 tx .. +7
    unsigned short* my = win + wave * 3 * WIN;
    // staging: lane -> one voxel (8 channels = two 16-byte loads) of the chunk: [plane hz][row hr][column hc]
    const int pv = lane < NPC ? lane : NPC - 1;
    const int phz = pv / 27, pvv = pv % 27, phr = pvv / 9, phc = pvv % 9;
    const int pgy = iy0 + 2 * ty + phr, pgx = ix0 + 8 * tx + phc;
    const bool pin = pgy < Hi && pgx < Wi;
    const size_t prel = pin ? (((size_t)phz * Hi + pgy) * Wi + pgx) * 8 : 0;
    const int ploff = win_slot(phz * 3 + phr, phc) * 8;
    f32x4 pre[2];
    bool pre_z1 = true;
    auto load_a = [&](int i, int c) {    // raw loads (lanes >= NPC repeat the last voxel); masked when they go to LDS
        pre_z1 = i + 1 < Di;
        const bool ok = pin && (pre_z1 || phz == 0);
        const float* src = x + (ok ? ((size_t)c * Vin + (size_t)i * HWi) * 8 + prel : (size_t)0);
        pre[0] = *reinterpret_cast<const f32x4*>(src);
        pre[1] = *reinterpret_cast<const f32x4*>(src + 4);
    };
    auto store_a = [&]() {
        const bool ok = pin && (pre_z1 || phz == 0);
        const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        g_u32x4 p1, p2, p3;
#if MVS_ABLATE == 83
        p1 = __builtin_bit_cast(g_u32x4, ok ? pre[0] : z4); p2 = __builtin_bit_cast(g_u32x4, ok ? pre[1] : z4); p3 = p1;
#else
        gs_split8(ok ? pre[0] : z4, ok ? pre[1] : z4, p1, p2, p3);
#endif
        *reinterpret_cast<g_u32x4*>(my + ploff) = p1;             // lanes >= NPC rewrite the last voxel with its own value
        *reinterpret_cast<g_u32x4*>(my + WIN + ploff) = p2;
        *reinterpret_cast<g_u32x4*>(my + 2 * WIN + ploff) = p3;
    };
    // MFMA lane roles (conv11_prob16_kernel): row r -> input voxel (r >> 3, r & 7) of the M-tile; g: dx = g & 1,
    // (z, y)-tap combo g >> 1; a lane's A fragment = the 8 channels of one voxel; column n = r -> (px, co)
    const int r = lane & 15, g = lane >> 4;
    int koff[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const Deconv16Tap t0 = deconv16_tap(ks, 0), t1 = deconv16_tap(ks, 1);
        koff[ks] = win_slot(((g >> 1) ? (t1.dz * 3 + t1.dy) : (t0.dz * 3 + t0.dy)) + (r >> 3), (r & 7) + (g & 1)) * 8;
    }
    const int px = r >> 3, co = r & 7;
    const float bv = bias[co];
    // element e of the accumulators: input voxel m = 4 g + e -> tile column 8 tx + (m & 7) = 8 tx + 4 (g & 1) + e
    const int sbase0 = (2 * (2 * ty + (g >> 1)) + 1) * RS + (co >> 2) * HS + (2 * (8 * tx + 4 * (g & 1)) + px + 1) * 4 + (co & 3);
    // skip values in the ACCUMULATOR layout (element e of class (pz, py) <-> output voxel (2 i + pz, 2 row + py,
    // 2 (column + e) + px), channel co): one dword per element, 16 lanes x 4 bytes = the two x-adjacent voxels of one
    // input column; they are added in registers, so the tile is written once (the 16-byte read-modify-write of the
    // first form cost 8 LDS operations per wave and step and a write -> read -> write latency chain in each scatter
    // phase).  Raw buffer loads: a position beyond the volume's rows / columns reads a neighbour's value or, past the
    // end of the tensor, zero -- it is masked when the tile is written.
    const int srow = 2 * (iy0 + 2 * ty + (g >> 1)), scol = 2 * (ix0 + 8 * tx + 4 * (g & 1)) + px;   // py = 0, e = 0
    const unsigned sbase = (unsigned)((((size_t)srow * Wo + scol) * 8 + co) * 4);
    const bool vy0 = srow < Ho, vy1 = srow + 1 < Ho;
    bool vx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) vx[e] = scol + 2 * e < Wo;
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(skip), (short)0, (int)((size_t)Do * HWo * 32), 0x00020000);
    float sk[4][4];
    auto load_skip = [&](int i) {   // planes 2 i, 2 i + 1
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned so = (unsigned)((((size_t)(2 * i + (c >> 1)) * Ho + (c & 1)) * Wo) * 32);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sk[c][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, (int)(sbase + 64 * e), (int)so, 0));
        }
    };
    const g_u32x4* bsrc = reinterpret_cast<const g_u32x4*>(bpan) + lane;
    f32x4 acc[4];
    auto mfma_chunk = [&](int c) {
#if MVS_ABLATE == 82
        return;
#endif
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int cls = deconv16_tap(ks, 0).cls;
#if MVS_ABLATE == 86
            const int kq = koff[0];
#else
            const int kq = koff[ks];
#endif
#if MVS_ABLATE == 85
            const int bq = c * 5;
#else
            const int bq = c * 5 + ks;
#endif
            const g_u32x4 a1 = *reinterpret_cast<const g_u32x4*>(my + kq);
            const g_u32x4 a2 = *reinterpret_cast<const g_u32x4*>(my + WIN + kq);
            const g_u32x4 a3 = *reinterpret_cast<const g_u32x4*>(my + 2 * WIN + kq);
            const g_u32x4 b1 = bsrc[(0 * 10 + bq) * 64];
            const g_u32x4 b2 = bsrc[(1 * 10 + bq) * 64];
            const g_u32x4 b3 = bsrc[(2 * 10 + bq) * 64];
            acc[cls] = gs_mfma(a3, b1, acc[cls]);   // the small terms first
            acc[cls] = gs_mfma(a1, b3, acc[cls]);
            acc[cls] = gs_mfma(a2, b2, acc[cls]);
            acc[cls] = gs_mfma(a2, b1, acc[cls]);
            acc[cls] = gs_mfma(a1, b2, acc[cls]);
            acc[cls] = gs_mfma(a1, b1, acc[cls]);
        }
    };
    // conv11 plane pz of this step -> the tile: ReLU(sum + bias) + skip, zero beyond the volume (the prob layer's padding)
    auto scatter = [&](int pz, const f32x4& s0, const f32x4& s1) {   // s0, s1: the sums of classes (pz, py = 0 / 1), bias included
#if MVS_ABLATE == 84
        return;
#endif
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float* dst = ct + c * RS + sbase0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = fmaxf((c ? s1 : s0)[e], 0.0f) + sk[2 * pz + c][e];
                dst[8 * e] = ((c ? vy1 : vy0) && vx[e]) ? v : 0.0f;
            }
        }
    };

    // ---- stencil roles: thread -> one position of the 16 x 32 tile ----
    const int lx = tid & 31, ly = tid >> 5;
    const int gyo = 2 * iy0 + ly, gxo = 2 * ix0 + lx;
    const bool pvalid = gyo < Ho && gxo < Wo && (ly >= 1 || iy0 == 0) && (ly <= OY - 2 || gyo == Ho - 1) &&
                        (lx >= 1 || ix0 == 0) && (lx <= OX - 2 || gxo == Wo - 1);
    const size_t pout = (size_t)gyo * Wo + gxo;
    const int cbase = ly * RS + lx * 4;   // LDS row r = tile row r - 1, x index lx = tile column lx - 1
    auto stencil = [&](f32x2v& k2, f32x2v& k1, f32x2v& k0) {
#if MVS_ABLATE == 81
        return;
#endif
        const float* base = ct + cbase;
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx) {
            f32x4 v[3][2];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    v[rr][hf] = *reinterpret_cast<const f32x4*>(base + rr * RS + hf * HS + kx * 4);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // uniform addresses: scalar loads, the weights are SGPR-pair operands of v_pk_fma_f32
                const float* w2 = pw + ((2 * 3 + ky) * 3 + kx) * 8;
                const float* w1 = pw + ((1 * 3 + ky) * 3 + kx) * 8;
                const float* w0 = pw + ((0 * 3 + ky) * 3 + kx) * 8;
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
                    const f32x4 q = v[ky][c2 >> 1];
                    const f32x2v d = (c2 & 1) ? (f32x2v){q.z, q.w} : (f32x2v){q.x, q.y};
                    k2 = __builtin_elementwise_fma(d, (f32x2v){w2[2 * c2], w2[2 * c2 + 1]}, k2);
                    k1 = __builtin_elementwise_fma(d, (f32x2v){w1[2 * c2], w1[2 * c2 + 1]}, k1);
                    k0 = __builtin_elementwise_fma(d, (f32x2v){w0[2 * c2], w0[2 * c2 + 1]}, k0);
                }
            }
        }
    };
    // logits leave by raw buffer stores: no branch around a VMEM instruction (conv11_prob16_kernel)
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(cost, (short)0, (int)((size_t)Do * HWo * 4), 0x00020000);
    auto emit = [&](int oz, const f32x2v& sv, bool live) {
        const unsigned off = (pvalid && live) ? (unsigned)(((size_t)oz * HWo + pout) * 4) : 0xFFFFFFFCu;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sv.x + sv.y), crs, (int)off, 0, 0);
    };

    const float pb = pbias[0];
    const f32x2v fresh = {pb, 0.0f};
    f32x2v A = fresh, B = fresh, C = fresh;

    for (int k = tid; k < PS / 4; k += 512) reinterpret_cast<f32x4*>(ct)[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < 3 * PAN / 8; k += 512) reinterpret_cast<g_u32x4*>(bpan)[k] = reinterpret_cast<const g_u32x4*>(bp)[k];
    // VMEM order of the prologue = the order a step leaves behind (input chunk 0 oldest, then the skip values): the
    // waits in front of store_a() are then COUNTED vmcnt on both paths into the loop
    load_a(i_first, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_skip(i_first);
    __syncthreads();

    // (sched_barrier: hipcc otherwise hoists the next loads above the use of the previous ones, into fresh registers,
    // and then has to wait for vmcnt(0) -- the loads it has just issued.  The LDS executes one wave's operations in
    // order; wave_barrier() keeps the compiler from reordering across these points.)
    auto stage = [&](int i_next, int c_next) {    // the chunk in `pre` -> the wave's window; request the next one
        store_a();
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        load_a(i_next, c_next);
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int i = i_first; i <= i_last; ++i) {
        const bool de = i >= za, dodd = i < zb;       // even / odd conv11 plane of this step wanted
        stage(i, 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = (f32x4){bv, bv, bv, bv};   // the bias (a lane's four elements share co)
        mfma_chunk(0);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stage(min(i + 1, i_last), 0);                // after chunk 0's reads in this wave's LDS order; unconditional
        mfma_chunk(1);
        if (de) scatter(0, acc[0], acc[1]);
        SPLIT_SYNC_A();
        if (de) stencil(A, B, C);              // even plane 2i: completes logit 2i - 1
        emit(max(2 * i - 1, 0), A, de && i > za);
        A = fresh;
        SPLIT_SYNC_B();
        if (dodd) scatter(1, acc[2], acc[3]);
        load_skip(min(i + 1, i_last));                // unconditional
        SPLIT_SYNC_B();
        if (dodd) stencil(B, C, A);            // odd plane 2i + 1: completes logit 2i
        emit(2 * i, B, dodd && de);
        const f32x2v t = A;  // (A, B, C) <- logits (2i+1, 2i+2, 2i+3)
        A = C;
        B = t;
        C = fresh;
        SPLIT_SYNC_A();
    }
    emit(Do - 1, A, zb == Di);   // the volume's last plane has no successor to complete it
}

bool conv11_prob_enabled(int dtype) {   // MVS_FUSE_PROB=0: conv11 and prob as two launches (A/B runs)
    static const bool on = [] {
        const char* e = getenv("MVS_FUSE_PROB");
        return !(e && e[0] == '0');
    }();
    // 16-bit storage: the fused kernel runs the transposed convolution on the 16-bit matrix cores, so it stands in
    // for the 16-bit MFMA layer kernels only (MVS_MFMA16=0 = fp32 arithmetic on the narrowed operands: two launches)
    return on && (dtype == MVS_F32 || mfma16_enabled());
}

// MVS_TAIL_SPLIT=0: the fp32-MFMA form of the fused tail (conv11_prob_priv_kernel)
static bool tail_split_enabled() {
    static const bool on = [] {
        const char* e = getenv("MVS_TAIL_SPLIT");
        return !(e && e[0] == '0');
    }();
    return on;
}

int launch_conv11_prob(const void* x, const void* skip, float* cost, const float* blob, int Di, int Hi, int Wi,
                       int dtype, hipStream_t s) {
    using namespace cp;
    if (dtype != MVS_F32 && dtype != MVS_F16 && dtype != MVS_BF16)
        return fail(MVS_ERR_BAD_DTYPE, "conv11_prob: unknown dtype %d", dtype);
    if ((size_t)Di * Hi * Wi * 64 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "conv11_prob: plane exceeds 31-bit offsets");
    const BlobLayout L = blob_layout();
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const int nbx = (Wo - 2 + PX - 1) / PX > 0 ? (Wo - 2 + PX - 1) / PX : 1;
    const int nby = (Ho - 2 + PY - 1) / PY > 0 ? (Ho - 2 + PY - 1) / PY : 1;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    // z chunks of >= 4 input planes (every chunk recomputes one input plane and pays a prologue of about half a step):
    // the split that fills the last round of two-blocks-per-CU best.  (Round 2 took the largest split that keeps all
    // blocks resident at once: fine at cfg2 (60 tiles x 8 chunks = 480 blocks for 512 slots), 57 % at cfg3 (294 tiles,
    // ONE chunk of 128 planes each).)
    const int ntile = nbx * nby, slots = 2 * cus;
    int best = 1;
    double best_eff = 0.0;
    for (int nz = 1; nz <= (Di + 3) / 4; ++nz) {
        const int zc = (Di + nz - 1) / nz, nzc_ = (Di + zc - 1) / zc;
        const long nb = (long)ntile * nzc_;
        const double eff = (double)nb / (double)(((nb + slots - 1) / slots) * slots) * zc / (zc + 1.5);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = nz; }
    }
    int ZC = (Di + best - 1) / best;
    if (ZC < 4) ZC = 4;
    if (ZC > Di) ZC = Di;
    const int nzc = (Di + ZC - 1) / ZC;
    const dim3 grid(nbx * nby * nzc);
    // the split-operand kernel addresses the skip tensor through a buffer descriptor (32-bit byte offsets)
    if (dtype == MVS_F32 && tail_split_enabled() && (size_t)Di * Hi * Wi * 256 < ((size_t)1 << 32)) {
        conv11_prob_split_kernel<<<grid, 512, 0, s>>>(
            static_cast<const float*>(x), reinterpret_cast<const unsigned short*>(blob + L.s16_off[9]), blob + L.b_off[9],
            static_cast<const float*>(skip), blob + L.w_off[10], blob + L.b_off[10], cost, Di, Hi, Wi, ZC, nbx, nby);
    } else if (dtype == MVS_F32) {
        conv11_prob_priv_kernel<MVS_F32><<<grid, 512, 0, s>>>(x, blob + L.gp_off[9], blob + L.b_off[9], skip,
                                                               blob + L.w_off[10], blob + L.b_off[10], cost, Di, Hi, Wi,
                                                               ZC, nbx, nby);
    } else {
        const unsigned short* bp16 = reinterpret_cast<const unsigned short*>(blob + L.h16_off[dtype == MVS_F16 ? 0 : 1][9]);
        if (dtype == MVS_F16)
            conv11_prob16_kernel<MVS_F16><<<grid, 512, 0, s>>>(x, bp16, blob + L.b_off[9], skip, blob + L.w_off[10],
                                                               blob + L.b_off[10], cost, Di, Hi, Wi, ZC, nbx, nby);
        else
            conv11_prob16_kernel<MVS_BF16><<<grid, 512, 0, s>>>(x, bp16, blob + L.b_off[9], skip, blob + L.w_off[10],
                                                                blob + L.b_off[10], cost, Di, Hi, Wi, ZC, nbx, nby);
    }
    return check_hip(hipGetLastError(), "conv11_prob launch");
}

}  // namespace mvs
