// warp_conv0_fused.hip -- homography warp + variance cost volume + conv0 in ONE kernel (gfx950).
//
// Replaces models/module.py:96-139 (homo_warping), models/mvsnet.py:145-177 (variance volume) and
// conv0 of CostRegNet (models/mvsnet.py:36,65; ConvBnReLU3D of models/module.py:26-33) without ever
// writing the 32-channel variance volume to HBM (503 MB written + ~900 MB re-read per map at the
// quoted config when the two stages run as separate kernels).
//
// A block (768 threads) owns an 8(y) x 16(x) pixel tile and marches through ZL depth planes.  Its
// 12 waves are specialised:
//   * waves 4-11 (producers, 512 threads) compute the variance of ONE halo plane (10 x 18 voxels x
//     32 channels) per step -- one task = (voxel, channel half, plane pair), 720 tasks in two
//     passes: bilinear gathers of the N-1 source views from the C8-planar features, sum and sum
//     of squares in registers, exactly the arithmetic of warp_variance_kernel -- written into a
//     4-slot LDS ring (zeros outside the volume = conv0's zero padding);
//   * waves 0-3 (consumers) run conv0 for the plane produced three steps earlier on
//     v_mfma_f32_16x16x4_f32 straight from the ring (planes z-1, z, z+1) in the Toeplitz-pair
//     formulation of conv3d_mfma.hip (N = 2 x-adjacent outputs x 8 channels): wave c takes the
//     K-chunk of channels 8c..8c+7 and keeps that chunk's B panel (18 k-steps, 72 VGPRs) in
//     registers for the whole march; the four partial accumulators are summed through LDS, then
//     bias + ReLU and the store of the 8-channel output plane.  (The 4x4x1 MFMA form, faster for
//     the stand-alone conv0, is wrong here: its 8-cycle instructions saturate the SIMD's vector
//     issue port and starve the producer waves' VALU work.)
// The gathers load the vector-L1 path and the VALU, conv0 the matrix pipe, so the two halves of
// the block overlap instead of running back to back; the in-plane halo makes the gather work
// 1.41x that of the unfused kernel, the depth halo 2/ZL more.
//
// LDS: ring 4 x (4 chunks x 180 voxels x 32 B) = 92 KB + reduction buffer 16 KB + 32 KB of B
// fragments -> one block per CU.
#include <cstdlib>

#include "mvs_internal.h"
#include "storage.h"
#include "warp_common.h"

namespace mvs {

namespace fz {
constexpr int TY = 8, TX = 16;            // output tile
constexpr int HY = TY + 2, HX = TX + 2;   // halo plane
constexpr int NVOX = HY * HX;             // 180
constexpr int PLANE_FLOATS = 4 * NVOX * 8;  // [chunk][voxel][8]
constexpr int RING = 4;
constexpr int KS = 18;                             // k-steps of 16 per chunk (36 pair taps x 8 ci / 16)
constexpr int RED_FLOATS = 4 * 4 * 64 * 4;         // partials of the 4 consumer waves: [w][M-tile][lane][4]
constexpr int KS_REG = 10;                         // k-steps whose B fragments stay in registers
constexpr int KS_LDS = KS - KS_REG;                // the rest sit in LDS (register budget: 168 VGPRs)
constexpr int BL_FLOATS = 4 * KS_LDS * 64 * 4;     // [consumer wave][k-step][lane][4]
constexpr int LDS_FLOATS = RING * PLANE_FLOATS + RED_FLOATS + BL_FLOATS;
constexpr int ZL = 24;                    // output planes per block
}  // namespace fz

// LDS float offset of channel-half `half` (0: ch 0-3, 1: ch 4-7) of halo voxel v in a chunk plane.
// Halves of odd groups of 8 voxels are swapped: 16 consecutive voxels then hit 16 distinct 16-byte
// bank slots (conflict-free ds_read_b128 at a 32-byte voxel stride).
__device__ __forceinline__ int fz_vox_off(int v, int half) { return v * 8 + ((half ^ ((v >> 3) & 1)) * 4); }

template <int DT>
__global__ __launch_bounds__(768) void warp_conv0_fused_kernel(
    const float* __restrict__ feats_p,  // [4][N][h][w][8]
    const float* __restrict__ rt,       // [(N-1)][12]
    const float* __restrict__ dv,       // [D]
    const float* __restrict__ bp,       // conv0 Toeplitz-pair panel [4 chunks][18][64 lanes][4]
    const float* __restrict__ bias,     // [8]
    void* __restrict__ y,               // [D][h][w][8] storage dtype DT
    int N, int D, int h, int w) {
    using namespace fz;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ring = lds;
    float* red = lds + RING * PLANE_FLOATS;
    float* blds = red + RED_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tilesx = (w + TX - 1) / TX, tilesy = (h + TY - 1) / TY;
    int b = blockIdx.x;
    const int bx = b % tilesx; b /= tilesx;
    const int by = b % tilesy;
    const int bz = b / tilesy;
    const int x0 = bx * TX, y0 = by * TY, zs = bz * ZL;
    const int ze = min(zs + ZL, D);
    const int nz = ze - zs;
    const int hw = h * w;

    const bool producer = wave >= 4;

    // ---------------------------------------------------------------- producer state
    // task = (plane pair, halo voxel, half): 720 tasks over the 512 producer threads, 2 passes
    const int ptid = tid - 256;
    // ---------------------------------------------------------------- consumer state
    // A fragment lane (r = lane&15 -> pair (row r>>3, pair r&7) of an M-tile of 2 rows x 8 pairs,
    // g = lane>>4): k-step ks covers pair-taps 2ks + (g>>1), channels 4(g&1)..+3 of the chunk
    const int cr = lane & 15, cg = lane >> 4;
    const int crow = cr >> 3, cxp = 2 * (cr & 7) + (cg >> 1), chalf = cg & 1;
    const int chunk = wave & 3;  // consumer wave c handles channels 8c..8c+7
    f32x4 breg[KS_REG];          // this wave's B panel, resident for the whole march
    const f32x4* bl = reinterpret_cast<const f32x4*>(blds) + (size_t)chunk * KS_LDS * 64 + lane;
    if (!producer) {
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(bp) + (size_t)chunk * KS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS_REG; ++ks) breg[ks] = bsrc[ks * 64];
#pragma unroll
        for (int ks = 0; ks < KS_LDS; ++ks)
            reinterpret_cast<f32x4*>(blds)[((size_t)chunk * KS_LDS + ks) * 64 + lane] = bsrc[(KS_REG + ks) * 64];
    }

    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const size_t plane_stride = (size_t)N * hw * 8;
    const float inv_n = 1.0f / (float)N;

    for (int t = 0; t < nz + 3; ++t) {
        if (producer) {
            // ---- fill halo plane P = zs - 1 + t into ring slot t & 3
            const int P = zs - 1 + t;
            float* slot = ring + (t & 3) * PLANE_FLOATS;
            if (t <= nz + 1) {
              const bool zin = P >= 0 && P < D;
              const float depth = dv[min(max(P, 0), D - 1)];
#pragma unroll 1
              for (int pass = 0; pass < 2; ++pass) {
                const int task = ptid + pass * 512;
                if (task >= 4 * NVOX) break;
                const int ppair = task / (2 * NVOX), prem = task - ppair * 2 * NVOX;
                const int half = prem & 1, v = prem >> 1;
                const int phy = v / HX, phx = v - phy * HX;
                const int gy = y0 + phy - 1, gx = x0 + phx - 1;
                f32x4 out[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) out[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (zin && gy >= 0 && gy < h && gx >= 0 && gx < w) {
                    const int p = gy * w + gx;
                    const float* fpl = feats_p + (size_t)(2 * ppair) * plane_stride + 4 * half;
                    float4 S[2], Q[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float4 r = *reinterpret_cast<const float4*>(fpl + q * plane_stride + (size_t)p * 8);
                        S[q] = r;
                        Q[q] = make_float4(r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w);
                    }
                    const float fx = (float)gx, fy = (float)gy;
                    for (int vw = 1; vw < N; ++vw) {
                        const float* r = rt + (size_t)(vw - 1) * 12;
                        const float qx = fmaf(r[0], fx, fmaf(r[1], fy, r[2]));
                        const float qy = fmaf(r[3], fx, fmaf(r[4], fy, r[5]));
                        const float qz = fmaf(r[6], fx, fmaf(r[7], fy, r[8]));
                        const Samp sm = make_samp(qx, qy, qz, r[9], r[10], r[11], depth, sx, sy, h, w,
                                                  0, 0, w, h);
                        const float* f0 = fpl + (size_t)vw * hw * 8;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float* f = f0 + q * plane_stride;
                            const float4 a = *reinterpret_cast<const float4*>(f + (size_t)sm.o00 * 8);
                            const float4 bb = *reinterpret_cast<const float4*>(f + (size_t)sm.o01 * 8);
                            const float4 c = *reinterpret_cast<const float4*>(f + (size_t)sm.o10 * 8);
                            const float4 e = *reinterpret_cast<const float4*>(f + (size_t)sm.o11 * 8);
                            float4 wv;
                            wv.x = fmaf(a.x, sm.w00, fmaf(bb.x, sm.w01, fmaf(c.x, sm.w10, e.x * sm.w11)));
                            wv.y = fmaf(a.y, sm.w00, fmaf(bb.y, sm.w01, fmaf(c.y, sm.w10, e.y * sm.w11)));
                            wv.z = fmaf(a.z, sm.w00, fmaf(bb.z, sm.w01, fmaf(c.z, sm.w10, e.z * sm.w11)));
                            wv.w = fmaf(a.w, sm.w00, fmaf(bb.w, sm.w01, fmaf(c.w, sm.w10, e.w * sm.w11)));
                            accum(S[q], Q[q], wv);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const float4 o = variance4(S[q], Q[q], inv_n);
                        // storage-dtype rounding of the (virtual) variance volume keeps the 16-bit
                        // modes bit-compatible with the unfused path
                        if (DT == MVS_F16) out[q] = (f32x4){(float)(_Float16)o.x, (float)(_Float16)o.y, (float)(_Float16)o.z, (float)(_Float16)o.w};
                        else if (DT == MVS_BF16) out[q] = (f32x4){(float)(__bf16)o.x, (float)(__bf16)o.y, (float)(__bf16)o.z, (float)(__bf16)o.w};
                        else out[q] = (f32x4){o.x, o.y, o.z, o.w};
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    *reinterpret_cast<f32x4*>(slot + (2 * ppair + q) * NVOX * 8 + fz_vox_off(v, half)) = out[q];
              }
            }
        } else if (t >= 3) {
            // ---- conv0 of output plane O = zs + t - 3 from ring planes t-3, t-2, t-1
            f32x4 acc[4];  // M-tile i = rows 2i, 2i+1 of the 8 x 16 tile
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int tap0 = 2 * ks;  // pair taps (kz, ky, kx') with kx' = tap % 4
                const int kz = tap0 / 12, ky = (tap0 / 4) % 3, kx0 = tap0 % 4;
                const float* pl = ring + ((t - 3 + kz) & 3) * PLANE_FLOATS + chunk * NVOX * 8;
                f32x4 a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = *reinterpret_cast<const f32x4*>(
                        pl + fz_vox_off((2 * i + crow + ky) * HX + cxp + kx0, chalf));
                const f32x4 bq = ks < KS_REG ? breg[ks < KS_REG ? ks : 0] : bl[(ks - KS_REG) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4*>(red + ((wave * 4 + i) * 64 + lane) * 4) = acc[i];
        }
        __syncthreads();  // plane t & 3 complete; partial sums of the consumer waves published
        if (t >= 3) {
            // all threads: sum the four K-chunk partials, bias + ReLU, store.  Element q of the
            // 4 x 64 x 4 accumulator image = (M-tile i, lane l, register e); D layout: column
            // n = l&15 = (j, co), row m = 4*(l>>4) + e = pair (row m>>3, pair m&7) of M-tile i.
            const int O = zs + t - 3;
#pragma unroll 1
            for (int q = tid; q < 4 * 64 * 4; q += 768) {
                const float sum = red[q] + red[q + 1024] + red[q + 2048] + red[q + 3072];
                const int e = q & 3, l = (q >> 2) & 63, i = q >> 8;
                const int n = l & 15, jj = n >> 3, co = n & 7;
                const int m = 4 * (l >> 4) + e;
                const int gy = y0 + 2 * i + (m >> 3), gx = x0 + 2 * (m & 7) + jj;
                if (gy < h && gx < w)
                    St<DT>::store1(y, (((size_t)O * h + gy) * w + gx) * 8 + co, fmaxf(sum + bias[co], 0.0f));
            }
        }
        __syncthreads();  // reduction buffer and ring slot (t+1)&3 free again
    }
}

template <int DT>
static int run_fused(const float* feats_p, const float* rt, const float* dv, const float* bp,
                     const float* bias, void* y, int N, int D, int h, int w, hipStream_t s) {
    using namespace fz;
    const int nb = ((w + TX - 1) / TX) * ((h + TY - 1) / TY) * ((D + ZL - 1) / ZL);
    const size_t lds_bytes = (size_t)LDS_FLOATS * sizeof(float);
    // > 64 KB of dynamic LDS needs the opt-in attribute (per device; cheap, so set on every call)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&warp_conv0_fused_kernel<DT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return check_hip(e, "warp_conv0_fused: hipFuncSetAttribute");
    warp_conv0_fused_kernel<DT><<<nb, 768, lds_bytes, s>>>(feats_p, rt, dv, bp, bias, y, N, D, h, w);
    return check_hip(hipGetLastError(), "warp_conv0_fused launch");
}

int launch_warp_conv0_fused(const float* feats_p, const float* rt, const float* dv, const float* bp,
                            const float* bias, void* y, int N, int D, int h, int w, int dtype,
                            hipStream_t s) {
    if (h < 2 || w < 2) return fail(MVS_ERR_BAD_SHAPE, "warp_conv0_fused: h,w must be >= 2");
    MVS_DISPATCH_DTYPE(dtype, (run_fused<DT>(feats_p, rt, dv, bp, bias, y, N, D, h, w, s)))
}

}  // namespace mvs
