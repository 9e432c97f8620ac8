// warp_conv0_fused.hip -- homography warp + variance cost volume + conv0 in ONE kernel (gfx950).
//
// Replaces models/module.py:96-139 (homo_warping), models/mvsnet.py:145-177 (variance volume) and
// conv0 of CostRegNet (models/mvsnet.py:36,65; ConvBnReLU3D of models/module.py:26-33) without ever
// writing the 32-channel variance volume to HBM (503 MB written + ~900 MB re-read per map at the
// quoted config when the two stages run as separate kernels).
//
// A block (1024 threads) owns an 8(y) x 16(x) pixel tile and marches through ZL depth planes.  Its
// 16 waves are specialised:
//   * waves 4-15 (producers, 768 threads) compute the variance of ONE halo plane (10 x 18 voxels x
//     32 channels) per step -- one task = (voxel, channel half, plane pair): bilinear gathers of
//     the N-1 source views from the C8-planar features, sum and sum of squares in registers,
//     exactly the arithmetic of warp_variance_kernel -- and write it into a 4-slot LDS ring (zeros
//     outside the volume = conv0's zero padding); three producer waves per SIMD keep enough
//     gathers in flight to hide their latency;
//   * waves 0-3 (consumers) run conv0 for the plane produced three steps earlier on
//     v_mfma_f32_4x4x1_16b_f32 straight from the ring (planes z-1, z, z+1): wave c takes the
//     K-chunk of channels 8c..8c+7 (27 taps x 8 channels), the four partial accumulators are
//     summed through LDS, then bias + ReLU and the store of the 8-channel output plane.
// The gathers load the vector-L1 path and the VALU, conv0 the matrix pipe, so the two halves of
// the block overlap instead of running back to back; the in-plane halo makes the gather work
// 1.41x that of the unfused kernel, the depth halo 2/ZL more.
//
// LDS: ring 4 x (4 chunks x 180 voxels x 32 B) = 92 KB, conv0 weights 27.6 KB (all chunks, 4x4x1
// panel layout of conv3d_mfma.hip), reduction buffer 12 KB  ->  one block per CU.
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
constexpr int WQ_FLOATS = 4 * 27 * 2 * 2 * 4 * 4;  // all chunks of the 4x4x1 weight panel
constexpr int RED_FLOATS = 3 * 4 * 64 * 4;         // partials of consumer waves 1..3: [w][mg*2+nt][lane][4]
constexpr int LDS_FLOATS = RING * PLANE_FLOATS + WQ_FLOATS + RED_FLOATS;
constexpr int ZL = 24;                    // output planes per block
}  // namespace fz

// LDS float offset of channel-half `half` (0: ch 0-3, 1: ch 4-7) of halo voxel v in a chunk plane.
// Halves of odd groups of 8 voxels are swapped: 16 consecutive voxels then hit 16 distinct 16-byte
// bank slots (conflict-free ds_read_b128 at a 32-byte voxel stride).
__device__ __forceinline__ int fz_vox_off(int v, int half) { return v * 8 + ((half ^ ((v >> 3) & 1)) * 4); }

template <int DT>
__global__ __launch_bounds__(1024) void warp_conv0_fused_kernel(
    const float* __restrict__ feats_p,  // [4][N][h][w][8]
    const float* __restrict__ rt,       // [(N-1)][12]
    const float* __restrict__ dv,       // [D]
    const float* __restrict__ bq,       // conv0 4x4x1 panel [4][27][2][2][4][4]
    const float* __restrict__ bias,     // [8]
    void* __restrict__ y,               // [D][h][w][8] storage dtype DT
    int N, int D, int h, int w) {
    using namespace fz;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ring = lds;
    float* wq = lds + RING * PLANE_FLOATS;
    float* red = wq + WQ_FLOATS;

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

    // conv0 weights -> LDS (once per block)
    for (int i = tid; i < WQ_FLOATS / 4; i += 1024)
        reinterpret_cast<f32x4*>(wq)[i] = reinterpret_cast<const f32x4*>(bq)[i];

    const bool producer = wave >= 4;

    // ---------------------------------------------------------------- producer state
    // task = (plane pair, halo voxel, half): 720 tasks over the 768 producer threads
    const int ptid = tid - 256;
    const int ppair = ptid / (2 * NVOX), prem = ptid - ppair * 2 * NVOX;
    const int phalf = prem & 1, pv = prem >> 1;
    const int phy = pv / HX, phx = pv - phy * HX;
    const bool ptask = producer && ptid < 4 * NVOX;
    // ---------------------------------------------------------------- consumer state
    // lane -> voxel of a 64-voxel M-group: 4 rows x 16 x; M-group mg covers rows 4mg..4mg+3
    const int crow = lane >> 4, cx = lane & 15;
    const int chunk = wave & 3;  // consumer wave c handles channels 8c..8c+7
    f32x4 acc[2][2];

    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const size_t plane_stride = (size_t)N * hw * 8;
    const float inv_n = 1.0f / (float)N;

    __syncthreads();  // weights visible

    for (int t = 0; t < nz + 3; ++t) {
        if (producer) {
            // ---- fill halo plane P = zs - 1 + t into ring slot t & 3
            const int P = zs - 1 + t;
            float* slot = ring + (t & 3) * PLANE_FLOATS;
            if (t <= nz + 1 && ptask) {
                const bool zin = P >= 0 && P < D;
                const float depth = dv[min(max(P, 0), D - 1)];
                const int half = phalf, v = pv;
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
        } else if (t >= 3) {
            // ---- conv0 of output plane O = zs + t - 3 from ring planes t-3, t-2, t-1
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* wb = wq + chunk * (27 * 2 * 2 * 4 * 4) + (lane & 3) * 4;
#pragma unroll 1
            for (int kz = 0; kz < 3; ++kz) {  // not unrolled: bounds the live LDS reads (128-VGPR budget)
                const float* pl = ring + ((t - 3 + kz) & 3) * PLANE_FLOATS + chunk * NVOX * 8;
                const float* wbz = wb + kz * (9 * 2 * 2 * 16);
#pragma unroll 1
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int tap9 = ky * 3 + kx;
                        const int v0 = (crow + ky) * HX + cx + kx;   // M-group 0: rows 0..3
                        const int v1 = v0 + 4 * HX;                  // M-group 1: rows 4..7
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const f32x4 a0 = *reinterpret_cast<const f32x4*>(pl + fz_vox_off(v0, half));
                            const f32x4 a1 = *reinterpret_cast<const f32x4*>(pl + fz_vox_off(v1, half));
                            const f32x4 b0 = *reinterpret_cast<const f32x4*>(wbz + ((tap9 * 2 + half) * 2 + 0) * 16);
                            const f32x4 b1 = *reinterpret_cast<const f32x4*>(wbz + ((tap9 * 2 + half) * 2 + 1) * 16);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                acc[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[k], b0[k], acc[0][0], 0, 0, 0);
                                acc[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[k], b1[k], acc[0][1], 0, 0, 0);
                                acc[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[k], b0[k], acc[1][0], 0, 0, 0);
                                acc[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[k], b1[k], acc[1][1], 0, 0, 0);
                            }
                        }
                    }
            }
            if (wave > 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        *reinterpret_cast<f32x4*>(red + (((wave - 1) * 4 + m * 2 + n) * 64 + lane) * 4) = acc[m][n];
            }
        }
        __syncthreads();  // plane t & 3 complete; partial sums of waves 1..3 published
        if (wave == 0 && t >= 3) {
            const int O = zs + t - 3;
            const int blk = lane >> 2, j = lane & 3;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4 s = acc[m][n];
#pragma unroll
                    for (int wv = 0; wv < 3; ++wv) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(red + ((wv * 4 + m * 2 + n) * 64 + lane) * 4);
                        s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
                    }
                    const float bv = bias[4 * n + j];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int vox = 4 * blk + i;  // voxel of the M-group: (row, x) = (vox >> 4, vox & 15)
                        const int gy = y0 + 4 * m + (vox >> 4), gx = x0 + (vox & 15);
                        if (gy < h && gx < w)
                            St<DT>::store1(y, (((size_t)O * h + gy) * w + gx) * 8 + 4 * n + j,
                                           fmaxf(s[i] + bv, 0.0f));
                    }
                }
        }
        __syncthreads();  // reduction buffer and ring slot (t+1)&3 free again
    }
}

template <int DT>
static int run_fused(const float* feats_p, const float* rt, const float* dv, const float* bq,
                     const float* bias, void* y, int N, int D, int h, int w, hipStream_t s) {
    using namespace fz;
    const int nb = ((w + TX - 1) / TX) * ((h + TY - 1) / TY) * ((D + ZL - 1) / ZL);
    const size_t lds_bytes = (size_t)LDS_FLOATS * sizeof(float);
    // > 64 KB of dynamic LDS needs the opt-in attribute (per device; cheap, so set on every call)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&warp_conv0_fused_kernel<DT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return check_hip(e, "warp_conv0_fused: hipFuncSetAttribute");
    warp_conv0_fused_kernel<DT><<<nb, 1024, lds_bytes, s>>>(feats_p, rt, dv, bq, bias, y, N, D, h, w);
    return check_hip(hipGetLastError(), "warp_conv0_fused launch");
}

int launch_warp_conv0_fused(const float* feats_p, const float* rt, const float* dv, const float* bq,
                            const float* bias, void* y, int N, int D, int h, int w, int dtype,
                            hipStream_t s) {
    if (h < 2 || w < 2) return fail(MVS_ERR_BAD_SHAPE, "warp_conv0_fused: h,w must be >= 2");
    MVS_DISPATCH_DTYPE(dtype, (run_fused<DT>(feats_p, rt, dv, bq, bias, y, N, D, h, w, s)))
}

}  // namespace mvs
