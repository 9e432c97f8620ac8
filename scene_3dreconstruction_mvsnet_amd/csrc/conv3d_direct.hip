// conv3d_direct.hip -- VALU direct 3x3x3 convolution / transposed convolution, channels-last.
//
// Baseline kernels for every CostRegNet layer (reference models/mvsnet.py:35-73 with the
// ConvBnReLU3D block of models/module.py:26-33).  BatchNorm is already folded into the packed
// weights (mvs_pack_weights), so each layer is  y = act(conv(x) + bias) (+ skip).
//
//   x    [CIN/8][Di][Hi][Wi][8]    C8-planar fp32 (see warp_variance.hip)
//   wgt  [27][CIN][COUT]           tap-major packed weights (tap = kd*9 + kh*3 + kw)
//   y    [COUT/8][Do][Ho][Wo][8]   (COUT == 1: plain [Do][Ho][Wo])
//
// One thread = one output voxel x CPT output channels.  Weight addresses are wave-uniform, so
// the compiler streams them through the scalar cache; activations are 16-byte vector loads that
// hit L1/L2 for the 27-fold tap reuse.  The MFMA implicit-GEMM kernels in conv3d_mfma.hip
// replace these on the FLOP-heavy layers; these stay as the generic fallback for every shape.
#include <cstdlib>

#include "mvs_internal.h"
#include "storage.h"

#ifndef MVS_ABLATE
#define MVS_ABLATE 0
#endif

namespace mvs {

template <int CIN, int COUT, int CPT, int STRIDE, bool DECONV, bool RELU, bool SKIP>
__global__ __launch_bounds__(256) void conv3d_direct_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ wgt,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ skip,
                                                            float* __restrict__ y, int Di, int Hi,
                                                            int Wi, int Do, int Ho, int Wo) {
    const size_t nvox = (size_t)Do * Ho * Wo;
    const size_t nvox_in = (size_t)Di * Hi * Wi;
    const int g = blockIdx.y;  // output-channel group (wave-uniform)
    const int co0 = g * CPT;
    int ow, oh, od;
    size_t vi;
    if (DECONV) {
        // blockIdx.z = parity class (pd,ph,pw) of the output voxel: o = 2*i - 1 + k has solutions
        // k = 1 (o even) or k in {0, 2} (o odd), so the tap set is uniform across the block and
        // only the 27/8 useful taps on average are visited.
        const size_t ci = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // index inside the class
        if (ci >= nvox_in) return;
        const int pw = blockIdx.z & 1, ph = (blockIdx.z >> 1) & 1, pd = blockIdx.z >> 2;
        ow = 2 * (int)(ci % Wi) + pw;
        oh = 2 * (int)((ci / Wi) % Hi) + ph;
        od = 2 * (int)(ci / ((size_t)Wi * Hi)) + pd;
        vi = ((size_t)od * Ho + oh) * Wo + ow;
    } else {
        vi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (vi >= nvox) return;
        ow = (int)(vi % Wo); oh = (int)((vi / Wo) % Ho); od = (int)(vi / ((size_t)Wo * Ho));
    }

    float acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = bias[co0 + j];

    const int pcls = DECONV ? (int)blockIdx.z : 0;
    for (int kd = 0; kd < 3; ++kd) {
        int id;
        bool okd;
        if (DECONV) {
            if (((kd & 1) != 0) == (((pcls >> 2) & 1) != 0)) continue;  // wave-uniform parity skip
            const int t = od + 1 - kd;  // o = 2*i - 1 + k
            id = t >> 1;
            okd = (t >= 0) && id < Di;
        } else {
            id = od * STRIDE + kd - 1;
            okd = id >= 0 && id < Di;
        }
        for (int kh = 0; kh < 3; ++kh) {
            int ih;
            bool okh;
            if (DECONV) {
                if (((kh & 1) != 0) == (((pcls >> 1) & 1) != 0)) continue;
                const int t = oh + 1 - kh;
                ih = t >> 1;
                okh = (t >= 0) && ih < Hi;
            } else {
                ih = oh * STRIDE + kh - 1;
                okh = ih >= 0 && ih < Hi;
            }
            for (int kw = 0; kw < 3; ++kw) {
                int iw;
                bool okw;
                if (DECONV) {
                    if (((kw & 1) != 0) == ((pcls & 1) != 0)) continue;
                    const int t = ow + 1 - kw;
                    iw = t >> 1;
                    okw = (t >= 0) && iw < Wi;
                } else {
                    iw = ow * STRIDE + kw - 1;
                    okw = iw >= 0 && iw < Wi;
                }
                if (!(okd && okh && okw)) continue;
                const size_t vin = ((size_t)id * Hi + ih) * Wi + iw;
                const float* wp = wgt + (size_t)((kd * 3 + kh) * 3 + kw) * CIN * COUT + co0;
#pragma unroll
                for (int ci = 0; ci < CIN; ci += 4) {
                    const float4 xv = *reinterpret_cast<const float4*>(
                        x + ((size_t)(ci >> 3) * nvox_in + vin) * 8 + (ci & 7));
#pragma unroll
                    for (int j = 0; j < CPT; ++j) {
                        acc[j] = fmaf(xv.x, wp[(ci + 0) * COUT + j], acc[j]);
                        acc[j] = fmaf(xv.y, wp[(ci + 1) * COUT + j], acc[j]);
                        acc[j] = fmaf(xv.z, wp[(ci + 2) * COUT + j], acc[j]);
                        acc[j] = fmaf(xv.w, wp[(ci + 3) * COUT + j], acc[j]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int co = co0 + j;
        const size_t o = (COUT == 1) ? vi : ((size_t)(co >> 3) * nvox + vi) * 8 + (co & 7);
        float v = acc[j];
        if (RELU) v = fmaxf(v, 0.0f);
        if (SKIP) v += skip[o];  // skip + relu(bn(deconv(x)))   (models/mvsnet.py:69-71)
        y[o] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// ConvTranspose3d k3 s2 p1 op1 (+folded BN) + ReLU + skip, gather form.
//   y[o] += x[i] * w[k]  for  o = 2*i - 1 + k  per axis  =>  even o: (k=1, i=o/2);
//   odd o: (k=0, i=(o+1)/2) and (k=2, i=(o-1)/2).
// blockIdx.z = (pd, ph): parity class of the output z/y, so the z/y tap set is block-uniform.
// A thread owns input voxel (zi, yi, xi) and produces BOTH x-parities (ow = 2xi, 2xi+1): 64 B
// contiguous per thread for 8 channels, fully coalesced across the wave; the x taps are
//   even: x[xi]*w[kw=1]            odd: x[xi]*w[kw=2] + x[xi+1]*w[kw=0].
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int CPT>
__global__ __launch_bounds__(256) void deconv3d_direct_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ wgt,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ skip,
                                                              float* __restrict__ y, int Di, int Hi,
                                                              int Wi) {
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t nvox_in = (size_t)Di * Hi * Wi, nvox = nvox_in * 8;
    const size_t ti = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= nvox_in) return;
    const int co0 = blockIdx.y * CPT;
    const int ph = blockIdx.z & 1, pd = blockIdx.z >> 1;
    const int xi = (int)(ti % Wi), yi = (int)((ti / Wi) % Hi), zi = (int)(ti / ((size_t)Wi * Hi));
    const int od = 2 * zi + pd, oh = 2 * yi + ph;

    float acc0[CPT], acc1[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) { acc0[j] = bias[co0 + j]; acc1[j] = acc0[j]; }

    const bool x1ok = xi + 1 < Wi;
    for (int kd = 0; kd < 3; ++kd) {
        if ((kd & 1) == pd) continue;  // block-uniform
        const int id = (od + 1 - kd) >> 1;
        if (id >= Di) continue;
        for (int kh = 0; kh < 3; ++kh) {
            if ((kh & 1) == ph) continue;
            const int ih = (oh + 1 - kh) >> 1;
            if (ih >= Hi) continue;
            const size_t vin = ((size_t)id * Hi + ih) * Wi + xi;
            const float* w0 = wgt + (size_t)((kd * 3 + kh) * 3 + 0) * CIN * COUT + co0;
            const float* w1 = w0 + CIN * COUT;
            const float* w2 = w1 + CIN * COUT;
#pragma unroll
            for (int ci = 0; ci < CIN; ci += 4) {
                const float* xp = x + ((size_t)(ci >> 3) * nvox_in + vin) * 8 + (ci & 7);
                const float4 a = *reinterpret_cast<const float4*>(xp);
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                if (x1ok) b = *reinterpret_cast<const float4*>(xp + 8);
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    acc0[j] = fmaf(a.x, w1[(ci + 0) * COUT + j], acc0[j]);
                    acc0[j] = fmaf(a.y, w1[(ci + 1) * COUT + j], acc0[j]);
                    acc0[j] = fmaf(a.z, w1[(ci + 2) * COUT + j], acc0[j]);
                    acc0[j] = fmaf(a.w, w1[(ci + 3) * COUT + j], acc0[j]);
                    acc1[j] = fmaf(a.x, w2[(ci + 0) * COUT + j], acc1[j]);
                    acc1[j] = fmaf(a.y, w2[(ci + 1) * COUT + j], acc1[j]);
                    acc1[j] = fmaf(a.z, w2[(ci + 2) * COUT + j], acc1[j]);
                    acc1[j] = fmaf(a.w, w2[(ci + 3) * COUT + j], acc1[j]);
                    acc1[j] = fmaf(b.x, w0[(ci + 0) * COUT + j], acc1[j]);
                    acc1[j] = fmaf(b.y, w0[(ci + 1) * COUT + j], acc1[j]);
                    acc1[j] = fmaf(b.z, w0[(ci + 2) * COUT + j], acc1[j]);
                    acc1[j] = fmaf(b.w, w0[(ci + 3) * COUT + j], acc1[j]);
                }
            }
        }
    }
    const size_t vo = ((size_t)od * Ho + oh) * Wo + 2 * xi;
    static_assert(CPT % 4 == 0, "CPT must be a multiple of 4");
#pragma unroll
    for (int j = 0; j < CPT; j += 4) {
        const int co = co0 + j;
        const size_t o = ((size_t)(co >> 3) * nvox + vo) * 8 + (co & 7);
        const float4 s0 = *reinterpret_cast<const float4*>(skip + o);
        const float4 s1 = *reinterpret_cast<const float4*>(skip + o + 8);
        // skip + relu(bn(deconv(x)))   (models/mvsnet.py:69-71)
        float4 r0, r1;
        r0.x = fmaxf(acc0[j + 0], 0.f) + s0.x; r0.y = fmaxf(acc0[j + 1], 0.f) + s0.y;
        r0.z = fmaxf(acc0[j + 2], 0.f) + s0.z; r0.w = fmaxf(acc0[j + 3], 0.f) + s0.w;
        r1.x = fmaxf(acc1[j + 0], 0.f) + s1.x; r1.y = fmaxf(acc1[j + 1], 0.f) + s1.y;
        r1.z = fmaxf(acc1[j + 2], 0.f) + s1.z; r1.w = fmaxf(acc1[j + 3], 0.f) + s1.w;
        *reinterpret_cast<float4*>(y + o) = r0;
        *reinterpret_cast<float4*>(y + o + 8) = r1;
    }
}

template <int CIN, int COUT, int CPT>
static int run_deconv(const void* x, const void* skip, void* y, const float* wgt, const float* bias,
                      int Di, int Hi, int Wi, hipStream_t s) {
    const size_t nthreads = (size_t)Di * Hi * Wi;
    dim3 grid((unsigned)((nthreads + 255) / 256), COUT / CPT, 4);
    deconv3d_direct_kernel<CIN, COUT, CPT><<<grid, 256, 0, s>>>(
        static_cast<const float*>(x), wgt, bias, static_cast<const float*>(skip),
        static_cast<float*>(y), Di, Hi, Wi);
    return check_hip(hipGetLastError(), "deconv3d_direct launch");
}

// ---------------------------------------------------------------------------------------------
// prob through LDS: a block stages the 6 x 10 x 34 halo tile of the 8-channel input once and
// produces a 4 x 8 x 32 tile of logits (4 z-adjacent outputs per thread).  A global-gather form
// is bound by the vector-L1 path (every voxel is fetched up to 27 times); here
// each voxel is fetched ~2x (halo) and the 54 taps per thread come from LDS.  The two 16-byte
// halves of a voxel are swapped for odd groups of 8 x positions so that the 32-byte voxel
// stride stays conflict-free for ds_read_b128.
// ---------------------------------------------------------------------------------------------
typedef float f32x2p __attribute__((ext_vector_type(2)));
namespace pl {
constexpr int TZ = 4, TY = 8, TX = 32;
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int NPIECE = HZ * HY * HX * 2;
constexpr int PPT = (NPIECE + 255) / 256;  // 16
}  // namespace pl

// Persistent blocks (two per CU) loop over the tiles; the halo tile of the NEXT tile is requested into
// registers before the current tile is multiplied from LDS, so the staging latency -- which bounded the
// one-tile-per-block form at 2.2 TB/s (load everything, wait, write LDS, compute) -- is hidden behind
// the 864 FMAs per thread of the current tile.
template <int DT>
__global__ __launch_bounds__(256, 2) void prob_lds_kernel(const void* __restrict__ x,  // storage DT
                                                          const float* __restrict__ wgt,  // [27][8]
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, int D, int H, int W) {
    using namespace pl;
    __shared__ __attribute__((aligned(16))) float tile[HZ * HY * HX * 8];
    const int tid = threadIdx.x;
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    const int ntiles = nbx * nby * ((D + TZ - 1) / TZ);
    int x0 = 0, y0 = 0, z0 = 0;

    // Staging bookkeeping, ONCE per thread: piece i of this thread is halo voxel (hz, hy, hx), 16-byte
    // half `half`, the same for every tile.  Decomposing p = tid + 256 i by the non-power-of-two tile
    // extents, bounds checks and 64-bit addresses for 16 pieces cost ~400 VALU instructions -- per TILE
    // that was a third of the kernel's vector time (SQ_ACTIVE_INST_VALU = 80 % of its wall time).  Per
    // tile only a scalar base offset and six scalar range limits are left.
    int rel[PPT], lo[PPT];          // element offset relative to the tile's halo origin; LDS float offset
    unsigned pos[PPT];              // hz | hy << 8 | hx << 16 (0xFFFFFF.. = no such piece)
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = tid + i * 256;
        const int half = p & 1, v = p >> 1;
        const int hx = v % HX, tt = v / HX;
        const int hy = tt % HY, hz = tt / HY;
        const bool have = p < NPIECE;
        rel[i] = have ? ((hz * H + hy) * W + hx) * 8 + half * 4 : 0;
        pos[i] = have ? (unsigned)(hz | (hy << 8) | (hx << 16)) : 0x00FFFFFFu;
        lo[i] = have ? v * 8 + ((half ^ ((hx >> 3) & 1)) * 4) : -1;
    }
    float4 stg[PPT];
    auto load_tile = [&](int t) {   // global -> registers (zeros outside the volume)
        const int bx = t % nbx, by = (t / nbx) % nby, bz = t / (nbx * nby);
        x0 = bx * TX; y0 = by * TY; z0 = bz * TZ;
        // halo origin (z0-1, y0-1, x0-1) as a (possibly negative) element offset: wave-uniform
        const long long base = ((((long long)(z0 - 1) * H + (y0 - 1)) * W) + (x0 - 1)) * 8;
        // valid halo index ranges of this tile (scalar): hz in [zl, zh] etc.
        const int zl = max(0, 1 - z0), zh = min(HZ - 1, D - z0), yl = max(0, 1 - y0), yh = min(HY - 1, H - y0),
                  xl = max(0, 1 - x0), xh = min(HX - 1, W - x0);
        const bool interior = zl == 0 && zh == HZ - 1 && yl == 0 && yh == HY - 1 && xl == 0 && xh == HX - 1;
        if (interior) {            // wave-uniform: no masks at all
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
#if MVS_ABLATE == 6  // diagnostic: no staging loads
                stg[i] = make_float4(1.f, 2.f, 3.f, 4.f);
#else
                const f32x4 val = St<DT>::load4(x, (size_t)(base + (pos[i] != 0x00FFFFFFu ? rel[i] : (int)(-base))));
                stg[i] = make_float4(val[0], val[1], val[2], val[3]);
#endif
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int hz = pos[i] & 255, hy = (pos[i] >> 8) & 255, hx = pos[i] >> 16;
                const bool ok = hz >= zl && hz <= zh && hy >= yl && hy <= yh && hx >= xl && hx <= xh;
#if MVS_ABLATE == 6
                stg[i] = make_float4(1.f, 2.f, 3.f, 4.f);
#else
                const f32x4 val = St<DT>::load4(x, ok ? (size_t)(base + rel[i]) : (size_t)0);
                stg[i] = ok ? make_float4(val[0], val[1], val[2], val[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
            }
        }
    };
    auto store_tile = [&]() {       // registers -> LDS
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            if (lo[i] >= 0) *reinterpret_cast<float4*>(tile + lo[i]) = stg[i];
    };

    int t = blockIdx.x;
    if (t >= ntiles) return;
    load_tile(t);
    store_tile();
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;  // thread -> (y, x) of the tile, 4 z outputs
    const float bv = bias[0];
#if MVS_ABLATE == 5  // diagnostic: one tap only
#define MVS_PROB_KH 1
#else
#define MVS_PROB_KH 3
#endif
    for (;;) {
        const int cx0 = x0, cy0 = y0, cz0 = z0;      // origin of the tile now in LDS
        const int tn = t + gridDim.x;
        const bool has_next = tn < ntiles;
        if (has_next) load_tile(tn);                  // in flight during the FMAs below
        // (even, odd channel) partial sums: the FMAs are v_pk_fma_f32 with the weights as SGPR pairs -- half the
        // vector instructions of a scalar fmaf chain (this kernel runs two waves per SIMD and is bound by them)
        f32x2p acc[TZ];
#pragma unroll
        for (int j = 0; j < TZ; ++j) acc[j] = (f32x2p){bv, 0.0f};
#pragma unroll 1
        for (int kh = 0; kh < MVS_PROB_KH; ++kh)
#pragma unroll 1
            for (int kw = 0; kw < MVS_PROB_KH; ++kw) {
                const int hx = tx + kw;
                const int sw = ((hx >> 3) & 1) * 4;
                const float* vp = tile + ((ty + kh) * HX + hx) * 8;
                float4 a[HZ], bq[HZ];
#pragma unroll
                for (int c = 0; c < HZ; ++c) {
                    a[c] = *reinterpret_cast<const float4*>(vp + c * HY * HX * 8 + sw);         // channels 0..3
                    bq[c] = *reinterpret_cast<const float4*>(vp + c * HY * HX * 8 + (4 - sw));  // channels 4..7
                }
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    // uniform address -> scalar (SMEM) loads, the weights are SGPR operands of the FMAs;
                    // measured 5 % faster than broadcast reads of an LDS copy (one LDS read less per 4 FMAs)
                    const float* wv = wgt + ((kd * 3 + kh) * 3 + kw) * 8;
                    const f32x2p w01 = {wv[0], wv[1]}, w23 = {wv[2], wv[3]}, w45 = {wv[4], wv[5]}, w67 = {wv[6], wv[7]};
#pragma unroll
                    for (int j = 0; j < TZ; ++j) {
                        const int c = j + kd;  // input plane of output j through tap kd
                        acc[j] = __builtin_elementwise_fma((f32x2p){a[c].x, a[c].y}, w01, acc[j]);
                        acc[j] = __builtin_elementwise_fma((f32x2p){a[c].z, a[c].w}, w23, acc[j]);
                        acc[j] = __builtin_elementwise_fma((f32x2p){bq[c].x, bq[c].y}, w45, acc[j]);
                        acc[j] = __builtin_elementwise_fma((f32x2p){bq[c].z, bq[c].w}, w67, acc[j]);
                    }
                }
            }
        const int gy = cy0 + ty, gx = cx0 + tx;
        if (gy < H && gx < W) {
#pragma unroll
            for (int j = 0; j < TZ; ++j)
                if (cz0 + j < D) y[((size_t)(cz0 + j) * H + gy) * W + gx] = acc[j].x + acc[j].y;
        }
        if (!has_next) break;
        __syncthreads();   // every thread is done reading the current tile
        store_tile();      // the next tile (requested before the FMAs)
        __syncthreads();
        t = tn;
    }
}

static int run_prob(const void* x, void* y, const float* wgt, const float* bias, int D, int H, int W,
                    int dtype, hipStream_t s) {
    using namespace pl;
    const int ntiles = ((W + TX - 1) / TX) * ((H + TY - 1) / TY) * ((D + TZ - 1) / TZ);
    // persistent grid: two blocks per CU (65 KB of LDS each), next tile requested under this tile's FMAs
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int nb = ntiles < 2 * cus ? ntiles : 2 * cus;
    float* yo = static_cast<float*>(y);
    if (dtype == MVS_F32) prob_lds_kernel<MVS_F32><<<nb, 256, 0, s>>>(x, wgt, bias, yo, D, H, W);
    else if (dtype == MVS_F16) prob_lds_kernel<MVS_F16><<<nb, 256, 0, s>>>(x, wgt, bias, yo, D, H, W);
    else prob_lds_kernel<MVS_BF16><<<nb, 256, 0, s>>>(x, wgt, bias, yo, D, H, W);
    return check_hip(hipGetLastError(), "prob_lds launch");
}

template <int CIN, int COUT, int CPT, int STRIDE, bool DECONV, bool RELU, bool SKIP>
static int run_direct(const void* x, const void* skip, void* y, const float* wgt, const float* bias,
                      int Di, int Hi, int Wi, hipStream_t s) {
    int Do, Ho, Wo;
    if (DECONV) { Do = 2 * Di; Ho = 2 * Hi; Wo = 2 * Wi; }
    else { Do = (Di - 1) / STRIDE + 1; Ho = (Hi - 1) / STRIDE + 1; Wo = (Wi - 1) / STRIDE + 1; }
    const size_t nthreads = DECONV ? (size_t)Di * Hi * Wi : (size_t)Do * Ho * Wo;
    dim3 grid((unsigned)((nthreads + 255) / 256), COUT / CPT, DECONV ? 8 : 1);
    conv3d_direct_kernel<CIN, COUT, CPT, STRIDE, DECONV, RELU, SKIP><<<grid, 256, 0, s>>>(
        static_cast<const float*>(x), wgt, bias, static_cast<const float*>(skip),
        static_cast<float*>(y), Di, Hi, Wi, Do, Ho, Wo);
    return check_hip(hipGetLastError(), "conv3d_direct launch");
}

int launch_conv_layer_direct(int layer, const void* x, const void* skip, void* y, const float* wgt,
                             const float* bias, int Di, int Hi, int Wi, int dtype, hipStream_t s) {
    if (dtype != MVS_F32) return fail(MVS_ERR_BAD_DTYPE, "conv3d: dtype %d not implemented", dtype);
    switch (layer) {
        case 0: return run_direct<32, 8, 8, 1, false, true, false>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 1: return run_direct<8, 16, 16, 2, false, true, false>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 2: return run_direct<16, 16, 16, 1, false, true, false>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 3: return run_direct<16, 32, 16, 2, false, true, false>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 4: return run_direct<32, 32, 8, 1, false, true, false>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 5: return run_direct<32, 64, 4, 2, false, true, false>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 6: return run_direct<64, 64, 4, 1, false, true, false>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 7: return run_direct<64, 32, 8, 2, true, true, true>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 8: return run_direct<32, 16, 16, 2, true, true, true>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 9: return run_deconv<16, 8, 8>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 10:
            return run_prob(x, y, wgt, bias, Di, Hi, Wi, MVS_F32, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "unknown CostRegNet layer %d", layer);
    }
}

// MVS_MFMA16=0: fp32 MFMA arithmetic on the 16-bit stored operands instead of the 16-bit matrix cores
bool mfma16_enabled() {
    static const bool v = [] {
        const char* e = getenv("MVS_MFMA16");
        return !(e && e[0] == '0');
    }();
    return v;
}

// MVS_FORCE_DIRECT=1 routes every layer through the direct kernels (A/B checks in tests).
static bool force_direct() {
    static const bool v = [] {
        const char* e = getenv("MVS_FORCE_DIRECT");
        return e && e[0] == '1';
    }();
    return v;
}

int launch_conv_layer(int layer, const void* x, const void* skip, void* y, const float* blob,
                      int Di, int Hi, int Wi, int dtype, hipStream_t s) {
    const BlobLayout L = blob_layout();
    if (force_direct()) {
        if (dtype != MVS_F32)
            return fail(MVS_ERR_BAD_DTYPE, "MVS_FORCE_DIRECT kernels are fp32-storage only (dtype %d)", dtype);
        return launch_conv_layer_direct(layer, x, skip, y, blob + L.w_off[layer], blob + L.b_off[layer],
                                        Di, Hi, Wi, dtype, s);
    }
    // 16-bit storage: 16-bit MFMA arithmetic (conv3d_mfma16.hip) unless MVS_MFMA16=0 asks for fp32
    // arithmetic on the narrowed operands
    if (mfma16_enabled() && (dtype == MVS_F16 || dtype == MVS_BF16) && layer <= 9)
        return launch_layer_mfma16(layer, x, skip, y, blob + L.h16_off[dtype == MVS_F16 ? 0 : 1][layer],
                                   blob + L.b_off[layer], Di, Hi, Wi, dtype, s);
    if (layer == 0) {
        // conv0: Winograd F(4,3) along z on the 4x4x1 MFMA (1/2 of the direct form's MFMAs; conv_winograd.hip)
        // unless MVS_CONV0_WINO=0 asks for the direct form (cross-check) or the volume needs 64-bit offsets
        static const bool wino = [] {
            const char* e = getenv("MVS_CONV0_WINO");
            return !(e && e[0] == '0');
        }();
        // fp32 volumes: the same F(4,3) scheme with every operand split into three bf16 pieces, six cross products per
        // fp32 product on the bf16 matrix cores, fp32 accumulation (conv0_split.hip: fp32-equivalent arithmetic -- the
        // per-layer bounds of the fp32 kernel hold unchanged -- at 0.28 instead of 0.36 ms).  MVS_CONV0_SPLIT=0: the
        // fp32-MFMA kernel below; =2: the split kernel's first form (one tile per block)
        static const bool split = [] {
            const char* e = getenv("MVS_CONV0_SPLIT");
            return !(e && e[0] == '0');
        }();
        if (wino && Di % 4 == 0 && (size_t)Di * Hi * Wi * 32 < ((size_t)1 << 31)) {
            if (split && dtype == MVS_F32)
                return launch_conv0_wino43_split(x, y, blob + L.c0w43s_off, blob + L.b_off[0], Di, Hi, Wi, dtype, s);
            return launch_conv0_wino43(x, y, blob + L.c0w43_off, blob + L.b_off[0], Di, Hi, Wi, dtype, s);
        }
        return launch_conv0_mfma(x, y, blob + L.c0q_off, blob + L.b_off[0], Di, Hi, Wi, dtype, s);
    }
    // fp32 volumes: the level-1 .. 3 layers on the bf16 matrix cores with split operands (conv3d_mfma16.hip, round 4)
    // unless MVS_SPLIT_LAYERS=0 keeps the fp32-MFMA kernels below
    static const bool split_layers = [] {
        const char* e = getenv("MVS_SPLIT_LAYERS");
        return !(e && e[0] == '0');
    }();
    if (split_layers && dtype == MVS_F32 && split_layer_covers(layer))
        return launch_layer_split(layer, x, skip, y, blob + L.s16_off[layer], blob + L.b_off[layer], Di, Hi, Wi, s);
    if (layer == 2 || layer == 4) {
        // stride-1 layers conv2 / conv4: Winograd F(2,3) along z (conv_winograd.hip) unless
        // MVS_CONV_WINO=0.  conv6 (64 -> 64 on 7,680 voxels) stays direct: with two-plane tiles it has
        // only 240 blocks for 256 CUs and measured 0.035 vs 0.033 ms.
        static const bool wz = [] {
            const char* e = getenv("MVS_CONV_WINO");
            return !(e && e[0] == '0');
        }();
        if (wz) return launch_convwz_mfma(layer, x, y, blob + L.wz_off[layer], blob + L.b_off[layer], Di, Hi, Wi, dtype, s);
    }
    // the 1/8-resolution layers conv5 / conv6 / conv7: all-K-resident split-K kernels (conv3d_small.hip)
    if (convs_covers(layer)) {
        if (kLayers[layer].kind == kConv)
            return launch_convs_mfma(layer, x, y, blob + L.gp_off[layer], blob + L.b_off[layer], Di, Hi, Wi, dtype, s);
        return launch_deconvs_mfma(layer, x, skip, y, blob + L.gp_off[layer], blob + L.b_off[layer], Di, Hi, Wi, dtype, s);
    }
    if (layer >= 1 && layer <= 6)
        return launch_convg_mfma(layer, x, y, blob + L.gp_off[layer], blob + L.b_off[layer], Di, Hi, Wi,
                                 dtype, s);
    if (layer >= 7 && layer <= 9)
        return launch_deconvg_mfma(layer, x, skip, y, blob + L.gp_off[layer], blob + L.b_off[layer], Di,
                                   Hi, Wi, dtype, s);
    if (layer == 10)
        return run_prob(x, y, blob + L.w_off[10], blob + L.b_off[10], Di, Hi, Wi, dtype, s);
    return fail(MVS_ERR_BAD_SHAPE, "unknown CostRegNet layer %d", layer);
}

}  // namespace mvs
