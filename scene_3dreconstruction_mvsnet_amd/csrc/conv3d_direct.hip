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
        case 9: return run_direct<16, 8, 8, 2, true, true, true>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        case 10: return run_direct<8, 1, 1, 1, false, false, false>(x, skip, y, wgt, bias, Di, Hi, Wi, s);
        default: return fail(MVS_ERR_BAD_SHAPE, "unknown CostRegNet layer %d", layer);
    }
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
    if (dtype != MVS_F32) return fail(MVS_ERR_BAD_DTYPE, "conv3d: dtype %d not implemented", dtype);
    const BlobLayout L = blob_layout();
    if (layer == 0 && !force_direct())
        return launch_conv0_mfma(x, y, blob + L.c0p_off, blob + L.b_off[0], Di, Hi, Wi, s);
    return launch_conv_layer_direct(layer, x, skip, y, blob + L.w_off[layer], blob + L.b_off[layer],
                                    Di, Hi, Wi, dtype, s);
}

}  // namespace mvs
