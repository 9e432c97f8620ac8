// mvs_internal.h -- shared host-side definitions of libmvs_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "mvs_abi.h"

namespace mvs {

constexpr int kC = 32;  // feature channels (FeatureNet output, models/mvsnet.py:24)

// CostRegNet layer table (models/mvsnet.py:35-62).
enum LayerKind { kConv = 0, kDeconv = 1 };
struct LayerSpec {
    int cin, cout, stride, kind;
    int level_in, level_out;  // resolution level: voxel dims = (D,h,w) >> level
};
constexpr LayerSpec kLayers[MVS_NUM_LAYERS] = {
    {32, 8, 1, kConv, 0, 0},     // 0 conv0
    {8, 16, 2, kConv, 0, 1},     // 1 conv1
    {16, 16, 1, kConv, 1, 1},    // 2 conv2
    {16, 32, 2, kConv, 1, 2},    // 3 conv3
    {32, 32, 1, kConv, 2, 2},    // 4 conv4
    {32, 64, 2, kConv, 2, 3},    // 5 conv5
    {64, 64, 1, kConv, 3, 3},    // 6 conv6
    {64, 32, 2, kDeconv, 3, 2},  // 7 conv7  (+conv4)
    {32, 16, 2, kDeconv, 2, 1},  // 8 conv9  (+conv2)
    {16, 8, 2, kDeconv, 1, 0},   // 9 conv11 (+conv0)
    {8, 1, 1, kConv, 0, 0},      // 10 prob (bias, no BN / ReLU)
};

// Packed weight blob: for every layer l, fp32 weights [27 taps][cin][cout] with BN folded in,
// followed by fp32 bias[cout]; each section starts 256-byte aligned.  Tap index = kd*9+kh*3+kw
// of the reference kernel (for deconv layers: the k of  o = 2*i - 1 + k).
struct BlobLayout {
    size_t w_off[MVS_NUM_LAYERS];  // in floats
    size_t b_off[MVS_NUM_LAYERS];  // in floats
    size_t c0q_off;                // conv0 4x4x1 panel [4][27][2][2][4][4] (conv3d_mfma.hip)
    size_t gp_off[MVS_NUM_LAYERS]; // generic MFMA panels of layers 1..6 and deconv panels of 7..9
    size_t h16_off[2][10];         // 16-bit MFMA panels of layers 0..9 for MVS_F16 ([0]) / MVS_BF16 ([1])
    size_t wz_off[MVS_NUM_LAYERS]; // Winograd-z panels of the stride-1 layers 2 and 4 (conv_winograd.hip)
    size_t c0w43_off;              // conv0 Winograd F(4,3)-z panel [4][6][9][2][2][4][4] (conv_winograd.hip)
    size_t c0w43s_off;             // the same weights as three bf16 pieces, Toeplitz panel [4][6][3][3][64][8] (conv0_split.hip)
    size_t s16_off[10];            // split-operand panels of layers 2..4, 7, 8 (conv3d_mfma16.hip) and 9 (conv11_prob.hip): three bf16 pieces in the h16 layout
    size_t total_floats;
};
inline BlobLayout blob_layout() {
    BlobLayout L{};
    size_t off = 0;
    for (int l = 0; l < MVS_NUM_LAYERS; ++l) {
        L.w_off[l] = off;
        off += (size_t)27 * kLayers[l].cin * kLayers[l].cout;
        off = (off + 63) & ~(size_t)63;
        L.b_off[l] = off;
        off += (size_t)kLayers[l].cout;
        off = (off + 63) & ~(size_t)63;
    }
    L.c0q_off = off;
    off += (size_t)4 * 27 * 2 * 2 * 4 * 4;
    for (int l = 1; l <= 6; ++l) {
        L.gp_off[l] = off;
        off += (size_t)(kLayers[l].cin / 8) * (kLayers[l].cout / 16) * 14 * 64 * 4;
    }
    for (int l = 7; l <= 9; ++l) {
        L.gp_off[l] = off;
        off += (size_t)(kLayers[l].cin / 8) * (2 * kLayers[l].cout / 16) * 9 * 64 * 4;
    }
    for (int d = 0; d < 2; ++d)
        for (int l = 0; l < 10; ++l) {
            const size_t elems = l == 0 ? (size_t)4 * 9 * 64 * 8
                               : l <= 6 ? (size_t)(kLayers[l].cin / 8) * (kLayers[l].cout / 16) * 7 * 64 * 8
                                        : (size_t)(kLayers[l].cin / 8) * (2 * kLayers[l].cout / 16) * 5 * 64 * 8;
            L.h16_off[d][l] = off;
            off += (elems / 2 + 63) & ~(size_t)63;
        }
    for (int l = 2; l <= 4; l += 2) {
        L.wz_off[l] = off;
        off += (size_t)(kLayers[l].cin / 8) * 4 * (kLayers[l].cout / 16) * 5 * 64 * 4;
    }
    L.c0w43_off = off;
    off += (size_t)4 * 6 * 9 * 2 * 2 * 4 * 4;
    L.c0w43s_off = off;
    off += (size_t)4 * 6 * 3 * 3 * 64 * 8 / 2;
    for (int l = 2; l <= 4; ++l) {
        const size_t elems = (size_t)(kLayers[l].cin / 8) * (kLayers[l].cout / 16) * 7 * 64 * 8;
        L.s16_off[l] = off;
        off += (3 * elems / 2 + 63) & ~(size_t)63;
    }
    for (int l = 9; l >= 7; --l) {   // conv11 first: the offsets of round 4's earlier builds stay where they were
        const size_t elems = (size_t)(kLayers[l].cin / 8) * (2 * kLayers[l].cout / 16) * 5 * 64 * 8;
        L.s16_off[l] = off;
        off += (3 * elems / 2 + 63) & ~(size_t)63;
    }
    L.total_floats = off;
    return L;
}

inline size_t dtype_size(int dtype) { return dtype == MVS_F32 ? 4 : 2; }

// Device workspace carve-up (byte offsets, 256-byte aligned).
struct Workspace {
    size_t feats_t;  // [4][N][h][w][8] fp32 C8-planar copy of the features
    size_t rt;       // [(N-1)][12] fp32
    size_t var;      // [4][D][h][w][8]
    size_t act[10];  // outputs of layers 0..9 (C8-planar, storage dtype)
    size_t cost;     // [D][h][w] fp32
    size_t total;
};
inline Workspace workspace_layout(int N, int C, int D, int h, int w, int dtype) {
    Workspace W{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t es = dtype_size(dtype);
    W.feats_t = take((size_t)N * C * h * w * 4);
    W.rt = take((size_t)(N > 1 ? N - 1 : 1) * 12 * 4);
    W.var = take((size_t)D * h * w * kC * es);
    for (int l = 0; l < 10; ++l) {
        const int lv = kLayers[l].level_out;
        W.act[l] = take((size_t)(D >> lv) * (h >> lv) * (w >> lv) * kLayers[l].cout * es);
    }
    W.cost = take((size_t)D * h * w * 4);
    W.total = off;
    return W;
}

// ---- FeatureNet (models/mvsnet.py:15-24): conv0..conv6 (ConvBnReLU) + feature (Conv2d + bias) ----
struct FeatLayerSpec {
    int cin, cout, k, stride, relu;
};
constexpr FeatLayerSpec kFeatLayers[MVS_FEATURE_LAYERS] = {
    {3, 8, 3, 1, 1},   {8, 8, 3, 1, 1},   {8, 16, 5, 2, 1},  {16, 16, 3, 1, 1},
    {16, 16, 3, 1, 1}, {16, 32, 5, 2, 1}, {32, 32, 3, 1, 1}, {32, 32, 3, 1, 0}};
// feature blob: per layer the MFMA B panel [cin/8][cout/16][(k*k+1)/2][64][4] then 16*NT bias floats
struct FeatBlob {
    size_t panel_off[MVS_FEATURE_LAYERS], bias_off[MVS_FEATURE_LAYERS];
    size_t l0_direct_off;  // conv0 for the fused conv0+conv1 kernel: [27][8] weights (k = (ci*3+ky)*3+kx), bias [8]
    size_t total_floats;
};
inline FeatBlob feat_blob_layout() {
    FeatBlob L{};
    size_t off = 0;
    for (int l = 0; l < MVS_FEATURE_LAYERS; ++l) {
        const FeatLayerSpec& S = kFeatLayers[l];
        const size_t nch = (S.cin + 7) / 8, nt = (S.cout + 15) / 16, ks = (S.k * S.k + 1) / 2;
        L.panel_off[l] = off;
        off += nch * nt * ks * 64 * 4;
        L.bias_off[l] = off;
        off += (nt * 16 + 63) / 64 * 64;
    }
    L.l0_direct_off = off;
    off += 256;
    L.total_floats = off;
    return L;
}
// FeatureNet scratch: two ping-pong activation buffers (largest activation = 8 channels at full
// resolution) and a C8-planar output [4][N][H/4][W/4][8]
struct FeatWorkspace {
    size_t bufA, bufB, c8, total;
    int h4, w4;
};
inline FeatWorkspace feat_workspace_layout(int N, int H, int W) {
    FeatWorkspace F{};
    const int H2 = (H - 1) / 2 + 1, W2 = (W - 1) / 2 + 1;
    F.h4 = (H2 - 1) / 2 + 1;
    F.w4 = (W2 - 1) / 2 + 1;
    const size_t full = ((size_t)N * H * W * 8 * 4 + 255) & ~(size_t)255;
    const size_t half = ((size_t)N * H2 * W2 * 16 * 4 + 255) & ~(size_t)255;
    const size_t big = full > half ? full : half;
    F.bufA = 0;
    F.bufB = big;
    F.c8 = 2 * big;
    F.total = 2 * big + (((size_t)N * F.h4 * F.w4 * 32 * 4 + 255) & ~(size_t)255);
    return F;
}
bool feat16_gather();
int launch_feature_layer(int l, const float* x, float* y, const float* blob, int N, int Hi, int Wi,
                         hipStream_t s);
int launch_feature_net_c8(const void* imgs, int fmt, const float* blob, float* feats_c8, float* bufA, float* bufB,
                          int N, int H, int W, hipStream_t s);   // fmt: mvs_image_format
int launch_c8_to_nchw(const float* in, float* out, int N, int C, int h, int w, hipStream_t s);
int launch_narrow_features(const float* in, void* out, size_t n, int dtype, hipStream_t s);
void pack_fconv_weights(const float* w, int cin, int cout, int k, float* bp);

// Transposed-convolution k-steps of the gather-form MFMA kernels (conv3d_mfma.hip, conv11_prob.hip):
// k-step -> (output parity class 2*pz + py, kz, dz, ky, dy).  Per axis o = 2i - 1 + k: parity 0 has the
// one tap (k=1, d=0), parity 1 the two taps (k=2, d=0) and (k=0, d=1).
struct DeconvStep { int cls, kz, dz, ky, dy; };
__host__ __device__ constexpr DeconvStep deconv_step(int ks) {
    // ks 0: cls0 | 1,2: cls1 (py=1) | 3,4: cls2 (pz=1) | 5..8: cls3 (pz=1, py=1)
    return ks == 0 ? DeconvStep{0, 1, 0, 1, 0}
         : ks == 1 ? DeconvStep{1, 1, 0, 2, 0}
         : ks == 2 ? DeconvStep{1, 1, 0, 0, 1}
         : ks == 3 ? DeconvStep{2, 2, 0, 1, 0}
         : ks == 4 ? DeconvStep{2, 0, 1, 1, 0}
         : ks == 5 ? DeconvStep{3, 2, 0, 2, 0}
         : ks == 6 ? DeconvStep{3, 2, 0, 0, 1}
         : ks == 7 ? DeconvStep{3, 0, 1, 2, 0}
                   : DeconvStep{3, 0, 1, 0, 1};
}

// 16-bit (16x16x32 MFMA) form of the same transposed convolution (conv3d_mfma16.hip, conv11_prob.hip): a lane's A
// fragment is the 8 channels of ONE voxel, lane group g -> dx = g & 1 and (z, y)-tap combo q = g >> 1
struct Deconv16Tap { int cls, kz, dz, ky, dy, valid; };
// k-step ks, combo q (= g>>1): 5 k-steps x 2 (z,y)-tap combos, each with dx = g&1
__host__ __device__ constexpr Deconv16Tap deconv16_tap(int ks, int q) {
    return ks == 0 ? (q == 0 ? Deconv16Tap{0, 1, 0, 1, 0, 1} : Deconv16Tap{0, 1, 0, 1, 0, 0})
         : ks == 1 ? (q == 0 ? Deconv16Tap{1, 1, 0, 2, 0, 1} : Deconv16Tap{1, 1, 0, 0, 1, 1})
         : ks == 2 ? (q == 0 ? Deconv16Tap{2, 2, 0, 1, 0, 1} : Deconv16Tap{2, 0, 1, 1, 0, 1})
         : ks == 3 ? (q == 0 ? Deconv16Tap{3, 2, 0, 2, 0, 1} : Deconv16Tap{3, 2, 0, 0, 1, 1})
                   : (q == 0 ? Deconv16Tap{3, 0, 1, 2, 0, 1} : Deconv16Tap{3, 0, 1, 0, 1, 1});
}

// thread-local error text
int fail(int code, const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// kernel launchers (implemented in the .hip files); all enqueue on `s` and return a status
// proj/rt != NULL: the relative projections are computed by an extra block row of the same launch
int launch_nchw_to_c8(const float* in, void* out, int N, int C, int h, int w, int dtype, hipStream_t s,
                      const float* proj = nullptr, float* rt = nullptr);
int launch_warp_variance16(const void* feats16, const float* rt, const float* dv, void* var, int N, int D,
                           int h, int w, int dtype, hipStream_t s);
int launch_relative_proj(const float* proj, float* rt, int N, hipStream_t s);
int launch_warp_variance(const float* feats_t, const float* rt, const float* dv, void* var, int N,
                         int D, int h, int w, int dtype, hipStream_t s);
bool warp_tc_fits(int N, int D, int h, int w, int fes, int ves);   // the tap-cache kernel's range
int launch_warp_variance_tc(const float* feats_p, const float* rt, const float* dv, void* var, int N, int D,
                            int h, int w, int dtype, hipStream_t s);
int launch_warp_variance_tc16(const void* feats16, const float* rt, const float* dv, void* var, int N, int D,
                              int h, int w, int dtype, hipStream_t s);
int launch_homo_warp(const float* fea, const float* rt, const float* dv, float* out, int C, int D,
                     int h, int w, hipStream_t s);
int launch_conv_layer(int layer, const void* x, const void* skip, void* y, const float* blob,
                      int Di, int Hi, int Wi, int dtype, hipStream_t s);
int launch_conv_layer_direct(int layer, const void* x, const void* skip, void* y, const float* wgt,
                             const float* bias, int Di, int Hi, int Wi, int dtype, hipStream_t s);
int launch_conv0_wino43(const void* x, void* y, const float* bw, const float* bias, int D, int H, int W,
                        int dtype, hipStream_t s);
void pack_conv0_wino43_weights(const float* wfold, float* bw);
// conv0 with split bf16 operands on the 16-bit matrix cores, fp32 volumes (conv0_split.hip)
int launch_conv0_wino43_split(const void* x, void* y, const void* bp, const float* bias, int D, int H, int W,
                              int dtype, hipStream_t s);
void pack_conv0_wino43_split_weights(const float* wfold, void* out);
int launch_convwz_mfma(int layer, const void* x, void* y, const float* bp, const float* bias, int D, int H,
                       int W, int dtype, hipStream_t s);
void pack_convwz_weights(const float* wfold, int cin, int cout, float* bp);
int launch_conv0_mfma(const void* x, void* y, const float* bq, const float* bias, int D, int H, int W,
                      int dtype, hipStream_t s);
void pack_conv0_4x4_weights(const float* wfold, float* bq);
int launch_convg_mfma(int layer, const void* x, void* y, const float* bp, const float* bias, int Di,
                      int Hi, int Wi, int dtype, hipStream_t s);
void pack_convg_weights(const float* wfold, int cin, int cout, float* bp);
int launch_deconvg_mfma(int layer, const void* x, const void* skip, void* y, const float* bp,
                        const float* bias, int Di, int Hi, int Wi, int dtype, hipStream_t s);
void pack_deconvg_weights(const float* wfold, int cin, int cout, float* bp);
// all-K-resident split-K kernels for the small levels (conv3d_small.hip); same panels as convg / deconvg
bool convs_covers(int layer);
int launch_convs_mfma(int layer, const void* x, void* y, const float* bp, const float* bias, int Di, int Hi, int Wi,
                      int dtype, hipStream_t s);
int launch_deconvs_mfma(int layer, const void* x, const void* skip, void* y, const float* bp, const float* bias, int Di,
                        int Hi, int Wi, int dtype, hipStream_t s);
// conv11 (+ conv0 skip) and prob in one kernel (conv11_prob.hip): fp32 storage, or 16-bit storage on the 16-bit MFMA
int launch_conv11_prob(const void* x, const void* skip, float* cost, const float* blob, int Di, int Hi,
                       int Wi, int dtype, hipStream_t s);
bool conv11_prob_enabled(int dtype);
bool mfma16_enabled();   // MVS_MFMA16 != 0
int launch_layer_mfma16(int layer, const void* x, const void* skip, void* y, const void* panel,
                        const float* bias, int Di, int Hi, int Wi, int dtype, hipStream_t s);
void pack_mfma16_panel(int layer, const float* wfold, int dt, void* out);
// fp32 volumes with split bf16 operands (three pieces, six cross products): the tile kernels of conv3d_mfma16.hip
bool split_layer_covers(int layer);
int launch_layer_split(int layer, const void* x, const void* skip, void* y, const void* panel, const float* bias,
                       int Di, int Hi, int Wi, hipStream_t s);
void pack_split_panels(int layer, const float* wfold, void* out);
int launch_softargmin(const float* cost, const float* dv, float* depth, float* conf, int D, int h,
                      int w, hipStream_t s);
int launch_depth_regression(const float* p, const float* dv, float* depth, int D, int h, int w,
                            hipStream_t s);

}  // namespace mvs
