// mvs_host.hip -- C-ABI entry points of libmvs_hip.so (declared in include/mvs_abi.h).
// Host-side orchestration only: argument validation, workspace carve-up, weight packing and
// the per-layer launch sequence of CostRegNet (reference models/mvsnet.py:64-73).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mvs_internal.h"

namespace mvs {

static thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return MVS_OK;
    return fail(MVS_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

static int check_dims(int N, int C, int D, int h, int w, int dtype) {
    if (dtype != MVS_F32 && dtype != MVS_F16 && dtype != MVS_BF16)
        return fail(MVS_ERR_BAD_DTYPE, "unknown dtype %d", dtype);
    if (N < 1 || N > 64) return fail(MVS_ERR_BAD_SHAPE, "number of views N=%d outside [1,64]", N);
    if (C != kC) return fail(MVS_ERR_BAD_SHAPE, "feature channels C=%d, expected %d", C, kC);
    if (D < 8 || h < 8 || w < 8 || (D % 8) || (h % 8) || (w % 8))
        return fail(MVS_ERR_BAD_SHAPE,
                    "D,h,w = %d,%d,%d must be positive multiples of 8 (three stride-2 stages with "
                    "additive skips, models/mvsnet.py:38-60,69-71)", D, h, w);
    if ((size_t)D * h * w * kC >= ((size_t)1 << 32))
        return fail(MVS_ERR_BAD_SHAPE, "volume of %zu elements exceeds 32-bit indexing",
                    (size_t)D * h * w * kC);
    return MVS_OK;
}

// 16-bit modes: the warp gathers from the fp32 feature copy (round 4: the kernel is bound by vector-instruction issue,
// and widening 16-bit taps on every re-gather costs more than the halved gather bytes save -- cfg3 bf16 1.379 -> 1.347
// ms, cfg5 fp16 0.1226 -> 0.1131; the features then are not rounded at all).  MVS_FEAT16=1: the narrowed copy (the
// default of rounds 2-3)
bool feat16_gather() {
    static const bool on = [] {
        const char* e = getenv("MVS_FEAT16");
        return e && e[0] == '1';
    }();
    return on;
}

}  // namespace mvs

using namespace mvs;

extern "C" {

int mvs_abi_version(void) { return MVS_ABI_VERSION; }

const char* mvs_last_error_string(void) { return g_err.c_str(); }

int mvs_query_workspace(int N, int C, int D, int h, int w, int dtype, size_t* bytes) {
    if (!bytes) return fail(MVS_ERR_NULL, "bytes is NULL");
    if (int st = check_dims(N, C, D, h, w, dtype)) return st;
    *bytes = workspace_layout(N, C, D, h, w, dtype).total;
    return MVS_OK;
}

int mvs_query_weights_blob(size_t* bytes) {
    if (!bytes) return fail(MVS_ERR_NULL, "bytes is NULL");
    *bytes = blob_layout().total_floats * sizeof(float);
    return MVS_OK;
}

int mvs_pack_weights(const float* const* conv_weights, const float* const* bn_params,
                     const float* prob_bias, float eps, void* blob_out, size_t blob_bytes) {
    if (!conv_weights || !bn_params || !prob_bias || !blob_out)
        return fail(MVS_ERR_NULL, "mvs_pack_weights: NULL argument");
    const BlobLayout L = blob_layout();
    if (blob_bytes < L.total_floats * sizeof(float))
        return fail(MVS_ERR_WORKSPACE, "weight blob needs %zu bytes, got %zu",
                    L.total_floats * sizeof(float), blob_bytes);
    float* blob = static_cast<float*>(blob_out);
    std::memset(blob, 0, L.total_floats * sizeof(float));
    for (int l = 0; l < MVS_NUM_LAYERS; ++l) {
        const LayerSpec& S = kLayers[l];
        const float* wsrc = conv_weights[l];
        if (!wsrc) return fail(MVS_ERR_NULL, "conv_weights[%d] is NULL", l);
        float* wdst = blob + L.w_off[l];
        float* bdst = blob + L.b_off[l];
        for (int co = 0; co < S.cout; ++co) {
            float scale = 1.0f, shift = 0.0f;
            if (l < 10) {
                const float* g = bn_params[4 * l + 0];
                const float* b = bn_params[4 * l + 1];
                const float* m = bn_params[4 * l + 2];
                const float* v = bn_params[4 * l + 3];
                if (!g || !b || !m || !v) return fail(MVS_ERR_NULL, "bn_params of layer %d NULL", l);
                // y = (x - mean) / sqrt(var + eps) * gamma + beta   (models/module.py:29-33)
                scale = g[co] / std::sqrt(v[co] + eps);
                shift = b[co] - m[co] * scale;
            } else {
                shift = prob_bias[0];
            }
            bdst[co] = shift;
            for (int ci = 0; ci < S.cin; ++ci)
                for (int t = 0; t < 27; ++t) {
                    // Conv3d: [Cout][Cin][27]; ConvTranspose3d: [Cin][Cout][27]
                    const size_t src = (S.kind == kConv)
                                           ? ((size_t)co * S.cin + ci) * 27 + t
                                           : ((size_t)ci * S.cout + co) * 27 + t;
                    wdst[((size_t)t * S.cin + ci) * S.cout + co] = wsrc[src] * scale;
                }
        }
    }
    pack_conv0_4x4_weights(blob + L.w_off[0], blob + L.c0q_off);
    pack_conv0_wino43_weights(blob + L.w_off[0], blob + L.c0w43_off);
    pack_conv0_wino43_split_weights(blob + L.w_off[0], blob + L.c0w43s_off);
    for (int l = 2; l <= 4; ++l) pack_split_panels(l, blob + L.w_off[l], blob + L.s16_off[l]);
    for (int l = 7; l <= 9; ++l) pack_split_panels(l, blob + L.w_off[l], blob + L.s16_off[l]);
    for (int l = 2; l <= 4; l += 2)
        pack_convwz_weights(blob + L.w_off[l], kLayers[l].cin, kLayers[l].cout, blob + L.wz_off[l]);
    for (int l = 1; l <= 6; ++l)
        pack_convg_weights(blob + L.w_off[l], kLayers[l].cin, kLayers[l].cout, blob + L.gp_off[l]);
    for (int l = 7; l <= 9; ++l)
        pack_deconvg_weights(blob + L.w_off[l], kLayers[l].cin, kLayers[l].cout, blob + L.gp_off[l]);
    for (int d = 0; d < 2; ++d)
        for (int l = 0; l < 10; ++l)
            pack_mfma16_panel(l, blob + L.w_off[l], d == 0 ? MVS_F16 : MVS_BF16, blob + L.h16_off[d][l]);
    return MVS_OK;
}

int mvs_relative_proj(const float* proj, float* rt_out, int N, void* stream) {
    if (!proj || !rt_out) return fail(MVS_ERR_NULL, "mvs_relative_proj: NULL argument");
    if (N < 1 || N > 64) return fail(MVS_ERR_BAD_SHAPE, "N=%d outside [1,64]", N);
    if (N == 1) return MVS_OK;
    return launch_relative_proj(proj, rt_out, N, static_cast<hipStream_t>(stream));
}

// feats == NULL: the C8-planar feature copy (fp32, or the storage dtype when the 16-bit gather is
// on) already sits in the workspace -- written there by FeatureNet's last layer (featnet.hip).
// proj != NULL (and feats != NULL): `rt` is an OUTPUT, filled by an extra block of the transpose launch
static int warp_variance_impl(const float* feats, const float* rt, const float* depth_values,
                              void* var_out, void* workspace, size_t workspace_bytes, int N, int C, int D,
                              int h, int w, int dtype, void* stream, const float* proj = nullptr) {
    if (!depth_values || !var_out || !workspace || (N > 1 && !rt))
        return fail(MVS_ERR_NULL, "mvs_warp_variance: NULL argument");
    if (int st = check_dims(N, C, D, h, w, dtype)) return st;
    const Workspace W = workspace_layout(N, C, D, h, w, dtype);
    if (workspace_bytes < W.rt)  // only the transposed-feature region is needed here
        return fail(MVS_ERR_WORKSPACE, "workspace needs >= %zu bytes, got %zu", W.rt, workspace_bytes);
    if (reinterpret_cast<uintptr_t>(workspace) & 255)
        return fail(MVS_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* feats_t = reinterpret_cast<float*>(static_cast<char*>(workspace) + W.feats_t);
    if (feat16_gather() && dtype != MVS_F32) {
        if (feats)
            if (int st = launch_nchw_to_c8(feats, feats_t, N, C, h, w, dtype, s, proj, const_cast<float*>(rt))) return st;
        return launch_warp_variance16(feats_t, rt, depth_values, var_out, N, D, h, w, dtype, s);
    }
    if (feats)
        if (int st = launch_nchw_to_c8(feats, feats_t, N, C, h, w, MVS_F32, s, proj, const_cast<float*>(rt))) return st;
    return launch_warp_variance(feats_t, rt, depth_values, var_out, N, D, h, w, dtype, s);
}

int mvs_warp_variance(const float* feats, const float* rt, const float* depth_values,
                      void* var_out, void* workspace, size_t workspace_bytes, int N, int C, int D,
                      int h, int w, int dtype, void* stream) {
    if (!feats) return fail(MVS_ERR_NULL, "mvs_warp_variance: NULL argument");
    return warp_variance_impl(feats, rt, depth_values, var_out, workspace, workspace_bytes, N, C, D, h, w,
                              dtype, stream);
}

// CostRegNet from the variance volume (models/mvsnet.py:64-73)
static int costreg_impl(const void* var, const void* weights_blob, float* cost_out, void* workspace,
                        size_t workspace_bytes, int D, int h, int w, int dtype, void* stream) {
    if (!weights_blob || !cost_out || !workspace)
        return fail(MVS_ERR_NULL, "mvs_costreg_forward: NULL argument");
    if (int st = check_dims(1, kC, D, h, w, dtype)) return st;
    // The activation regions do not depend on N; use N=1 offsets relative to the act[] base so
    // that any workspace sized by mvs_query_workspace(N>=1,...) is large enough.
    const Workspace W = workspace_layout(1, kC, D, h, w, dtype);
    if (workspace_bytes < W.total)
        return fail(MVS_ERR_WORKSPACE, "workspace needs >= %zu bytes, got %zu", W.total, workspace_bytes);
    if (reinterpret_cast<uintptr_t>(workspace) & 255)
        return fail(MVS_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    const float* blob = static_cast<const float*>(weights_blob);
    auto act = [&](int l) { return static_cast<void*>(ws + W.act[l]); };
    auto run = [&](int l, const void* x, const void* skip, void* y) {
        const int lv = kLayers[l].level_in;
        return launch_conv_layer(l, x, skip, y, blob, D >> lv, h >> lv, w >> lv, dtype, s);
    };
    // models/mvsnet.py:64-73
    int st;
    if ((st = run(0, var, nullptr, act(0)))) return st;         // conv0
    if ((st = run(1, act(0), nullptr, act(1)))) return st;     // conv1 (s2)
    if ((st = run(2, act(1), nullptr, act(2)))) return st;     // conv2
    if ((st = run(3, act(2), nullptr, act(3)))) return st;     // conv3 (s2)
    if ((st = run(4, act(3), nullptr, act(4)))) return st;     // conv4
    if ((st = run(5, act(4), nullptr, act(5)))) return st;     // conv5 (s2)
    if ((st = run(6, act(5), nullptr, act(6)))) return st;     // conv6
    if ((st = run(7, act(6), act(4), act(7)))) return st;      // conv4 + conv7(x)
    if ((st = run(8, act(7), act(2), act(8)))) return st;      // conv2 + conv9(x)
    if (conv11_prob_enabled(dtype))                            // conv0 + conv11(x) and prob in one kernel
        return launch_conv11_prob(act(8), act(0), cost_out, blob, D >> 1, h >> 1, w >> 1, dtype, s);
    if ((st = run(9, act(8), act(0), act(9)))) return st;      // conv0 + conv11(x)
    return run(10, act(9), nullptr, cost_out);                 // prob
}

int mvs_costreg_forward(const void* var, const void* weights_blob, float* cost_out,
                        void* workspace, size_t workspace_bytes, int D, int h, int w, int dtype,
                        void* stream) {
    if (!var) return fail(MVS_ERR_NULL, "mvs_costreg_forward: NULL argument");
    return costreg_impl(var, weights_blob, cost_out, workspace, workspace_bytes, D, h, w, dtype, stream);
}

int mvs_conv_layer(int layer, const void* x, const void* skip, void* y, const void* weights_blob,
                   int Di, int Hi, int Wi, int dtype, void* stream) {
    if (!x || !y || !weights_blob) return fail(MVS_ERR_NULL, "mvs_conv_layer: NULL argument");
    if (layer < 0 || layer >= MVS_NUM_LAYERS) return fail(MVS_ERR_BAD_SHAPE, "layer %d outside [0,%d)", layer, MVS_NUM_LAYERS);
    if (dtype != MVS_F32 && dtype != MVS_F16 && dtype != MVS_BF16)
        return fail(MVS_ERR_BAD_DTYPE, "unknown dtype %d", dtype);
    const LayerSpec& S = kLayers[layer];
    if (S.kind == kDeconv && !skip) return fail(MVS_ERR_NULL, "layer %d needs its skip tensor", layer);
    if (Di < 1 || Hi < 1 || Wi < 1 || (S.stride == 2 && S.kind == kConv && ((Di | Hi | Wi) & 1)))
        return fail(MVS_ERR_BAD_SHAPE, "layer %d: input dims %d,%d,%d unsupported", layer, Di, Hi, Wi);
    return launch_conv_layer(layer, x, skip, y, static_cast<const float*>(weights_blob), Di, Hi, Wi,
                             dtype, static_cast<hipStream_t>(stream));
}

int mvs_conv11_prob(const void* x, const void* skip, float* cost_out, const void* weights_blob,
                    int Di, int Hi, int Wi, int dtype, void* stream) {
    if (!x || !skip || !cost_out || !weights_blob) return fail(MVS_ERR_NULL, "mvs_conv11_prob: NULL argument");
    if (dtype != MVS_F32 && dtype != MVS_F16 && dtype != MVS_BF16)
        return fail(MVS_ERR_BAD_DTYPE, "mvs_conv11_prob: unknown dtype %d", dtype);
    if (Di < 1 || Hi < 1 || Wi < 1) return fail(MVS_ERR_BAD_SHAPE, "mvs_conv11_prob: input dims %d,%d,%d", Di, Hi, Wi);
    return launch_conv11_prob(x, skip, cost_out, static_cast<const float*>(weights_blob), Di, Hi, Wi, dtype,
                              static_cast<hipStream_t>(stream));
}

int mvs_softargmin_conf(const float* cost, const float* depth_values, float* depth_out,
                        float* conf_out, int D, int h, int w, void* stream) {
    if (!cost || !depth_values || !depth_out || !conf_out)
        return fail(MVS_ERR_NULL, "mvs_softargmin_conf: NULL argument");
    if (D < 1 || h < 1 || w < 1) return fail(MVS_ERR_BAD_SHAPE, "D,h,w = %d,%d,%d", D, h, w);
    return launch_softargmin(cost, depth_values, depth_out, conf_out, D, h, w,
                             static_cast<hipStream_t>(stream));
}

// feats == NULL: features already in the workspace's C8 slot (see warp_variance_impl)
static int depth_infer_impl(const float* feats, const float* proj, const float* depth_values,
                            const void* weights_blob, float* depth_out, float* conf_out, void* workspace,
                            size_t workspace_bytes, int N, int C, int D, int h, int w, int dtype,
                            void* stream) {
    if (!proj || !depth_values || !weights_blob || !depth_out || !conf_out || !workspace)
        return fail(MVS_ERR_NULL, "mvs_depth_infer: NULL argument");
    if (int st = check_dims(N, C, D, h, w, dtype)) return st;
    const Workspace W = workspace_layout(N, C, D, h, w, dtype);
    if (workspace_bytes < W.total)
        return fail(MVS_ERR_WORKSPACE, "workspace needs >= %zu bytes, got %zu", W.total, workspace_bytes);
    if (reinterpret_cast<uintptr_t>(workspace) & 255)
        return fail(MVS_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    char* ws = static_cast<char*>(workspace);
    float* rt = reinterpret_cast<float*>(ws + W.rt);
    void* var = ws + W.var;
    float* cost = reinterpret_cast<float*>(ws + W.cost);
    int st;
    // from NCHW features the relative projections ride along with the layout transpose (one launch
    // less); with the features already C8-planar in the workspace they get their own 1-block launch
    const bool fold_proj = feats != nullptr && N > 1 && N <= 256;
    if (!fold_proj && (st = mvs_relative_proj(proj, rt, N, stream))) return st;
    // CostRegNet activations live behind the variance volume; hand costreg the sub-workspace that
    // starts at act[0] laid out as for N = 1 (same relative offsets).
    const Workspace W1 = workspace_layout(1, C, D, h, w, dtype);
    char* sub = ws + (W.act[0] - W1.act[0]);
    const size_t sub_bytes = workspace_bytes - (size_t)(sub - ws);
    // The variance volume is materialised and conv0 runs as its own kernel: a fused producer/consumer
    // kernel measured 1.40 ms against 0.16 + 0.34 ms at cfg2 (csrc/attic/warp_conv0_fused.hip).
    if ((st = warp_variance_impl(feats, rt, depth_values, var, workspace, workspace_bytes, N, C, D, h, w, dtype,
                                 stream, fold_proj ? proj : nullptr)))
        return st;
    if ((st = costreg_impl(var, weights_blob, cost, sub, sub_bytes, D, h, w, dtype, stream))) return st;
    return mvs_softargmin_conf(cost, depth_values, depth_out, conf_out, D, h, w, stream);
}

int mvs_depth_infer(const float* feats, const float* proj, const float* depth_values,
                    const void* weights_blob, float* depth_out, float* conf_out, void* workspace,
                    size_t workspace_bytes, int N, int C, int D, int h, int w, int dtype,
                    void* stream) {
    if (!feats) return fail(MVS_ERR_NULL, "mvs_depth_infer: NULL argument");
    return depth_infer_impl(feats, proj, depth_values, weights_blob, depth_out, conf_out, workspace,
                            workspace_bytes, N, C, D, h, w, dtype, stream);
}

// ---- FeatureNet (models/mvsnet.py:10-30) and the whole MVSNet.forward from images -------------

int mvs_query_feature_blob(size_t* bytes) {
    if (!bytes) return fail(MVS_ERR_NULL, "bytes is NULL");
    *bytes = feat_blob_layout().total_floats * sizeof(float);
    return MVS_OK;
}

int mvs_pack_feature_weights(const float* const* conv_weights, const float* const* bn_params,
                             const float* feature_bias, float eps, void* blob_out, size_t blob_bytes) {
    if (!conv_weights || !bn_params || !feature_bias || !blob_out)
        return fail(MVS_ERR_NULL, "mvs_pack_feature_weights: NULL argument");
    const FeatBlob L = feat_blob_layout();
    if (blob_bytes < L.total_floats * sizeof(float))
        return fail(MVS_ERR_WORKSPACE, "feature blob needs %zu bytes, got %zu",
                    L.total_floats * sizeof(float), blob_bytes);
    float* blob = static_cast<float*>(blob_out);
    std::memset(blob, 0, L.total_floats * sizeof(float));
    for (int l = 0; l < MVS_FEATURE_LAYERS; ++l) {
        const FeatLayerSpec& S = kFeatLayers[l];
        if (!conv_weights[l]) return fail(MVS_ERR_NULL, "feature conv weight %d is NULL", l);
        const size_t per_co = (size_t)S.cin * S.k * S.k;
        std::vector<float> wf((size_t)S.cout * per_co);
        for (int co = 0; co < S.cout; ++co) {
            float scale = 1.0f, shift = feature_bias[co];
            if (l < 7) {  // eval BatchNorm2d folded: y = (conv - mean) * gamma / sqrt(var + eps) + beta
                const float* g = bn_params[4 * l + 0];
                const float* b = bn_params[4 * l + 1];
                const float* m = bn_params[4 * l + 2];
                const float* v = bn_params[4 * l + 3];
                if (!g || !b || !m || !v) return fail(MVS_ERR_NULL, "feature BN parameter of layer %d is NULL", l);
                scale = g[co] / std::sqrt(v[co] + eps);
                shift = b[co] - m[co] * scale;
            }
            for (size_t i = 0; i < per_co; ++i) wf[co * per_co + i] = conv_weights[l][co * per_co + i] * scale;
            blob[L.bias_off[l] + co] = shift;
        }
        pack_fconv_weights(wf.data(), S.cin, S.cout, S.k, blob + L.panel_off[l]);
        if (l == 0) {  // plain copy for the VALU half of the fused conv0+conv1 kernel
            float* d = blob + L.l0_direct_off;
            for (int co = 0; co < 8; ++co) {
                for (int k = 0; k < 27; ++k) d[k * 8 + co] = wf[co * 27 + k];
                d[27 * 8 + co] = blob[L.bias_off[0] + co];
            }
        }
    }
    return MVS_OK;
}

static int check_image_dims(int N, int H, int W) {
    if (N < 1 || N > 65535) return fail(MVS_ERR_BAD_SHAPE, "number of images N=%d outside [1,65535]", N);
    if (H < 4 || W < 4) return fail(MVS_ERR_BAD_SHAPE, "image H,W = %d,%d must be >= 4", H, W);
    if ((size_t)N * H * W * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "N*H*W = %zu exceeds 31-bit offsets", (size_t)N * H * W);
    return MVS_OK;
}

int mvs_query_feature_workspace(int N, int H, int W, size_t* bytes) {
    if (!bytes) return fail(MVS_ERR_NULL, "bytes is NULL");
    if (int st = check_image_dims(N, H, W)) return st;
    *bytes = feat_workspace_layout(N, H, W).total;
    return MVS_OK;
}

int mvs_feature_layer(int layer, const float* x, float* y, const void* feature_blob, int N, int Hi, int Wi,
                      void* stream) {
    if (!x || !y || !feature_blob) return fail(MVS_ERR_NULL, "mvs_feature_layer: NULL argument");
    if (layer < 0 || layer >= MVS_FEATURE_LAYERS) return fail(MVS_ERR_BAD_SHAPE, "feature layer %d outside [0,8)", layer);
    if (int st = check_image_dims(N, Hi, Wi)) return st;
    return launch_feature_layer(layer, x, y, static_cast<const float*>(feature_blob), N, Hi, Wi,
                                static_cast<hipStream_t>(stream));
}

int mvs_feature_net(const float* imgs, const void* feature_blob, float* feats_out, void* workspace,
                    size_t workspace_bytes, int N, int H, int W, void* stream) {
    return mvs_feature_net_fmt(imgs, MVS_IMG_F32_CHW, feature_blob, feats_out, workspace, workspace_bytes, N, H, W, stream);
}

int mvs_feature_net_fmt(const void* imgs, int image_format, const void* feature_blob, float* feats_out, void* workspace,
                        size_t workspace_bytes, int N, int H, int W, void* stream) {
    if (!imgs || !feature_blob || !feats_out || !workspace)
        return fail(MVS_ERR_NULL, "mvs_feature_net: NULL argument");
    if (image_format < MVS_IMG_F32_CHW || image_format > MVS_IMG_U8_HWC)
        return fail(MVS_ERR_BAD_DTYPE, "mvs_feature_net: unknown image format %d", image_format);
    if (int st = check_image_dims(N, H, W)) return st;
    const FeatWorkspace F = feat_workspace_layout(N, H, W);
    if (workspace_bytes < F.total)
        return fail(MVS_ERR_WORKSPACE, "feature workspace needs >= %zu bytes, got %zu", F.total, workspace_bytes);
    if (reinterpret_cast<uintptr_t>(workspace) & 255)
        return fail(MVS_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    char* ws = static_cast<char*>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* c8 = reinterpret_cast<float*>(ws + F.c8);
    if (int st = launch_feature_net_c8(imgs, image_format, static_cast<const float*>(feature_blob), c8,
                                       reinterpret_cast<float*>(ws + F.bufA),
                                       reinterpret_cast<float*>(ws + F.bufB), N, H, W, s))
        return st;
    return launch_c8_to_nchw(c8, feats_out, N, kC, F.h4, F.w4, s);
}

int mvs_query_forward_workspace(int N, int H, int W, int D, int dtype, size_t* bytes) {
    if (!bytes) return fail(MVS_ERR_NULL, "bytes is NULL");
    if (H % 4 || W % 4) return fail(MVS_ERR_BAD_SHAPE, "image H,W = %d,%d must be multiples of 32", H, W);
    if (int st = check_dims(N, kC, D, H / 4, W / 4, dtype)) return st;
    if (int st = check_image_dims(N, H, W)) return st;
    *bytes = workspace_layout(N, kC, D, H / 4, W / 4, dtype).total + feat_workspace_layout(N, H, W).total;
    return MVS_OK;
}

int mvs_forward_images(const float* imgs, const float* proj, const float* depth_values,
                       const void* feature_blob, const void* weights_blob, float* depth_out,
                       float* conf_out, void* workspace, size_t workspace_bytes, int N, int H, int W,
                       int D, int dtype, void* stream) {
    return mvs_forward_images_fmt(imgs, MVS_IMG_F32_CHW, proj, depth_values, feature_blob, weights_blob, depth_out,
                                  conf_out, workspace, workspace_bytes, N, H, W, D, dtype, stream);
}

int mvs_forward_images_fmt(const void* imgs, int image_format, const float* proj, const float* depth_values,
                           const void* feature_blob, const void* weights_blob, float* depth_out,
                           float* conf_out, void* workspace, size_t workspace_bytes, int N, int H, int W,
                           int D, int dtype, void* stream) {
    if (image_format < MVS_IMG_F32_CHW || image_format > MVS_IMG_U8_HWC)
        return fail(MVS_ERR_BAD_DTYPE, "mvs_forward_images: unknown image format %d", image_format);
    if (!imgs || !proj || !depth_values || !feature_blob || !weights_blob || !depth_out || !conf_out ||
        !workspace)
        return fail(MVS_ERR_NULL, "mvs_forward_images: NULL argument");
    if (H % 4 || W % 4) return fail(MVS_ERR_BAD_SHAPE, "image H,W = %d,%d must be multiples of 32", H, W);
    const int h = H / 4, w = W / 4;
    if (int st = check_dims(N, kC, D, h, w, dtype)) return st;
    if (int st = check_image_dims(N, H, W)) return st;
    const Workspace Wd = workspace_layout(N, kC, D, h, w, dtype);
    const FeatWorkspace F = feat_workspace_layout(N, H, W);
    if (workspace_bytes < Wd.total + F.total)
        return fail(MVS_ERR_WORKSPACE, "workspace needs >= %zu bytes, got %zu", Wd.total + F.total, workspace_bytes);
    if (reinterpret_cast<uintptr_t>(workspace) & 255)
        return fail(MVS_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    char* ws = static_cast<char*>(workspace);
    char* fws = ws + Wd.total;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* feats_t = reinterpret_cast<float*>(ws + Wd.feats_t);
    float* bufA = reinterpret_cast<float*>(fws + F.bufA);
    float* bufB = reinterpret_cast<float*>(fws + F.bufB);
    const bool narrow = feat16_gather() && dtype != MVS_F32;
    // the last layer writes the path's C8 feature slot directly (fp32), or a scratch copy that is
    // then narrowed into it for the 16-bit gather
    float* c8 = narrow ? reinterpret_cast<float*>(fws + F.c8) : feats_t;
    if (int st = launch_feature_net_c8(imgs, image_format, static_cast<const float*>(feature_blob), c8, bufA, bufB, N, H,
                                       W, s))
        return st;
    if (narrow)
        if (int st = launch_narrow_features(c8, feats_t, (size_t)N * kC * h * w, dtype, s)) return st;
    return depth_infer_impl(nullptr, proj, depth_values, weights_blob, depth_out, conf_out, workspace,
                            Wd.total, N, kC, D, h, w, dtype, stream);
}

int mvs_homo_warp(const float* src_fea, const float* rt, const float* depth_values, float* out,
                  int C, int D, int h, int w, void* stream) {
    if (!src_fea || !rt || !depth_values || !out) return fail(MVS_ERR_NULL, "mvs_homo_warp: NULL argument");
    if (C < 1 || D < 1 || h < 2 || w < 2) return fail(MVS_ERR_BAD_SHAPE, "C,D,h,w = %d,%d,%d,%d", C, D, h, w);
    return launch_homo_warp(src_fea, rt, depth_values, out, C, D, h, w, static_cast<hipStream_t>(stream));
}

int mvs_depth_regression(const float* p, const float* depth_values, float* depth_out, int D,
                         int h, int w, void* stream) {
    if (!p || !depth_values || !depth_out) return fail(MVS_ERR_NULL, "mvs_depth_regression: NULL argument");
    if (D < 1 || h < 1 || w < 1) return fail(MVS_ERR_BAD_SHAPE, "D,h,w = %d,%d,%d", D, h, w);
    return launch_depth_regression(p, depth_values, depth_out, D, h, w, static_cast<hipStream_t>(stream));
}

}  // extern "C"
