// softargmin.hip -- softmax over depth, depth expectation and photometric confidence (gfx950).
//
// Replaces, in one pass over the cost logits and without materialising the probability volume:
//   models/mvsnet.py:192-193  squeeze + F.softmax(dim=1)
//   models/module.py:144-147  depth_regression  (depth = sum_d p_d * depth_values_d)
//   models/mvsnet.py:214-218  sum4 = 4*avg_pool3d(pad(p,(1,2))), idx = trunc(sum_d p_d*d),
//                             confidence = sum4[idx] = p[idx-1]+p[idx]+p[idx+1]+p[idx+2]
//
// cost is [D][h][w] fp32: consecutive lanes = consecutive pixels, so every depth row is read
// fully coalesced.  A block owns 16 pixels; its 256 threads split D into 16 slices, each doing an
// online softmax (running max / sum / weighted sums), merged through LDS.  HBM-bound:
// algorithmic bytes = D*h*w*4 + 2*h*w*4.  The kernel is latency, not bandwidth.
#include "mvs_internal.h"

namespace mvs {

// Round 3: a slice's logits (<= 16 per thread for D <= 256) are requested in ONE batch and stay in registers; the four
// confidence taps come from those registers (every slice adds the exp of its own depths inside [idx - 1, idx + 2],
// merged through LDS) instead of a third dependent trip to global memory, and each slice takes its maximum first
// (one exp per logit, no running rescale).  Three dependent round trips -> one and half the exps moved the kernel
// from 0.0147 to 0.0142 ms at cfg2 only: it is neither, but 1,280 short blocks of launch / barrier latency.
// Larger D falls back to the looping form.
// Round 4: PIX / NS are template parameters.  With 16 pixels per block a wave's load touched four 64-byte segments (a
// quarter of four cache lines each); 32 pixels x 8 slices reads whole 128-byte lines (cfg3: 121 MB of logits at 2.0 ->
// see DESIGN section 6) and still launches hw / 32 blocks.
template <int MAXPER, int PIX, int NS>
__global__ __launch_bounds__(256) void softargmin_conf_kernel(const float* __restrict__ cost,
                                                              const float* __restrict__ dv,
                                                              float* __restrict__ depth,
                                                              float* __restrict__ conf, int D,
                                                              int hw) {
    static_assert(PIX * NS == 256, "pixels per block x depth slices = 256 threads");
    __shared__ float s_m[NS][PIX], s_s[NS][PIX], s_d[NS][PIX], s_i[NS][PIX], s_c[NS][PIX];
    __shared__ float s_M[PIX], s_inv[PIX];
    __shared__ int s_idx[PIX];
    const int lane = threadIdx.x & (PIX - 1), slice = threadIdx.x / PIX;
    const int p_raw = blockIdx.x * PIX + lane;
    const bool active = p_raw < hw;
    const int p = active ? p_raw : hw - 1;   // inactive lanes shadow the last pixel (no divergent barriers)
    const int per = (D + NS - 1) / NS;
    const int d0 = slice * per, d1 = min(d0 + per, D);
    float cbuf[MAXPER];
#pragma unroll
    for (int j = 0; j < MAXPER; ++j) cbuf[j] = cost[(size_t)min(d0 + j, D - 1) * hw + p];   // one batch of independent loads
    // the slice's maximum first, then ONE exp per logit (no running rescale: half the exps, no divergent branch)
    float m = -INFINITY, s = 0.0f, sd = 0.0f, si = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXPER; ++j)
        if (d0 + j < d1) m = fmaxf(m, cbuf[j]);
#pragma unroll
    for (int j = 0; j < MAXPER; ++j) {
        const int d = d0 + j;
        if (d < d1) {
            const float e = expf(cbuf[j] - m);
            s += e;
            sd = fmaf(e, dv[d], sd);
            si = fmaf(e, (float)d, si);
        }
    }
    s_m[slice][lane] = m; s_s[slice][lane] = s; s_d[slice][lane] = sd; s_i[slice][lane] = si;
    __syncthreads();
    if (slice == 0) {
        float M = s_m[0][lane];
#pragma unroll
        for (int k = 1; k < NS; ++k) M = fmaxf(M, s_m[k][lane]);
        float S = 0.0f, SD = 0.0f, SI = 0.0f;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float r = (s_m[k][lane] == -INFINITY) ? 0.0f : expf(s_m[k][lane] - M);
            S = fmaf(s_s[k][lane], r, S);
            SD = fmaf(s_d[k][lane], r, SD);
            SI = fmaf(s_i[k][lane], r, SI);
        }
        const float inv = 1.0f / S;
        if (active) depth[p] = SD * inv;
        int idx = (int)(SI * inv);  // .long() truncation (values are >= 0)
        idx = min(max(idx, 0), D - 1);
        s_M[lane] = M; s_inv[lane] = inv; s_idx[lane] = idx;
    }
    __syncthreads();
    {   // this slice's share of p[idx-1] + p[idx] + p[idx+1] + p[idx+2] (zero-padded), in depth order
        const float M = s_M[lane];
        const int idx = s_idx[lane];
        float c4 = 0.0f;
#pragma unroll
        for (int j = 0; j < MAXPER; ++j) {
            const int d = d0 + j;
            if (d < d1 && d >= idx - 1 && d <= idx + 2) c4 += expf(cbuf[j] - M);
        }
        s_c[slice][lane] = c4;
    }
    __syncthreads();
    if (slice != 0 || !active) return;
    float c4 = 0.0f;
#pragma unroll
    for (int k = 0; k < NS; ++k) c4 += s_c[k][lane];   // at most two slices are non-zero
    conf[p] = c4 * s_inv[lane];
}

// the looping form for D > 256 (a slice does not fit 16 registers): logits eight at a time, the confidence taps
// re-read from global memory
__global__ __launch_bounds__(256) void softargmin_conf_loop_kernel(const float* __restrict__ cost,
                                                                   const float* __restrict__ dv,
                                                                   float* __restrict__ depth,
                                                                   float* __restrict__ conf, int D,
                                                                   int hw) {
    constexpr int PIX = 16, NS = 16;
    __shared__ float s_m[NS][PIX], s_s[NS][PIX], s_d[NS][PIX], s_i[NS][PIX];
    const int lane = threadIdx.x & (PIX - 1), slice = threadIdx.x / PIX;
    const int p = blockIdx.x * PIX + lane;
    const bool active = p < hw;
    const int per = (D + NS - 1) / NS;
    const int d0 = slice * per, d1 = min(d0 + per, D);
    float m = -INFINITY, s = 0.0f, sd = 0.0f, si = 0.0f;
    if (active) {
        for (int db = d0; db < d1; db += 8) {
            float cbuf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) cbuf[j] = cost[(size_t)min(db + j, D - 1) * hw + p];   // independent loads
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = db + j;
                if (d >= d1) break;
                const float c = cbuf[j];
                if (c > m) {
                    const float r = expf(m - c);
                    s *= r; sd *= r; si *= r;
                    m = c;
                }
                const float e = expf(c - m);
                s += e;
                sd = fmaf(e, dv[d], sd);
                si = fmaf(e, (float)d, si);
            }
        }
    }
    s_m[slice][lane] = m; s_s[slice][lane] = s; s_d[slice][lane] = sd; s_i[slice][lane] = si;
    __syncthreads();
    if (slice != 0 || !active) return;
    float M = s_m[0][lane];
#pragma unroll
    for (int k = 1; k < NS; ++k) M = fmaxf(M, s_m[k][lane]);
    float S = 0.0f, SD = 0.0f, SI = 0.0f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float r = (s_m[k][lane] == -INFINITY) ? 0.0f : expf(s_m[k][lane] - M);
        S = fmaf(s_s[k][lane], r, S);
        SD = fmaf(s_d[k][lane], r, SD);
        SI = fmaf(s_i[k][lane], r, SI);
    }
    const float inv = 1.0f / S;
    depth[p] = SD * inv;
    int idx = (int)(SI * inv);  // .long() truncation (values are >= 0)
    idx = min(max(idx, 0), D - 1);
    float c4 = 0.0f;
#pragma unroll
    for (int k = -1; k <= 2; ++k) {
        const int j = idx + k;
        if (j >= 0 && j < D) c4 += expf(cost[(size_t)j * hw + p] - M);
    }
    conf[p] = c4 * inv;
}

int launch_softargmin(const float* cost, const float* dv, float* depth, float* conf, int D, int h,
                      int w, hipStream_t s) {
    const int hw = h * w;
    // 32 pixels x 8 slices when that still gives every CU two blocks (>= 512 blocks), else 16 x 16
    if (hw >= 32 * 512 && D <= 256) {
        const int per = (D + 7) / 8;
        if (per <= 16) softargmin_conf_kernel<16, 32, 8><<<(hw + 31) / 32, 256, 0, s>>>(cost, dv, depth, conf, D, hw);
        else if (per <= 24) softargmin_conf_kernel<24, 32, 8><<<(hw + 31) / 32, 256, 0, s>>>(cost, dv, depth, conf, D, hw);
        else softargmin_conf_kernel<32, 32, 8><<<(hw + 31) / 32, 256, 0, s>>>(cost, dv, depth, conf, D, hw);
        return check_hip(hipGetLastError(), "softargmin launch");
    }
    const int per = (D + 15) / 16;
    if (per <= 8) softargmin_conf_kernel<8, 16, 16><<<(hw + 15) / 16, 256, 0, s>>>(cost, dv, depth, conf, D, hw);
    else if (per <= 16) softargmin_conf_kernel<16, 16, 16><<<(hw + 15) / 16, 256, 0, s>>>(cost, dv, depth, conf, D, hw);
    else softargmin_conf_loop_kernel<<<(hw + 15) / 16, 256, 0, s>>>(cost, dv, depth, conf, D, hw);
    return check_hip(hipGetLastError(), "softargmin launch");
}

// depth = sum_d p[d] * depth_values[d]        (models/module.py:144-147)
__global__ __launch_bounds__(256) void depth_regression_kernel(const float* __restrict__ p,
                                                               const float* __restrict__ dv,
                                                               float* __restrict__ depth, int D,
                                                               int hw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw) return;
    float acc = 0.0f;
    for (int d = 0; d < D; ++d) acc = fmaf(p[(size_t)d * hw + i], dv[d], acc);
    depth[i] = acc;
}

int launch_depth_regression(const float* p, const float* dv, float* depth, int D, int h, int w,
                            hipStream_t s) {
    const int hw = h * w;
    depth_regression_kernel<<<(hw + 255) / 256, 256, 0, s>>>(p, dv, depth, D, hw);
    return check_hip(hipGetLastError(), "depth_regression launch");
}

}  // namespace mvs
