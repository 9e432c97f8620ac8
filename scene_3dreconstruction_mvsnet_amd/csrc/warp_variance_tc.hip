// warp_variance_tc.hip -- warp + variance with the bilinear taps cached in registers across depth.
//
// Why: the plain kernel (warp_variance.hip) is bound by the CU's 64 B/clk vector-L1 path: 8 GB of
// taps per cfg2 map.  But for a fixed (pixel, source view) the sampling position moves along the
// epipolar line by only a fraction of a texel per depth step (DTU-like rigs: 0.03-0.6 px), so the
// 2x2 texel cell -- hence all four taps -- is the same as at the previous depth ~80 % of the time.
// Only the bilinear weights change.  This kernel keeps the four taps of every source view in
// registers while a thread marches through a slab of depths and re-gathers a view's taps only when
// its cell index changed.  Same arithmetic as the plain kernel (same taps, same weight formula, same
// fma nesting), so results are bit-identical to it.
//
// Mapping: thread = (pixel, plane): 8 channels, 32 B per tap.  The four lanes of a quad own the four
// planes of one pixel; lane j of the quad evaluates the projection + weights of source view j+1
// (make_samp, ~60 VALU instructions) and the quad shares the results with DPP quad broadcasts, so
// the coordinate work per pixel does not grow with the finer thread granularity.
// Register budget: 32 VGPRs of cached taps per source view -> N-1 <= 4 source views (N <= 5); other
// view counts use the plain kernel.
#include <cstdlib>

#include "mvs_internal.h"
#include "storage.h"
#include "warp_common.h"

// Diagnostic builds (`make ablate11` .. `ablate14`, wrong results, timing only): 11 = 20 extra VALU per depth
// step, 12 = volume stores predicated off at run time (the arithmetic stays), 13 = no re-gathers after a
// block's first, 14 = 12 + 13.  Product builds leave MVS_ABLATE at 0.
#ifndef MVS_ABLATE
#define MVS_ABLATE 0
#endif

namespace mvs {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float quad_bcast(float v, int lane) {
    const int i = __float_as_int(v);
    int r;
    switch (lane) {  // dpp_ctrl quad_perm(l,l,l,l) must be an immediate
        case 0: r = __builtin_amdgcn_mov_dpp(i, 0x00, 0xF, 0xF, true); break;
        case 1: r = __builtin_amdgcn_mov_dpp(i, 0x55, 0xF, 0xF, true); break;
        case 2: r = __builtin_amdgcn_mov_dpp(i, 0xAA, 0xF, 0xF, true); break;
        default: r = __builtin_amdgcn_mov_dpp(i, 0xFF, 0xF, 0xF, true); break;
    }
    return __int_as_float(r);
}
__device__ __forceinline__ int quad_bcast(int v, int lane) {
    switch (lane) {
        case 0: return __builtin_amdgcn_mov_dpp(v, 0x00, 0xF, 0xF, true);
        case 1: return __builtin_amdgcn_mov_dpp(v, 0x55, 0xF, 0xF, true);
        case 2: return __builtin_amdgcn_mov_dpp(v, 0xAA, 0xF, 0xF, true);
        default: return __builtin_amdgcn_mov_dpp(v, 0xFF, 0xF, 0xF, true);
    }
}

// ---------------------------------------------------------------------------------------------
// The tap-cache kernel (second form; the first form, 8 channels per thread, is in attic/), templated on the
// channels per thread CPT.
//
// What rocprofv3 showed for the first form (profiles/r02_warp_pmc.md): only ~240 VALU and ~5 load
// instructions are executed per wave and depth step -- the re-gather branches are skipped at wave
// level three times out of four -- yet a step takes ~3,300 cycles, 53 % of them parked in
// s_waitcnt.  vmcnt counts loads and stores together in issue order, so every wait for re-gathered
// taps also waits for all older stores of the wave, i.e. for the HBM write latency; with 250 VGPRs
// only 8 waves per CU are resident, each with two depth steps of stores in flight: 32 KB per CU,
// which at ~3.6 us store latency is the 2.3 TB/s the kernel ran at.  The kernel is bound by the
// bytes it can keep in flight.  Hence:
//   * CPT = 4 channels per thread (8 lanes per pixel): the tap cache is 16 VGPRs per view instead of
//     32, the kernel fits 128 VGPRs and 16 waves per CU are resident (twice the stores in flight,
//     twice the VALU issue rate -- one wave issues at most one VALU per ~5 cycles on gfx950);
//   * ONE re-gather path per view (packed key o00 | dy | dx; all four taps are gathered again when
//     the 2x2 cell changed) instead of three; the taps of depth d+1 are requested right after depth
//     d's blend has read the cache and BEFORE depth d's stores, so the wait of the next step leaves
//     the two youngest stores in flight;
//   * raw buffer loads / stores: the per-view base travels in the scalar offset, a gather costs one
//     VALU (texel -> byte offset); out-of-range lanes of the last block fall outside the descriptor;
//   * three sampling records rotate through the depth loop (this depth's weights, next depth's cell
//     key, the projection being evaluated for d+2): no register copies.
// Same taps, weights and fma nesting as the first form and the plain kernel: bit-identical output.
//   * with 8 lanes per pixel and at most 4 source views (PAIR): quad q of a pixel evaluates the projections
//     of depth d + q, so one pass through make_samp_key serves two depth steps; the step for d + q moves
//     quad q's record into both quads (one bank-masked DPP row shift per field), then broadcasts lane v as
//     before: 136 instead of 152 VALU per wave and step.
// Measured at cfg2: 0.2345 ms (first form) -> 0.157 ms -> 0.152 ms with the pairing.  Ablation builds
// (make ablate11 .. ablate14, kernel only): +20 VALU per step +5 us, stores predicated off -20 us, no
// re-gathers after a block's first -23 us, neither -36 us (0.121 ms: the vector-instruction floor).  Tried
// on top and dropped: distributing the sampling records through a per-wave LDS ring (0.159 ms, no gain);
// non-temporal stores (slower); depth slabs of 48 / 96 (the 5,120 blocks of slab 24 are exactly five
// rounds of the 1,024 resident blocks).
// ---------------------------------------------------------------------------------------------

// Round 4 note (a dead end that cost correctness): v_pk_fma_f32 / v_pk_mul_f32 written as inline asm with op_sel so that
// two bilinear weights share one aligned register pair (128 instead of 133 VGPRs: a fourth wave per SIMD for N = 5 in fp32,
// -5 %).  Bit-identical to the plain kernel in every single-stream test -- and wrong now and then when a bf16-MFMA
// kernel of ANOTHER stream ran on the same CUs: one 16-lane pass of one depth step off by ~1 % in a LOW result that read the
// HIGH half of the weight pair (op_sel:[0,1,..]), 85 of 1,200 launches beside the split-operand tail
// (tools/probes/pair_two_streams.py); not an aliasing of destination and weight (tied / early-clobber operands: no change),
// not a missing wait state (s_nop in front: no change).  The blend below uses only the form hipcc itself emits (a scalar
// splat = op_sel_hi:[1,0,1], both halves read the LOW source): 0 of 1,200.  tests/test_gpu_fullsize.py::
// test_cfg2_two_maps_in_flight_are_bit_identical_to_one and test_two_host_threads_on_two_streams_share_one_module watch it.

struct SampK {
    int key;                    // (o00 << 2) | (dy << 1) | dx : the view's clamped 2x2 cell
    float w00, w01, w10, w11;   // bilinear weights (0 outside the image, NaN for non-finite coordinates)
};

__device__ __forceinline__ int clamp0(int v, int hi) {   // min(max(v, 0), hi) in one v_med3_i32
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(hi));
    return r;
}

// Same values as make_samp (warp_common.h) over the whole image -- identical sampling coordinate, weights and
// clamped tap offsets -- in ~50 instead of ~60 vector instructions: separable zero-padding (w00 = wx0 * wy0 with
// wx0 = x0 in range ? 1 - ax : 0; both factors are >= 0, so the products equal the selected products bit for
// bit), unsigned range tests, v_med3 for the clamps.
__device__ __forceinline__ SampK make_samp_key(float qx, float qy, float qz, float tx, float ty, float tz, float d,
                                               float sx, float sy, int h, int w) {
    const float X = fmaf(qx, d, tx), Y = fmaf(qy, d, ty), Z = fmaf(qz, d, tz);
    const float rz = __builtin_amdgcn_rcpf(Z);
    const float ix = (X * rz) * sx - 0.5f;   // px*W/(W-1) - 0.5      (module.py:129-136)
    const float iy = (Y * rz) * sy - 0.5f;
    const bool bad = !(fabsf(ix) <= 3.0e38f) || !(fabsf(iy) <= 3.0e38f);
    const float cx = __builtin_amdgcn_fmed3f(ix, -2.0f, (float)w + 1.0f);   // NaN -> -2 like fmin(fmax(ix, -2), w + 1)
    const float cy = __builtin_amdgcn_fmed3f(iy, -2.0f, (float)h + 1.0f);
    const float fx0 = floorf(cx), fy0 = floorf(cy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float ax = cx - fx0, ay = cy - fy0;
    const bool in = (cx == ix) && (cy == iy);
    const bool x0ok = in && (unsigned)x0 < (unsigned)w, x1ok = in && (unsigned)(x0 + 1) < (unsigned)w;
    const bool y0ok = (unsigned)y0 < (unsigned)h, y1ok = (unsigned)(y0 + 1) < (unsigned)h;
    float wx0 = x0ok ? 1.0f - ax : 0.0f, wx1 = x1ok ? ax : 0.0f;
    const float wy0 = y0ok ? 1.0f - ay : 0.0f, wy1 = y1ok ? ay : 0.0f;
    if (bad) { wx0 = NAN; wx1 = NAN; }   // NaN * (anything, 0 included) = NaN: all four weights
    SampK k;
    k.w00 = wx0 * wy0; k.w01 = wx1 * wy0; k.w10 = wx0 * wy1; k.w11 = wx1 * wy1;
    const int xa = clamp0(x0, w - 1), xb = clamp0(x0 + 1, w - 1);
    const int ya = clamp0(y0, h - 1), yb = clamp0(y0 + 1, h - 1);
    // o00 = ya*w + xa; o01 = o00 + dx, o10 = o00 + dy*w, o11 = o10 + dx with dx, dy in {0, 1} (clamped window)
    const int o00 = ya * w + xa;
    k.key = ((o00 * 2 + (yb - ya)) * 2) + (xb - xa);
    return k;
}

typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
typedef unsigned u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned))));
typedef _Float16 f16x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));

// CPT consecutive channels of one texel -> CPT/4 f32x4 (16-bit features are widened once here)
template <int FDT, int CPT>
__device__ __forceinline__ void gather_tap_buf(f32x4 (&t)[CPT / 4], __amdgpu_buffer_rsrc_t rs, unsigned voff,
                                               unsigned soff) {
    if constexpr (FDT == MVS_F32) {
#pragma unroll
        for (int i = 0; i < CPT / 4; ++i)
            t[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(voff + 16u * i), (int)soff, 0));
    } else if constexpr (CPT == 8) {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
        if constexpr (FDT == MVS_F16) {
            const f16x8s hv = __builtin_bit_cast(f16x8s, raw);
            t[0] = (f32x4){(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
            t[1] = (f32x4){(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
        } else {
            const bf16x8s hv = __builtin_bit_cast(bf16x8s, raw);
            t[0] = (f32x4){(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
            t[1] = (f32x4){(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
        }
    } else {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 0);
        if constexpr (FDT == MVS_F16) {
            const f16x4s hv = __builtin_bit_cast(f16x4s, raw);
            t[0] = (f32x4){(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        } else {
            const bf16x4s hv = __builtin_bit_cast(bf16x4s, raw);
            t[0] = (f32x4){(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        }
    }
}

// CPT consecutive channels of one voxel.  The scalar offset field stays an immediate 0 on purpose: a
// 16-byte buffer store whose soffset is an SGPR reads its data registers late, hipcc (ROCm 7.2) assumes
// that form has no store-data hazard and lets the next VALU instruction overwrite them -- on gfx950
// the last lanes of each row then stored the NEXT value of the second dword.  With an immediate
// soffset the compiler inserts the wait state itself.
template <int DT, int CPT, int NTS>
__device__ __forceinline__ void store_voxel_buf(__amdgpu_buffer_rsrc_t rs, unsigned voff, const f32x2 (&o)[CPT / 2]) {
    constexpr int aux = NTS ? 2 : 0;   // nt: measured slower (0.253 vs 0.233 ms), kept as MVS_WARP_NT=1
    if constexpr (DT == MVS_F32) {
#pragma unroll
        for (int i = 0; i < CPT / 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(
                __builtin_bit_cast(u32x4, (f32x4){o[2 * i].x, o[2 * i].y, o[2 * i + 1].x, o[2 * i + 1].y}), rs,
                (int)(voff + 16u * i), 0, aux);
    } else if constexpr (CPT == 8) {
        if constexpr (DT == MVS_F16) {
            const f16x8s hv = {(_Float16)o[0].x, (_Float16)o[0].y, (_Float16)o[1].x, (_Float16)o[1].y,
                               (_Float16)o[2].x, (_Float16)o[2].y, (_Float16)o[3].x, (_Float16)o[3].y};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rs, (int)voff, 0, aux);
        } else {
            const bf16x8s hv = {(__bf16)o[0].x, (__bf16)o[0].y, (__bf16)o[1].x, (__bf16)o[1].y,
                                (__bf16)o[2].x, (__bf16)o[2].y, (__bf16)o[3].x, (__bf16)o[3].y};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rs, (int)voff, 0, aux);
        }
    } else {
        if constexpr (DT == MVS_F16) {
            const f16x4s hv = {(_Float16)o[0].x, (_Float16)o[0].y, (_Float16)o[1].x, (_Float16)o[1].y};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), rs, (int)voff, 0, aux);
        } else {
            const bf16x4s hv = {(__bf16)o[0].x, (__bf16)o[0].y, (__bf16)o[1].x, (__bf16)o[1].y};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), rs, (int)voff, 0, aux);
        }
    }
}

// the lanes 4q .. 4q+3 of every 8-lane pixel group hand their value to the group's other quad
template <int Q>
__device__ __forceinline__ int pair_pick(int v) {
    if constexpr (Q == 0) return __builtin_amdgcn_update_dpp(v, v, 0x114, 0xF, 0xA, false);   // row_shr:4 into banks 1, 3
    else return __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);                     // row_shl:4 into banks 0, 2
}
template <int Q>
__device__ __forceinline__ float pair_pick(float v) { return __int_as_float(pair_pick<Q>(__float_as_int(v))); }

// PK = 1 (16-bit features, CPT = 4 only; experiment of round 4): the cached taps stay PACKED (8 instead of
// 16 VGPRs per view) and are widened inside the blend at every step -- bit-identical results (widening is exact), fewer
// registers (more resident waves), more vector instructions per step.  Measured: section 10 of DESIGN.md.
// four waves per SIMD: the N = 5 fp32 kernel allocates 130 VGPRs when left alone and fits 128 without spills when asked to
// (every other variant is below 128 anyway); `make ablate61`: no occupancy request, `ablate62`: five waves (spills)
#if MVS_ABLATE == 61
#define MVS_WARP_LB __launch_bounds__(256)
#elif MVS_ABLATE == 62
#define MVS_WARP_LB __launch_bounds__(256, 5)
#else
#define MVS_WARP_LB __launch_bounds__(256, 4)
#endif
template <int DT, int FDT, int NV, int CPT, int NTS, int PAIR, int PK = 0>
__global__ MVS_WARP_LB void warp_variance_tc2_kernel(const void* __restrict__ feats_p,   // [4][N][hw][8] FDT
                                                                const float* __restrict__ rt,
                                                                const float* __restrict__ dv,
                                                                void* __restrict__ var, int N, int D, int h,
                                                                int w, int slab, int depth_fastest) {
    // block order: pixel block fastest (default: the features stay L2-resident anyway), or depth slab
    // fastest when the feature maps exceed the L2s (1600x1184 inputs): blocks resident together then
    // work on the same pixels at neighbouring depths and share their source footprints in L2
    const int pblk = depth_fastest ? blockIdx.y : blockIdx.x;
    const int dblk = depth_fastest ? blockIdx.x : blockIdx.y;
    constexpr unsigned FES = FDT == MVS_F32 ? 4u : 2u;   // bytes per feature element
    constexpr unsigned VES = DT == MVS_F32 ? 4u : 2u;    // bytes per volume element
    constexpr int LPP = 32 / CPT;                        // lanes per pixel: 4 (CPT = 8) or 8 (CPT = 4)
    constexpr int NH = CPT / 4, NP = CPT / 2;            // f32x4 / channel pairs per tap
    const int sub = threadIdx.x & (LPP - 1);             // this thread's channels: sub * CPT .. + CPT - 1
    const int pl = (sub * CPT) >> 3, cin = (sub * CPT) & 7;   // C8 plane and first channel inside it
    const int hw = h * w;
    const int p_raw = pblk * (256 / LPP) + (threadIdx.x / LPP);
    const bool live = p_raw < hw;
    const int p = live ? p_raw : hw - 1;  // keep whole quads / waves converged for the DPP exchange
    const int y = p / w, x = p - y * w;
    const int d0 = dblk * slab, d1 = min(d0 + slab, D);
    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float fx = (float)x, fy = (float)y;
    const float inv_n = 1.0f / (float)N;

    // buffer descriptors: whole feature copy / whole volume (the launcher guarantees < 4 GiB each)
    const unsigned fbytes = 4u * (unsigned)N * (unsigned)hw * 8u * FES;
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(feats_p), (short)0, (int)fbytes, 0x00020000);
    const unsigned vbytes = 4u * (unsigned)D * (unsigned)hw * 8u * VES;
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(var, (short)0, (int)vbytes, 0x00020000);
    const unsigned plane_b = ((unsigned)pl * (unsigned)N * (unsigned)hw * 8u + (unsigned)cin) * FES;   // this thread's channels of view 0, texel 0
    const unsigned view_b = (unsigned)hw * 8u * FES;                                  // bytes per view (scalar)
    // the depth plane advances in the vector offset (see store_voxel_buf) by a UNIFORM step (an SGPR: one VGPR less --
    // the 128th); out-of-range lanes of the last block start so far beyond the descriptor's range that `slab` steps
    // neither reach 2^32 nor come back into it: 0xFFFFFFE0 - slab * step >= vbytes (warp_tc_fits checks it)
    const unsigned dstep_v = (unsigned)hw * 8u * VES;
    unsigned out_v = live ? ((((unsigned)pl * (unsigned)D + (unsigned)d0) * (unsigned)hw + (unsigned)p) * 8u + (unsigned)cin) * VES
                          : 0xFFFFFFE0u - (unsigned)slab * dstep_v;

    f32x4 rf[NH];
    gather_tap_buf<FDT, CPT>(rf, frs, plane_b + (unsigned)p * 8u * FES, 0u);
    f32x2 refp[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) refp[j] = (f32x2){rf[j >> 1][(j & 1) * 2], rf[j >> 1][(j & 1) * 2 + 1]};

    // the projection this lane evaluates for its quad: source view min(lane & 3, NV-1) + 1 (with 8 lanes
    // per pixel both quads of a pixel evaluate all views: same instruction stream, no cross-quad traffic)
    const int myv = (sub & 3) < NV ? (sub & 3) : NV - 1;
    const float* r = rt + (size_t)myv * 12;
    const float qx = fmaf(r[0], fx, fmaf(r[1], fy, r[2]));
    const float qy = fmaf(r[3], fx, fmaf(r[4], fy, r[5]));
    const float qz = fmaf(r[6], fx, fmaf(r[7], fy, r[8]));
    const float tx = r[9], ty = r[10], tz = r[11];

    static_assert(!PK || (CPT == 4 && FDT != MVS_F32), "packed tap cache: 16-bit features, 4 channels per thread");
    f32x4 tap[PK ? 1 : NV][4][NH];  // cached taps: [view][00,01,10,11][16-byte piece]
    u32x2 tapk[PK ? NV : 1][4];     // PK: the same taps as stored (4 x 16 bit)
    int key[NV];           // cell key of the cached taps
#pragma unroll
    for (int v = 0; v < NV; ++v) key[v] = -1;

    // (re-)gather the taps of every view whose 2x2 cell differs from the cached one; lane v of each quad
    // holds view v's cell key in `keys`
    auto regather = [&](int keys) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int k = quad_bcast(keys, v);
#if MVS_ABLATE == 13 || MVS_ABLATE == 14
            if (k != key[v] && (key[v] == -1 || slab < 0)) {
#else
            if (k != key[v]) {   // uniform over the lanes of a pixel: they share key and cache state
#endif
                const unsigned o00 = (unsigned)k >> 2, dx = (unsigned)k & 1u, dyw = (k & 2) ? (unsigned)w : 0u;
                const unsigned b00 = plane_b + o00 * (8u * FES);
                const unsigned soff = (unsigned)(v + 1) * view_b;
                if constexpr (PK) {
                    tapk[v][0] = __builtin_amdgcn_raw_buffer_load_b64(frs, (int)b00, (int)soff, 0);
                    tapk[v][1] = __builtin_amdgcn_raw_buffer_load_b64(frs, (int)(b00 + dx * (8u * FES)), (int)soff, 0);
                    tapk[v][2] = __builtin_amdgcn_raw_buffer_load_b64(frs, (int)(b00 + dyw * (8u * FES)), (int)soff, 0);
                    tapk[v][3] = __builtin_amdgcn_raw_buffer_load_b64(frs, (int)(b00 + (dyw + dx) * (8u * FES)), (int)soff, 0);
                } else {
                    gather_tap_buf<FDT, CPT>(tap[v][0], frs, b00, soff);
                    gather_tap_buf<FDT, CPT>(tap[v][1], frs, b00 + dx * (8u * FES), soff);
                    gather_tap_buf<FDT, CPT>(tap[v][2], frs, b00 + dyw * (8u * FES), soff);
                    gather_tap_buf<FDT, CPT>(tap[v][3], frs, b00 + (dyw + dx) * (8u * FES), soff);
                }
                key[v] = k;
            }
        }
    };

    // blend the cached taps with the weights held by lane v of each quad, accumulate sum / sum of squares
    // on channel pairs (v_pk_fma_f32 / v_pk_mul_f32)
    auto blend = [&](float mw00, float mw01, float mw10, float mw11, f32x2 (&S)[NP], f32x2 (&Q)[NP]) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            S[j] = refp[j];
            Q[j] = refp[j] * refp[j];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            // one weight at a time, broadcast to both halves of a channel pair (hipcc folds the splat into
            // op_sel_hi:[1,0,1] -- "both halves read the low source"); wv = a*w00 + (b*w01 + (c*w10 + e*w11)) -- the
            // plain kernel's nesting, per component
            auto tapv = [&](int t, int j) -> f32x2 {
                const int hh = j >> 1, q = (j & 1) * 2;
                if constexpr (PK) {
                    auto widen = [](unsigned d) -> f32x2 {   // two 16-bit values of one dword -> fp32 (exact)
                        if constexpr (FDT == MVS_F16) {
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            const h2 h = __builtin_bit_cast(h2, d);
                            return (f32x2){(float)h[0], (float)h[1]};
                        } else {
                            return (f32x2){__uint_as_float(d << 16), __uint_as_float(d & 0xFFFF0000u)};
                        }
                    };
                    return widen(tapk[v][t][j]);
                } else {
                    return (f32x2){tap[v][t][hh][q], tap[v][t][hh][q + 1]};
                }
            };
            f32x2 wv[NP];
            {
                const float w = quad_bcast(mw11, v);
#pragma unroll
                for (int j = 0; j < NP; ++j) wv[j] = tapv(3, j) * (f32x2){w, w};
            }
            {
                const float w = quad_bcast(mw10, v);
#pragma unroll
                for (int j = 0; j < NP; ++j) wv[j] = __builtin_elementwise_fma(tapv(2, j), (f32x2){w, w}, wv[j]);
            }
            {
                const float w = quad_bcast(mw01, v);
#pragma unroll
                for (int j = 0; j < NP; ++j) wv[j] = __builtin_elementwise_fma(tapv(1, j), (f32x2){w, w}, wv[j]);
            }
            {
                const float w = quad_bcast(mw00, v);
#pragma unroll
                for (int j = 0; j < NP; ++j) wv[j] = __builtin_elementwise_fma(tapv(0, j), (f32x2){w, w}, wv[j]);
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                S[j] = S[j] + wv[j];
                Q[j] = __builtin_elementwise_fma(wv[j], wv[j], Q[j]);
            }
        }
        // pin the sums here: without it hipcc sinks the blend below the re-gather branches, keeps the
        // old taps alive across them and needs a second set of tap registers (378 VGPRs + AGPRs)
#pragma unroll
        for (int j = 0; j < NP; ++j) asm volatile("" : "+v"(S[j]), "+v"(Q[j]));
    };
    auto variance_store = [&](const f32x2 (&S)[NP], const f32x2 (&Q)[NP]) {
        f32x2 o[NP];
        const f32x2 IN = {inv_n, inv_n};
#pragma unroll
        for (int j = 0; j < NP; ++j) {   // var = Q/N - (S/N)^2   (models/mvsnet.py:177)
            const f32x2 m = S[j] * IN;
            o[j] = __builtin_elementwise_fma(-m, m, Q[j] * IN);
        }
#if MVS_ABLATE == 12 || MVS_ABLATE == 14
        if (slab < 0)
#endif
        store_voxel_buf<DT, CPT, NTS>(vrs, out_v, o);
        out_v += dstep_v;
#if MVS_ABLATE == 11
        asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n"
                     "v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n"
                     "v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n"
                     "v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n"
                     : "+v"(out_v));
#endif
    };

    if constexpr (PAIR && LPP == 8) {
        // 8 lanes per pixel, 4 views: quad q of a pixel evaluates the projections of depth da + q, so one
        // pass through make_samp_key serves two depth steps; the step for da + q first moves quad q's
        // record into both quads (one DPP per field), then broadcasts lane v as before.
        const int eq = sub >> 2;
        auto eval = [&](int dd) {   // dd is even-aligned to the slab start; both depths' values come by scalar load
            const float da = dv[min(dd, D - 1)], db = dv[min(dd + 1, D - 1)];
            return make_samp_key(qx, qy, qz, tx, ty, tz, eq ? db : da, sx, sy, h, w);
        };
        // P holds depths da, da + 1; N receives da + 2, da + 3
        auto pair_step = [&](const SampK& P, SampK& N, int da) {
            f32x2 S[NP], Q[NP];
            blend(pair_pick<0>(P.w00), pair_pick<0>(P.w01), pair_pick<0>(P.w10), pair_pick<0>(P.w11), S, Q);
            if (da + 1 < d1) regather(pair_pick<1>(P.key));   // wave-uniform
            variance_store(S, Q);
            N = eval(da + 2);
            if (da + 1 < d1) {
                blend(pair_pick<1>(P.w00), pair_pick<1>(P.w01), pair_pick<1>(P.w10), pair_pick<1>(P.w11), S, Q);
                if (da + 2 < d1) regather(pair_pick<0>(N.key));
                variance_store(S, Q);
            }
        };
        SampK sa = eval(d0), sb;
        regather(pair_pick<0>(sa.key));
        for (int d = d0; d < d1; d += 4) {   // two records alternate: no register copies
            pair_step(sa, sb, d);
            if (d + 2 < d1) pair_step(sb, sa, d + 2);
        }
    } else {
        // one depth step: blend with `cur`'s weights, request `nxt`'s taps, variance + stores, evaluate `nn`
        auto step = [&](const SampK& cur, const SampK& nxt, SampK& nn, int d) {
            f32x2 S[NP], Q[NP];
            blend(cur.w00, cur.w01, cur.w10, cur.w11, S, Q);
            if (d + 1 < d1) regather(nxt.key);   // wave-uniform: the slab's last step has no successor
            variance_store(S, Q);
            nn = make_samp_key(qx, qy, qz, tx, ty, tz, dv[min(d + 2, D - 1)], sx, sy, h, w);
        };
        SampK sa = make_samp_key(qx, qy, qz, tx, ty, tz, dv[d0], sx, sy, h, w);
        SampK sb = make_samp_key(qx, qy, qz, tx, ty, tz, dv[min(d0 + 1, D - 1)], sx, sy, h, w), sc;
        regather(sa.key);
        int d = d0;
        for (; d + 2 < d1; d += 3) {   // three records rotate: no register copies
            step(sa, sb, sc, d);
            step(sb, sc, sa, d + 1);
            step(sc, sa, sb, d + 2);
        }
        if (d < d1) {
            step(sa, sb, sc, d);
            if (d + 1 < d1) step(sb, sc, sa, d + 1);
        }
    }
}

template <int DT, int FDT>
int launch_tc2_dt(const void* feats_p, const float* rt, const float* dv, void* var, int N, int D, int h, int w,
                  hipStream_t s) {
    constexpr int CPT = 4;                 // 8 lanes per pixel, 16 waves per CU (CPT = 8: 0.166 vs 0.157 ms at cfg2)
    // depths a thread marches through (cfg2, kernel + transpose: 8 / 12 / 16 / 24 / 32 / 48 / 64 = 0.183 / 0.172 / 0.168 /
    // 0.161 / 0.160 / 0.160 / 0.159 ms in round 3)
    // Round 4 sweep (bench per-kernel pass, cfg2 / cfg5 / cfg3): 12: 0.168 / 0.129 / 1.63; 20: 0.160 / 0.122 / 1.48; 24: 0.157
    // / 0.122 / 1.46; 28: 0.157 / 0.122 / 1.52; 40: 0.155 / 0.120 / 1.44; 52: 0.156 / 0.121 / 1.44; 88: 0.163 / 0.131 / 1.44
    // -- and a slab count that is a multiple of 8 is a trap in the depth-slab-fastest order (cfg3, D = 256: slab 16 /
    // 32 / 64 = 1.95 / 1.86 / 1.78 ms): the linear block id then sends slab s of EVERY pixel block to XCD s % 8, so every
    // XCD's L2 streams the whole feature set
    int slab = 40;
    for (const int cand : {40, 44, 36, 32, 28, 24}) {   // the first whose slab count is not a multiple of 8 (D = 320: 36)
        slab = cand;
        if (((D + cand - 1) / cand) % 8 != 0) break;
    }
    constexpr int pix = 256 / (32 / CPT);
    constexpr size_t fes = FDT == MVS_F32 ? 4 : 2;
    // all views' features against the 32 MB of aggregate L2 (MVS_WARP_DEPTH_FASTEST=1 forces the order)
    static const int force_df = [] {   // 1 = always, 0 = never (A/B runs), unset = by size
        const char* e = getenv("MVS_WARP_DEPTH_FASTEST");
        return e ? atoi(e) : -1;
    }();
    const int df = force_df >= 0 ? force_df : ((size_t)N * h * w * 32 * fes > ((size_t)24 << 20) ? 1 : 0);
    const unsigned npb = (h * w + pix - 1) / pix, nsl = (D + slab - 1) / slab;
    const dim3 grid = df ? dim3(nsl, npb) : dim3(npb, nsl);
    // PAIR = 1: one projection pass per two depth steps (quad q of a pixel evaluates depth d + q)
    // MVS_WARP_PACKED=1 (16-bit features only): the packed tap cache
    static const bool packed = [] {
        const char* e = getenv("MVS_WARP_PACKED");
        return e && e[0] == '1';
    }();
    constexpr int CANPK = FDT != MVS_F32 ? 1 : 0;
#define MVS_TC2(NV)                                                                                                        \
    if (CANPK && packed) warp_variance_tc2_kernel<DT, FDT, NV, CPT, 0, 1, CANPK><<<grid, 256, 0, s>>>(feats_p, rt, dv, var, N, D, h, w, slab, df); \
    else warp_variance_tc2_kernel<DT, FDT, NV, CPT, 0, 1><<<grid, 256, 0, s>>>(feats_p, rt, dv, var, N, D, h, w, slab, df);
    switch (N - 1) {
        case 1: MVS_TC2(1) break;
        case 2: MVS_TC2(2) break;
        case 3: MVS_TC2(3) break;
        case 4: MVS_TC2(4) break;
        default: return fail(MVS_ERR_BAD_SHAPE, "warp_variance_tc2: N = %d outside [2,5]", N);
    }
#undef MVS_TC2
    return check_hip(hipGetLastError(), "warp_variance_tc2 launch");
}

}  // namespace

// the kernel needs 32-bit byte offsets into the feature copy and the volume, and o00 << 2 in an int; 2 <= N <= 5
// (16 VGPRs of cached taps per source view).  Other problems run the plain kernel (warp_variance.hip).
bool warp_tc_fits(int N, int D, int h, int w, int fes, int ves) {
    const size_t hw = (size_t)h * w;
    // (+ 48 depth planes of one channel plane: the parking range of the last block's out-of-image lanes, slab <= 44)
    return N >= 2 && N <= 5 && 4 * (size_t)N * hw * 8 * fes < ((size_t)1 << 31) &&
           4 * (size_t)D * hw * 8 * ves + 48 * hw * 8 * ves < ((size_t)1 << 32) - 64 && hw < ((size_t)1 << 29);
}

// fp32 features [4][N][h][w][8], volume in `dtype`
int launch_warp_variance_tc(const float* feats_p, const float* rt, const float* dv, void* var, int N, int D,
                            int h, int w, int dtype, hipStream_t s) {
    if (!warp_tc_fits(N, D, h, w, 4, dtype == MVS_F32 ? 4 : 2))
        return fail(MVS_ERR_BAD_SHAPE, "warp_variance_tc: N,D,h,w = %d,%d,%d,%d outside the kernel's range", N, D, h, w);
    switch (dtype) {
        case MVS_F32: return launch_tc2_dt<MVS_F32, MVS_F32>(feats_p, rt, dv, var, N, D, h, w, s);
        case MVS_F16: return launch_tc2_dt<MVS_F16, MVS_F32>(feats_p, rt, dv, var, N, D, h, w, s);
        case MVS_BF16: return launch_tc2_dt<MVS_BF16, MVS_F32>(feats_p, rt, dv, var, N, D, h, w, s);
        default: return fail(MVS_ERR_BAD_DTYPE, "warp_variance_tc: unknown dtype %d", dtype);
    }
}

// 16-bit features (same dtype as the volume): half the gather bytes
int launch_warp_variance_tc16(const void* feats16, const float* rt, const float* dv, void* var, int N, int D,
                              int h, int w, int dtype, hipStream_t s) {
    if (!warp_tc_fits(N, D, h, w, 2, 2))
        return fail(MVS_ERR_BAD_SHAPE, "warp_variance_tc16: N,D,h,w = %d,%d,%d,%d outside the kernel's range", N, D, h, w);
    switch (dtype) {
        case MVS_F16: return launch_tc2_dt<MVS_F16, MVS_F16>(feats16, rt, dv, var, N, D, h, w, s);
        case MVS_BF16: return launch_tc2_dt<MVS_BF16, MVS_BF16>(feats16, rt, dv, var, N, D, h, w, s);
        default: return fail(MVS_ERR_BAD_DTYPE, "warp_variance_tc16 needs fp16 or bf16 (dtype %d)", dtype);
    }
}

}  // namespace mvs
