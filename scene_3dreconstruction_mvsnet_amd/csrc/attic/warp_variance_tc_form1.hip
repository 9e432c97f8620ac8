// attic: the first form of the tap-cache warp + variance kernel (8 channels per thread, 250 VGPRs, 8 waves per CU;
// 0.2345 ms at cfg2 against 0.152 ms for the second form).  Not built.
// one tap = the 8 channels of one C8 plane at element offset `off` of the feature copy (fp32: 32 B,
// 16-bit features: 16 B, widened to fp32 once when gathered).  Plain loads on purpose: with
// inline-asm gathers (all views' gathers in flight at once, one manual s_waitcnt) the kernel ran 4 %
// faster, but hipcc is free to copy asm output registers at control-flow joins -- i.e. to read a
// register whose load is still in flight -- and did so as soon as the reload logic had two paths.
template <int FDT>
__device__ __forceinline__ void gather_tap(f32x4& lo, f32x4& hi, unsigned off, const void* base) {
    if constexpr (FDT == MVS_F32) {
        const float* p = static_cast<const float*>(base) + off;
        lo = *reinterpret_cast<const f32x4*>(p);
        hi = *reinterpret_cast<const f32x4*>(p + 4);
    } else {
        float v[8];
        load8_16<FDT>(base, off, v);
        lo = (f32x4){v[0], v[1], v[2], v[3]};
        hi = (f32x4){v[4], v[5], v[6], v[7]};
    }
}

template <int DT, int FDT, int NV>
__global__ __launch_bounds__(256) void warp_variance_tc_kernel(const void* __restrict__ feats_p,   // [4][N][hw][8] FDT
                                                               const float* __restrict__ rt,
                                                               const float* __restrict__ dv,
                                                               void* __restrict__ var, int N, int D, int h,
                                                               int w, int slab) {
    const int pl = threadIdx.x & 3;
    const int hw = h * w;
    const int p_raw = blockIdx.x * kTcPixPerBlock + (threadIdx.x >> 2);
    const bool live = p_raw < hw;
    const int p = live ? p_raw : hw - 1;  // keep whole quads / waves converged for the DPP exchange
    const int y = p / w, x = p - y * w;
    const int d0 = blockIdx.y * slab, d1 = min(d0 + slab, D);
    // 32-bit element offsets from the uniform base keep the gathers in the saddr + voffset form (one
    // VGPR per address instead of a 64-bit pair); the launcher guarantees 4*N*hw*8 floats < 2^31 bytes
    const unsigned plane = (unsigned)pl * (unsigned)N * (unsigned)hw * 8u;  // this thread's channel plane
    const float sx = (float)w / (float)(w - 1), sy = (float)h / (float)(h - 1);
    const float fx = (float)x, fy = (float)y;
    const float inv_n = 1.0f / (float)N;
    const size_t V0 = (size_t)D * hw;

    f32x4 r_lo, r_hi;
    gather_tap<FDT>(r_lo, r_hi, plane + (unsigned)p * 8u, feats_p);
    const f32x2 refp[4] = {{r_lo.x, r_lo.y}, {r_lo.z, r_lo.w}, {r_hi.x, r_hi.y}, {r_hi.z, r_hi.w}};

    // the projection this lane evaluates for its quad: source view min(pl, NV-1) + 1
    const int myv = pl < NV ? pl : NV - 1;
    const float* r = rt + (size_t)myv * 12;
    const float qx = fmaf(r[0], fx, fmaf(r[1], fy, r[2]));
    const float qy = fmaf(r[3], fx, fmaf(r[4], fy, r[5]));
    const float qz = fmaf(r[6], fx, fmaf(r[7], fy, r[8]));
    const float tx = r[9], ty = r[10], tz = r[11];

    f32x4 tap[NV][4][2];  // cached taps: [view][00,01,10,11][lo,hi]
    int key00[NV], key01[NV], key10[NV], key11[NV];  // offsets of the cached taps
#pragma unroll
    for (int v = 0; v < NV; ++v) key00[v] = key01[v] = key10[v] = key11[v] = -1;

    // Software pipeline per depth: (1) compare each view's tap offsets with the cached ones and issue
    // the re-gathers, (2) evaluate the NEXT depth's projection while those loads are in flight,
    // (3) blend with this depth's weights, accumulate, store.
    Samp mine = make_samp(qx, qy, qz, tx, ty, tz, dv[d0], sx, sy, h, w, 0, 0, w, h);
    for (int d = d0; d < d1; ++d) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int o00 = quad_bcast(mine.o00, v), o11 = quad_bcast(mine.o11, v);
            // o00 = ya*w + xa and o11 = yb*w + xb pin down all four (clamped) tap offsets
            if (o00 != key00[v] || o11 != key11[v]) {
                const int o01 = quad_bcast(mine.o01, v), o10 = quad_bcast(mine.o10, v);
                const unsigned vb = plane + (unsigned)(v + 1) * (unsigned)hw * 8u;  // elements
                // the usual move is one texel along x: the new left column is the old right column
                // (or vice versa) -- shift the cached taps and gather only the two new ones
                const bool left = (o00 == key01[v]) && (o10 == key11[v]);   // cell moved +1 in x
                const bool right = (o01 == key00[v]) && (o11 == key10[v]);   // cell moved -1 in x
                if (left) {
                    tap[v][0][0] = tap[v][1][0]; tap[v][0][1] = tap[v][1][1];
                    tap[v][2][0] = tap[v][3][0]; tap[v][2][1] = tap[v][3][1];
                    gather_tap<FDT>(tap[v][1][0], tap[v][1][1], vb + (unsigned)o01 * 8u, feats_p);
                    gather_tap<FDT>(tap[v][3][0], tap[v][3][1], vb + (unsigned)o11 * 8u, feats_p);
                } else if (right) {
                    tap[v][1][0] = tap[v][0][0]; tap[v][1][1] = tap[v][0][1];
                    tap[v][3][0] = tap[v][2][0]; tap[v][3][1] = tap[v][2][1];
                    gather_tap<FDT>(tap[v][0][0], tap[v][0][1], vb + (unsigned)o00 * 8u, feats_p);
                    gather_tap<FDT>(tap[v][2][0], tap[v][2][1], vb + (unsigned)o10 * 8u, feats_p);
                } else {
                    gather_tap<FDT>(tap[v][0][0], tap[v][0][1], vb + (unsigned)o00 * 8u, feats_p);
                    gather_tap<FDT>(tap[v][1][0], tap[v][1][1], vb + (unsigned)o01 * 8u, feats_p);
                    gather_tap<FDT>(tap[v][2][0], tap[v][2][1], vb + (unsigned)o10 * 8u, feats_p);
                    gather_tap<FDT>(tap[v][3][0], tap[v][3][1], vb + (unsigned)o11 * 8u, feats_p);
                }
                key00[v] = o00;
                key01[v] = o01;
                key10[v] = o10;
                key11[v] = o11;
            }
        }
        const Samp next = make_samp(qx, qy, qz, tx, ty, tz, dv[min(d + 1, D - 1)], sx, sy, h, w, 0, 0, w, h);
        // blend + accumulate on channel pairs: v_pk_fma_f32 / v_pk_mul_f32 do two lanes' worth per issue
        f32x2 S[4], Q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            S[j] = refp[j];
            Q[j] = refp[j] * refp[j];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float w00 = quad_bcast(mine.w00, v), w01 = quad_bcast(mine.w01, v);
            const float w10 = quad_bcast(mine.w10, v), w11 = quad_bcast(mine.w11, v);
            const f32x2 W00 = {w00, w00}, W01 = {w01, w01}, W10 = {w10, w10}, W11 = {w11, w11};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int hh = j >> 1, q = (j & 1) * 2;
                const f32x2 a = {tap[v][0][hh][q], tap[v][0][hh][q + 1]};
                const f32x2 bb = {tap[v][1][hh][q], tap[v][1][hh][q + 1]};
                const f32x2 c = {tap[v][2][hh][q], tap[v][2][hh][q + 1]};
                const f32x2 e = {tap[v][3][hh][q], tap[v][3][hh][q + 1]};
                // a*w00 + (b*w01 + (c*w10 + e*w11)) -- the plain kernel's nesting, per component
                const f32x2 wv = __builtin_elementwise_fma(a, W00, __builtin_elementwise_fma(bb, W01,
                                 __builtin_elementwise_fma(c, W10, e * W11)));
                S[j] = S[j] + wv;
                Q[j] = __builtin_elementwise_fma(wv, wv, Q[j]);
            }
        }
        if (live) {
            f32x2 o[4];
            const f32x2 IN = {inv_n, inv_n};
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // var = Q/N - (S/N)^2   (models/mvsnet.py:177)
                const f32x2 m = S[j] * IN;
                o[j] = __builtin_elementwise_fma(-m, m, Q[j] * IN);
            }
            const size_t oo = ((size_t)pl * V0 + (size_t)d * hw + p) * 8;
            St<DT>::store4(var, oo, (f32x4){o[0].x, o[0].y, o[1].x, o[1].y});
            St<DT>::store4(var, oo + 4, (f32x4){o[2].x, o[2].y, o[3].x, o[3].y});
        }
        mine = next;
    }
}

template <int DT, int FDT>
int launch_tc_dt(const void* feats_p, const float* rt, const float* dv, void* var, int N, int D, int h, int w,
                 int slab, hipStream_t s) {
    const dim3 grid((h * w + kTcPixPerBlock - 1) / kTcPixPerBlock, (D + slab - 1) / slab);
    if ((size_t)4 * N * h * w * 8 >= ((size_t)1 << 31))
        return fail(MVS_ERR_BAD_SHAPE, "warp_variance_tc: feature copy exceeds 31-bit offsets");
    switch (N - 1) {
        case 1: warp_variance_tc_kernel<DT, FDT, 1><<<grid, 256, 0, s>>>(feats_p, rt, dv, var, N, D, h, w, slab); break;
        case 2: warp_variance_tc_kernel<DT, FDT, 2><<<grid, 256, 0, s>>>(feats_p, rt, dv, var, N, D, h, w, slab); break;
        case 3: warp_variance_tc_kernel<DT, FDT, 3><<<grid, 256, 0, s>>>(feats_p, rt, dv, var, N, D, h, w, slab); break;
        case 4: warp_variance_tc_kernel<DT, FDT, 4><<<grid, 256, 0, s>>>(feats_p, rt, dv, var, N, D, h, w, slab); break;
        default: return fail(MVS_ERR_BAD_SHAPE, "warp_variance_tc: N = %d outside [2,5]", N);
    }
    return check_hip(hipGetLastError(), "warp_variance_tc launch");
}

