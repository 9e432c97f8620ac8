// attic: first form of the fused conv11 + prob kernel (shared input halo, six block barriers per step;
// 0.117-0.125 ms at cfg2 against 0.105 ms for the wave-private form, profiles/r02_conv11_prob.md).  Not built.
// NT = 256: 4 waves, 2 M-tiles and 2 logit rows per thread; NT = 512: 8 waves, one of each
template <int DT, int NT>
__global__ __launch_bounds__(NT, NT / 128) void conv11_prob_kernel(   // 2nd: waves per SIMD = two blocks per CU
    const void* __restrict__ x,        // [2][Di][Hi][Wi][8]  conv9's output (+ conv2 skip)
    const float* __restrict__ bp,      // deconv panel of layer 9: [2][1][9][64][4] (pack_deconvg_weights)
    const float* __restrict__ bias,    // [8]  folded BN shift of conv11
    const void* __restrict__ skip,     // [1][2Di][2Hi][2Wi][8]  conv0's output
    const float* __restrict__ pw,      // prob weights [27][8]
    const float* __restrict__ pbias,   // [1]
    float* __restrict__ cost,          // [2Di][2Hi][2Wi]
    int Di, int Hi, int Wi, int ZC, int nbx, int nby) {
    using namespace cp;
    constexpr int MPW = BY * BX / (NT / 64);       // M-tiles per wave
    constexpr int PPT = (NPIECE + NT - 1) / NT;    // staged pieces per thread
    constexpr int SKP = NSK / NT;                  // skip pieces per thread (half of them per plane)
    constexpr int RPT = OY * OX / NT;              // logit rows per thread
    __shared__ __attribute__((aligned(16))) float in_tile[IN_FLOATS];
    __shared__ __attribute__((aligned(16))) float ct[C_FLOATS];
    __shared__ __attribute__((aligned(16))) float bpan[2 * 9 * 64 * 4];   // the whole deconv panel, staged once

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby;
    const int bz = b / nby;
    const int ix0 = bx * (IX - 1), iy0 = by * (IY - 1);
    const int za = bz * ZC, zb = min(za + ZC, Di);      // logits of planes [2 za, 2 zb)
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t Vin = (size_t)Di * Hi * Wi;
    const size_t HWo = (size_t)Ho * Wo;

    // ---- input staging bookkeeping (per thread, z-independent) ----
    int grel[PPT], loff[PPT];
    unsigned inside = 0, hzbits = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = tid + i * NT;
        const int half = p & 1, v = p >> 1;
        const int hx = v % HX, t = v / HX;
        const int hy = t % HY, hz = t / HY;
        const int gy = iy0 + hy, gx = ix0 + hx;
        const bool ok = p < NPIECE && gy < Hi && gx < Wi;
        grel[i] = ok ? ((hz * Hi + gy) * Wi + gx) * 8 + half * 4 : 0;
        inside |= ok ? (1u << i) : 0u;
        hzbits |= (hz & 1) ? (1u << i) : 0u;
        loff[i] = (p < NPIECE) ? ((hz * HY + hy) * HXP + hx) * 8 + half * 4 : -1;
    }

    // ---- MFMA lane roles (as deconvg_mfma_kernel with one N-tile): row r -> input voxel (ry, rx) of the
    // M-tile, g: dx = g >> 1, channels 4 (g & 1) .. + 3; column n -> (px, co) ----
    const int r = lane & 15, g = lane >> 4;
    int abase[MPW];
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int t = wave * MPW + i;
        const int tx = t % BX, ty = t / BX;
        abase[i] = ((2 * ty + (r >> 3)) * HXP + 8 * tx + (r & 7) + (g >> 1)) * 8 + (g & 1) * 4;
    }
    // scatter targets of the accumulators: element e of M-tile i is input voxel m = 4 g + e of the tile
    const int px = r >> 3, co = r & 7;
    const float bv = bias[co];
    int sbase[MPW][4];
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int t = wave * MPW + i;
        const int tx = t % BX, ty = t / BX;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 4 * g + e;
            const int ly = 2 * ty + (m >> 3), lx = 8 * tx + (m & 7);
            // input voxels beyond the volume scatter their (meaningless) outputs like all others: those
            // positions lie inside the tile and the skip pass below overwrites them with zeros
            sbase[i][e] = (2 * ly + 1) * RS + (co >> 2) * HS + (2 * lx + px + 1) * 4 + (co & 3);
        }
    }

    // ---- skip pieces: q = tid + NT j -> x = q & 31, half = (q >> 5) & 1, row = (q >> 6) & 15, plane = q >> 10 ----
    int srel[SKP], sl[SKP];
    unsigned sok = 0;
#pragma unroll
    for (int j = 0; j < SKP; ++j) {
        const int q = tid + NT * j;
        const int sx_ = q & 31, half = (q >> 5) & 1, row = (q >> 6) & 15, pz = j / (SKP / 2);
        const int gy = 2 * iy0 + row, gx = 2 * ix0 + sx_;
        const bool ok = gy < Ho && gx < Wo;
        srel[j] = ok ? (int)((((size_t)pz * Ho + gy) * Wo + gx) * 8 + half * 4) : 0;
        sok |= ok ? (1u << j) : 0u;
        sl[j] = pz * PS + (row + 1) * RS + half * HS + (sx_ + 1) * 4;
    }

    // ---- stencil roles: thread -> column lx, rows RPT tyq .. + RPT - 1 of the conv11 tile ----
    const int lx = tid & 31, tyq = tid >> 5;
    const int gxo = 2 * ix0 + lx;
    bool pvalid[RPT];
    size_t pout[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int ly = RPT * tyq + j, gy = 2 * iy0 + ly;
        pvalid[j] = gy < Ho && gxo < Wo && (ly >= 1 || iy0 == 0) && (ly <= OY - 2 || gy == Ho - 1) &&
                    (lx >= 1 || ix0 == 0) && (lx <= OX - 2 || gxo == Wo - 1);
        pout[j] = (size_t)gy * Wo + gxo;
    }
    const int cbase = (RPT * tyq) * RS + lx * 4;   // LDS row r = tile row r - 1, x index lx = tile column lx - 1

    f32x4 stg[PPT], sk[SKP];
    f32x4 acc[4][MPW];

    // Loaded values are not touched before they are consumed (masks are applied when they go to LDS): a
    // select right after the load would park the wave until the data has arrived.
    bool stg_z1 = true;
    auto load_a = [&](int i, int c) {   // input planes i, i+1 of channel chunk c -> registers
        const size_t base = ((size_t)c * Vin + (size_t)i * Hi * Wi) * 8;
        stg_z1 = i + 1 < Di;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const bool ok = ((inside >> k) & 1u) && (stg_z1 || !((hzbits >> k) & 1u));
            stg[k] = St<DT>::load4(x, ok ? base + grel[k] : (size_t)0);
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const bool ok = ((inside >> k) & 1u) && (stg_z1 || !((hzbits >> k) & 1u));
            if (loff[k] >= 0) *reinterpret_cast<f32x4*>(in_tile + loff[k]) = ok ? stg[k] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto load_skip = [&](int i, bool de, bool dodd) {   // conv0 values of planes 2i (de) and 2i+1 (dodd)
        const size_t base = (size_t)(2 * i) * HWo * 8;
#pragma unroll
        for (int j = 0; j < SKP; ++j) {
#if MVS_ABLATE == 24
            const bool ok = ((sok >> j) & 1u) && ((j / (SKP / 2)) ? dodd : de) && ZC < 0;
#else
            const bool ok = ((sok >> j) & 1u) && ((j / (SKP / 2)) ? dodd : de);
#endif
            sk[j] = St<DT>::load4(skip, ok ? base + srel[j] : (size_t)0);   // masked when added (sok)
        }
    };
    auto mfma_step = [&](int ks, const f32x4 (&a)[MPW], const f32x4& bq) {
        const int cls = deconv_step(ks).cls;
#pragma unroll
        for (int i = 0; i < MPW; ++i) acc[cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, bq.x, acc[cls][i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MPW; ++i) acc[cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, bq.y, acc[cls][i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MPW; ++i) acc[cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, bq.z, acc[cls][i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MPW; ++i) acc[cls][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, bq.w, acc[cls][i], 0, 0, 0);
    };
    auto mfma_chunk = [&](int c, bool de, bool dodd) {
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(bpan) + c * 9 * 64 + lane;
#if MVS_ABLATE == 23
        if (ZC >= 0) return;
#endif
        auto fetch = [&](int ks, f32x4 (&a)[MPW], f32x4& bq) {
            const DeconvStep st = deconv_step(ks);
            const int koff = (st.dz * HY + st.dy) * HXP * 8;
#pragma unroll
            for (int i = 0; i < MPW; ++i) a[i] = *reinterpret_cast<const f32x4*>(in_tile + abase[i] + koff);
            bq = bsrc[ks * 64];
        };
        if (de && dodd) {   // the usual step: all 9 k-steps, operands of k-step ks+1 requested before the MFMAs of ks
            f32x4 a[2][MPW], bq[2];
            fetch(0, a[0], bq[0]);
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                if (ks + 1 < 9) fetch(ks + 1, a[(ks + 1) & 1], bq[(ks + 1) & 1]);
                mfma_step(ks, a[ks & 1], bq[ks & 1]);
            }
        } else {            // a chunk's first / last step: one z parity only
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                if (deconv_step(ks).cls < 2 ? !de : !dodd) continue;   // wave-uniform
                f32x4 a[MPW], bq;
                fetch(ks, a, bq);
                mfma_step(ks, a, bq);
            }
        }
    };
    // one conv11 plane of the LDS tile into the three running sums it feeds (taps kz = 2, 1, 0); the
    // next column's 8 pieces are requested before the 72 packed FMAs of the current one
    auto stencil = [&](int plane, f32x2v (&k2)[RPT], f32x2v (&k1)[RPT], f32x2v (&k0)[RPT]) {
        const float* base = ct + plane * PS + cbase;
#if MVS_ABLATE == 21
        if (ZC >= 0) return;
#endif
        auto column = [&](int kx, f32x4 (&v)[RPT + 2][2]) {
#pragma unroll
            for (int rr = 0; rr < RPT + 2; ++rr)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    v[rr][hf] = *reinterpret_cast<const f32x4*>(base + rr * RS + hf * HS + kx * 4);
        };
        auto taps = [&](int kx, const f32x4 (&v)[RPT + 2][2]) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // uniform addresses: scalar loads, the weights are SGPR-pair operands of v_pk_fma_f32
                const float* w2 = pw + ((2 * 3 + ky) * 3 + kx) * 8;
                const float* w1 = pw + ((1 * 3 + ky) * 3 + kx) * 8;
                const float* w0 = pw + ((0 * 3 + ky) * 3 + kx) * 8;
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    const int rr = j + ky;
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2) {
                        const f32x4 q = v[rr][c2 >> 1];
                        const f32x2v d = (c2 & 1) ? (f32x2v){q.z, q.w} : (f32x2v){q.x, q.y};
                        k2[j] = __builtin_elementwise_fma(d, (f32x2v){w2[2 * c2], w2[2 * c2 + 1]}, k2[j]);
                        k1[j] = __builtin_elementwise_fma(d, (f32x2v){w1[2 * c2], w1[2 * c2 + 1]}, k1[j]);
                        k0[j] = __builtin_elementwise_fma(d, (f32x2v){w0[2 * c2], w0[2 * c2 + 1]}, k0[j]);
                    }
                }
            }
        };
        if constexpr (RPT == 2) {   // 2 waves per SIMD: the next column is requested before the current one's FMAs
            f32x4 va[RPT + 2][2], vb[RPT + 2][2];
            column(0, va);
            column(1, vb);
            taps(0, va);
            column(2, va);
            taps(1, vb);
            taps(2, va);
        } else {                    // 4 waves per SIMD and a 128-register budget: one column at a time
#pragma unroll 1
            for (int kx = 0; kx < 3; ++kx) {
                f32x4 va[RPT + 2][2];
                column(kx, va);
                taps(kx, va);
            }
        }
    };
    auto emit = [&](int oz, const f32x2v (&s)[RPT]) {
        float* dst = cost + (size_t)oz * HWo;
#pragma unroll
        for (int j = 0; j < RPT; ++j)
            if (pvalid[j]) dst[pout[j]] = s[j].x + s[j].y;
    };

    // ---- march ----
    const float pb = pbias[0];
    const f32x2v fresh = {pb, 0.0f};
    f32x2v A[RPT], B[RPT], C[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) A[j] = B[j] = C[j] = fresh;
    const int i_first = za > 0 ? za - 1 : 0;          // the chunk's first logit plane needs conv11 plane 2 za - 1
    const int i_last = zb < Di ? zb : Di - 1;         // ... its last one conv11 plane 2 zb

    for (int k = tid; k < C_FLOATS / 4; k += NT) reinterpret_cast<f32x4*>(ct)[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < 2 * 9 * 64; k += NT) reinterpret_cast<f32x4*>(bpan)[k] = reinterpret_cast<const f32x4*>(bp)[k];
    load_skip(i_first, i_first >= za, i_first < zb);
    load_a(i_first, 0);
    __syncthreads();

#pragma unroll 1
    for (int i = i_first; i <= i_last; ++i) {
        const bool de = i >= za, dodd = i < zb;       // even / odd conv11 plane of this step wanted
        store_a();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < MPW; ++k) acc[c][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        load_a(i, 1);
        mfma_chunk(0, de, dodd);
        __syncthreads();
        store_a();
        __syncthreads();
        mfma_chunk(1, de, dodd);
        if (i < i_last) load_a(i + 1, 0);   // travels during the epilogue and the stencil
        // ReLU(acc + bias) into the LDS tile (every element has exactly one writer) ...
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 2 ? !de : !dodd) continue;
#if MVS_ABLATE == 22
            if (ZC >= 0) continue;
#endif
            const int pz = c >> 1, py = c & 1;
#pragma unroll
            for (int k = 0; k < MPW; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) ct[pz * PS + py * RS + sbase[k][e]] = fmaxf(acc[c][k][e] + bv, 0.0f);
        }
        __syncthreads();
        // ... plus the skip values, 16 bytes at a time; zeros outside the volume
#pragma unroll
        for (int pz = 0; pz < 2; ++pz) {
            if (pz ? !dodd : !de) continue;
            constexpr int H = SKP / 2;
            f32x4 t[H];
#pragma unroll
            for (int j = 0; j < H; ++j) t[j] = *reinterpret_cast<const f32x4*>(ct + sl[H * pz + j]);
#pragma unroll
            for (int j = 0; j < H; ++j)
                *reinterpret_cast<f32x4*>(ct + sl[H * pz + j]) =
                    ((sok >> (H * pz + j)) & 1u) ? t[j] + sk[H * pz + j] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (i < i_last) load_skip(i + 1, true, i + 1 < zb);
        __syncthreads();
        // even plane E = 2i: completes logit 2i - 1
        if (de) {
            stencil(0, A, B, C);
            if (i > za) emit(2 * i - 1, A);
        }
        // odd plane O = 2i + 1: completes logit 2i
#pragma unroll
        for (int j = 0; j < RPT; ++j) A[j] = fresh;
        if (dodd) {
            stencil(1, B, C, A);
            if (de) emit(2 * i, B);
        }
        // (A, B, C) <- logits (2i+1, 2i+2, 2i+3)
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const f32x2v t = A[j];
            A[j] = C[j];
            B[j] = t;
            C[j] = fresh;
        }
        __syncthreads();
    }
    if (zb == Di) emit(Do - 1, A);   // the volume's last plane has no successor to complete it
    (void)Do;
}

