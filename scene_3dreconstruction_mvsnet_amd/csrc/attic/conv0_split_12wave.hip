// attic/conv0_split_12wave.hip -- NOT BUILT.  The twelve-wave form of the split-operand conv0 (round 4), kept as the
// measured answer to "would a second producer wave per SIMD overlap the staging with the MFMAs?".
//
// It is a fragment of csrc/conv0_split.hip (it uses that file's c43s constants, split3x, mfma_bf16, c43p_MAXT) and was
// selected there by MVS_CONV0_SPLIT=3.  Correct: tests/layer_check.py and tests/conv0_check.py passed with the bounds of
// the default kernel (cfg2 rel-L1 3.5e-7, heavy-tailed 0.098 of its bound) -- including the hand-counted
// `s_waitcnt vmcnt(12)` around the tied-operand inline-asm B loads (165 VGPRs, no spills, three waves per SIMD).
// Slower: 0.295 ms against 0.278 for the eight-wave kernel on the same box (bench 1,326 vs 1,377 maps/s): the single B
// set forces the B-piece-outermost order (72 instead of 36 A-fragment reads per step, one combo of read-ahead), and what
// the extra producer wave wins the consumer loses.
//
// ---------------------------------------------------------------------------------------------
// Third form (MVS_CONV0_SPLIT=3): TWELVE waves per block -- the persistent scheme above with eight producer waves.
// The ablations of the 8-wave kernel (profiles/r04_conv0_split.md) say its producers (0.156 ms alone) and consumers
// (~0.10) add up instead of overlapping: one producer wave per SIMD issues a dependent VALU stream at one instruction per
// 5-6 cycles and gets ~1.4 issue slots per 16-cycle MFMA of the consumer on its SIMD.  Two producer waves per SIMD (one
// staging column each) double that; three waves per SIMD leave 168 VGPRs, which the consumer meets with ONE set of B
// fragments loaded in place (inline asm, hand-counted vmcnt) and the A fragments one combo ahead.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(768) void conv0_w43q_kernel(
    const float* __restrict__ x,             // [4][D][H][W][8] fp32
    const unsigned short* __restrict__ bp,   // [4 chunks][6 t][3 ky][3 pieces][64 lanes][8] bf16 Toeplitz panel
    const float* __restrict__ bias,          // [8]
    float* __restrict__ y,                   // [D][H][W][8] fp32
    int D, int H, int W, int nb
    ) {
    using namespace c43s;
    __shared__ __attribute__((aligned(16))) unsigned short buf[2 * TILE_E];
    __shared__ __attribute__((aligned(16))) float ex[NT_PLANES * NPOS * EXS];
    __shared__ int org[c43p_MAXT][4];   // origins of this block's tiles (three runtime divisions each: once, not per step)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool consumer = __builtin_amdgcn_readfirstlane(wave) < 4;   // wave-uniform by construction; tell the compiler
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    const int G = gridDim.x;
    const int ntile = (nb - (int)blockIdx.x + G - 1) / G;   // this block's tiles: blockIdx.x + j * G
    const size_t HW8 = (size_t)H * W * 8, V8 = (size_t)D * HW8;
    // tile index -> origin; every XCD owns a band of tile rows and walks it x-fastest, then row, then z
    // (conv0_w43_mfma); G is a multiple of 8 whenever nb is, so a block's tiles stay on its XCD's band
    auto origin = [&](int idx, int& x0, int& y0, int& z0) {
        int bx, by, bz;
        if (nby % 8 == 0 && G % 8 == 0) {
            const int xcd = idx & 7, rows = nby >> 3;
            int i = idx >> 3;
            bx = i % nbx; i /= nbx;
            by = xcd * rows + i % rows;
            bz = i / rows;
        } else {
            int b = idx;
            bx = b % nbx; b /= nbx;
            by = b % nby;
            bz = b / nby;
        }
        x0 = bx * TX; y0 = by * TY; z0 = bz * TZ;
    };

    for (int j = tid; j < ntile; j += 768) {
        int x0, y0, z0;
        origin((int)blockIdx.x + j * G, x0, y0, z0);
        org[j][0] = x0; org[j][1] = y0; org[j][2] = z0;
    }
    __syncthreads();
    // ---------------- producer state ----------------
    const int ptid = max(tid - 256, 0);   // producer thread 0..511 (consumers: unused)
    unsigned boff[1][NT_PLANES];
    int loff[1];            // in 8-byte units inside one (piece, plane)
    int chy[1], chx[1], crel[1];   // the column's halo row / column and its float offset from the tile origin
#pragma unroll
    for (int i = 0; i < 1; ++i) {
        // threads beyond the 408 columns shadow the last column (same loads, same values to the same LDS address):
        // no branch inside the staging code
        const int col = min(ptid, NCOL - 1);
        loff[i] = col;    // (v * 2 + half) = col
        const int half = col & 1, v = col >> 1;
        chx[i] = v % HX - 1;
        chy[i] = v / HX - 1;
        crel[i] = (chy[i] * W + chx[i]) * 8 + half * 4;
    }
    // per tile: 32-bit arithmetic only (the launcher guarantees < 2^31 bytes per chunk plane); the first form of this
    // function (64-bit products, two divisions per column) took 1.4 k cycles of the producers' step
    auto set_tile = [&](int j) {   // j-th tile of this block
        const int x0 = org[j][0], y0 = org[j][1], z0 = org[j][2];
        const int hw8 = (int)HW8, tbase = (y0 * W + x0) * 8;
#pragma unroll
        for (int i = 0; i < 1; ++i) {
            const int gy = y0 + chy[i], gx = x0 + chx[i];
            const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const int cb = tbase + crel[i];
#pragma unroll
            for (int q = 0; q < NT_PLANES; ++q) {
                const int gz = z0 - 1 + q;   // wave-uniform
                boff[i][q] = (ok && (unsigned)gz < (unsigned)D) ? (unsigned)(gz * hw8 + cb) * 4u
                                                                  : 0x80000000u;   // beyond the descriptor: zeros
            }
        }
    };
    f32x4 stg[2][1][NT_PLANES];
    auto issue_loads = [&](int c, f32x4 (&st)[1][NT_PLANES]) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x + (size_t)c * V8), (short)0, (int)(V8 * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 1; ++i)
#pragma unroll
            for (int q = 0; q < NT_PLANES; ++q)
                st[i][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)boff[i][q], 0, 0));
    };
    // transform + split of one staged chunk (`st`) into `dstbuf`; LOAD = true: the raw loads of chunk `cn` (two steps
    // ahead) into `sn` are issued one per (column, plane) segment in between -- as one burst the 12 loads took
    // 1.8-2.5 k cycles to enter the CU's vector-memory queue, with the whole transform waiting behind them
    // (the staging sets are addressed by compile-time index, never passed as references: a reference parameter next
    // to the by-reference capture made hipcc keep `stg` in scratch memory)
    auto process = [&](auto set_tag, unsigned short* dstbuf, auto load_tag, int cn) {
        constexpr int SET = decltype(set_tag)::value;      // staged chunk: stg[SET]; loads go to stg[SET ^ 1]
        constexpr bool LOAD = decltype(load_tag)::value;
        su32x2* t8 = reinterpret_cast<su32x2*>(dstbuf);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x + (size_t)cn * V8), (short)0, (int)(V8 * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 1; ++i) {
            f32x4 d[NT_PLANES];
#pragma unroll
            for (int q = 0; q < NT_PLANES; ++q) d[q] = stg[SET][i][q];
            const f32x4 t1 = d[4] - 4.0f * d[2], t2 = d[3] - 4.0f * d[1];
            const f32x4 t3 = d[4] - d[2], t4 = 2.0f * (d[3] - d[1]);
            f32x4 u[NT_PLANES];
            u[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
            u[1] = t1 + t2;
            u[2] = t1 - t2;
            u[3] = t3 + t4;
            u[4] = t3 - t4;
            u[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
#pragma unroll
            for (int t = 0; t < NT_PLANES; ++t) {
                if constexpr (LOAD)
                    stg[SET ^ 1][i][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)boff[i][t], 0, 0));
                su32x2 p1, p2, p3;
#if MVS_ABLATE == 74   // diagnostic: no split arithmetic
                p1 = p2 = p3 = (su32x2){__float_as_uint(u[t].x), __float_as_uint(u[t].z)};
#else
                split3x(u[t], p1, p2, p3);
#endif
                su32x2* dst = t8 + t * (PLANE_E / 4) + loff[i];
                dst[0] = p1;
                dst[PIECE_E / 4] = p2;
                dst[2 * (PIECE_E / 4)] = p3;
                // one plane at a time, in this order (also: interleaving the six splits of both columns for ILP
                // costs more registers than the two staging sets leave)
#if MVS_ABLATE != 56
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
    };
    // output transform + stores of a finished tile: thread -> (position, 4 channels).  Raw buffer stores: `live` =
    // false (a block's first step has no finished tile) or a position outside the volume puts the offset beyond the
    // descriptor and the hardware drops the store -- no branch around a VMEM instruction, so hipcc keeps COUNTED
    // vmcnt waits for the prefetched activations (a conditional store or load makes it wait for the loads just issued)
    const f32x4 obias = *reinterpret_cast<const f32x4*>(bias + (ptid & 1) * 4);
    const bool owave = ptid < 256;   // the output phase is run by producer waves 4-7
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, (int)(V8 * 4), 0x00020000);
    auto output = [&](int j, bool live) {   // j-th tile of this block
        const int x0 = org[j][0], y0 = org[j][1], z0 = org[j][2];
        const int pos = (ptid & 255) >> 1, ch = ptid & 1;
        const int gy = y0 + (pos >> 5), gx = x0 + (pos & 31);
        const bool ok = live && owave && gy < H && gx < W;
        f32x4 M[NT_PLANES];
#pragma unroll
        for (int q = 0; q < NT_PLANES; ++q) M[q] = *reinterpret_cast<const f32x4*>(ex + (q * NPOS + pos) * EXS + ch * 4);
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 s12 = M[1] + M[2], d12 = M[1] - M[2], s34 = M[3] + M[4], d34 = M[3] - M[4];
        f32x4 o[TZ];
        o[0] = (M[0] + s12) + s34;
        o[1] = d12 + 2.0f * d34;
        o[2] = s12 + 4.0f * s34;
        o[3] = (d12 + 8.0f * d34) + M[5];
        const unsigned base = (unsigned)((((size_t)z0 * H + gy) * W + gx) * 8 + ch * 4) * 4u;
#pragma unroll
        for (int q = 0; q < TZ; ++q) {
            const f32x4 v = __builtin_elementwise_max(o[q] + obias, zero);
            const unsigned off = (ok && z0 + q < D) ? base + (unsigned)q * (unsigned)(HW8 * 4) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4, v), yrs, (int)off, 0, 0);
        }
    };

    // ---------------- consumer state ----------------
    // work units of a consumer wave: u = 3 wave + i -> (plane u >> 1, rows 2 (u & 1) .. + 1); lane (r, g): halo x = 2 r + g
    const int r = lane & 15, g = lane >> 4;
    const int cw = wave & 3, u0 = 3 * cw, pA = u0 >> 1;
    int aoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = u0 + i, pl = u >> 1, mg = u & 1;
        aoff[i] = pl * PLANE_E + ((2 * mg) * HX + 2 * r + g) * 8;
    }
    // B fragments by raw buffer loads: ONE vector offset (the lane's 16 bytes + the wave's first plane), the fragment
    // in the scalar offset -- plain pointers made hipcc materialise (and spill) 35 64-bit addresses
    // ONE set (72 VGPRs at the 168-VGPR budget of three waves per SIMD): the loads are inline asm with the destination
    // TIED to the fragment's register -- hipcc will not load into a register an earlier MFMA has read, it renames (the
    // first attempt at this build allocated a second set that way and spilled 163 registers) -- and, being invisible to
    // the compiler's waitcnt pass, are waited for by hand: the consumer's only VMEM instructions are these loads, six
    // per B-piece group in the fixed order q = 0, 1, 2, so `s_waitcnt vmcnt(12)` before a group's first MFMA is exact
    su32x4 B[2][3][3];
    const unsigned long long bpa = reinterpret_cast<unsigned long long>(bp);
    su32x4 bdesc = {(unsigned)bpa, (unsigned)(bpa >> 32) & 0xFFFFu, (unsigned)(4 * NT_PLANES * 9 * 1024), 0x00020000u};
#pragma unroll
    for (int j = 0; j < 4; ++j) bdesc[j] = __builtin_amdgcn_readfirstlane(bdesc[j]);
    const unsigned bvoff = (unsigned)lane * 16u + (unsigned)pA * 9u * 1024u;
    auto load_group = [&](unsigned chunk_off, int qq) {   // the six fragments of B piece qq, IN PLACE
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const unsigned soff = chunk_off + (unsigned)((pl * 3 + ky) * 3 + qq) * 1024u;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(B[pl][ky][qq]) : "v"(bvoff), "s"(bdesc), "s"(soff));
            }
    };
    auto wait_group = [&](int qq) {   // all but the 12 youngest loads done = piece qq of THIS step has arrived
        asm volatile("s_waitcnt vmcnt(12)"
                     : "+v"(B[0][0][qq]), "+v"(B[0][1][qq]), "+v"(B[0][2][qq]), "+v"(B[1][0][qq]), "+v"(B[1][1][qq]), "+v"(B[1][2][qq]));
    };
    f32x4 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // acc[i][rr][e] = (plane, row 2 mg + rr, x = 2 (4 g + e) + jj, channel co), n = lane & 15 = (jj, co)
    auto exchange = [&]() {
        const int n = lane & 15, jj = n >> 3, co = n & 7;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = u0 + i, pl = u >> 1, mg = u & 1;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ex[(pl * NPOS + (2 * mg + rr) * TX + 2 * (4 * g + e) + jj) * EXS + co] = acc[i][rr][e];
                    acc[i][rr][e] = 0.0f;
                }
        }
    };

    // The two roles run DISJOINT loops (each with the same number of block barriers: one per chunk-step plus the one
    // after the prologue), so that the register allocator overlays the producers' staging sets with the consumers'
    // B fragments and accumulators instead of keeping both alive.  A loop iteration is one tile = four chunk-steps
    // (c is a compile-time constant, PB = c & 1 = LDS buffer / register set of the step).  EVERY vector-memory
    // instruction is unconditional -- the loads beyond a block's last step re-read its last tile, the B fragments
    // wrap around -- because one branch around a VMEM instruction makes hipcc wait for the loads it has just issued
    // (first build: `if (s + 2 < S) issue_loads` -> s_waitcnt vmcnt(8) / (1) / (0) right after the 12 new loads:
    // every step paid a full L2-miss latency, 6.0 k cycles per step with no unit busier than 40 %).
    if (consumer) {
        auto cstep_par = [&](auto c_tag, auto par_tag) {
            constexpr int c = decltype(c_tag)::value, PB = c & 1, PAR = decltype(par_tag)::value;
            constexpr unsigned bnext = (unsigned)(((c + 1) & 3) * NT_PLANES * 9) * 1024u;   // next step's chunk (wraps around)
            const unsigned short* tile = buf + PB * TILE_E;
            // B pieces outermost: (A piece p, B piece q) = q 0: p 0, 1, 2; q 1: p 0, 1; q 2: p 0 -- a piece's six
            // fragments are dead after its group and the next step's are requested into the same registers; 18 combos
            // of (q, p, unit), each 4 A fragments (read one combo ahead) -> 6 MFMAs on two alternating accumulators
            auto qp = [](int n, int& qq, int& p, int& i) {
                const int gg = n / 3;
                i = n % 3;
                qq = gg < 3 ? 0 : (gg < 5 ? 1 : 2);
                p = gg < 3 ? gg : (gg < 5 ? gg - 3 : 0);
            };
            su32x4 a[2][4];
            auto request = [&](int n) {
                int qq, pn, in;
                qp(n, qq, pn, in);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    a[n & 1][j] = *reinterpret_cast<const su32x4*>(tile + pn * PIECE_E + aoff[in] + j * HX * 8);
            };
            request(0);
#pragma unroll
            for (int n = 0; n < 18; ++n) {
                int qq, p, i;
                qp(n, qq, p, i);
                if (n + 1 < 18) request(n + 1);
                if (n == 0 || n == 9 || n == 15) wait_group(qq);
                const int pl = i == 0 ? 0 : (i == 2 ? 1 : PAR);   // even waves: planes (p, p, p+1), odd: (p, p+1, p+1)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
                        acc[i][rr] = mfma_bf16(a[n & 1][rr + ky], B[pl][ky][qq], acc[i][rr]);
                if (n == 8 || n == 14 || n == 17) load_group(bnext, qq);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c == 3) exchange();   // the producers read it during the next step (after the barrier)
        };
        auto cstep = [&](auto c_tag) {
            if (cw & 1) cstep_par(c_tag, std::integral_constant<int, 1>{});
            else cstep_par(c_tag, std::integral_constant<int, 0>{});
        };
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) B[pl][ky][qq] = (su32x4){0u, 0u, 0u, 0u};
        load_group(0u, 0);
        load_group(0u, 1);
        load_group(0u, 2);
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < ntile; ++k) {
            cstep(std::integral_constant<int, 0>{});
            __syncthreads();
            cstep(std::integral_constant<int, 1>{});
            __syncthreads();
            cstep(std::integral_constant<int, 2>{});
            __syncthreads();
            cstep(std::integral_constant<int, 3>{});
            __syncthreads();
        }
    } else {
#if MVS_ABLATE != 57
        __builtin_amdgcn_s_setprio(2);   // the producers are the critical path: their VALU wins the issue arbitration
#endif
        // step (k, c): loads of step + 2, output of tile k - 1 (c == 0), transform + split of step + 1
        auto pstep = [&](int k, auto c_tag) {
            constexpr int c = decltype(c_tag)::value, PB = c & 1;
            if (c == 2 && k + 1 < ntile) set_tile(k + 1);   // the loads below start the next tile
            if (c == 1) output(k > 0 ? k - 1 : 0, k > 0);   // waves 4-7 store, waves 8-11 run it with dropped stores
            // (after the last step: into a buffer nobody reads, loads that re-read the last tile)
            process(std::integral_constant<int, PB ^ 1>{}, buf + (PB ^ 1) * TILE_E, std::true_type{}, (c + 2) & 3);
        };
        // prologue: step 0 staged, step 1 in flight
        set_tile(0);
        issue_loads(0, stg[0]);
        issue_loads(1, stg[1]);
        process(std::integral_constant<int, 0>{}, buf, std::false_type{}, 0);
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < ntile; ++k) {
            pstep(k, std::integral_constant<int, 0>{});
            __syncthreads();
            pstep(k, std::integral_constant<int, 1>{});
            __syncthreads();
            pstep(k, std::integral_constant<int, 2>{});
            __syncthreads();
            pstep(k, std::integral_constant<int, 3>{});
            __syncthreads();
        }
        output(ntile - 1, true);
    }
}


