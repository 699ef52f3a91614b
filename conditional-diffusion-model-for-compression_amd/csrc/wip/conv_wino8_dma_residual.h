// WORK IN PROGRESS -- NOT COMPILED INTO libcdx.so.
// 8-wave Winograd body with the residual tile fetched by LDS-DMA behind the first halo loads and added in the epilogue
// (pre-transpose, one channel per lane; GroupNorm sums as one double pair per lane).  In-process it removes the 8 %
// "residual phase" of a four-chunk launch, BUT the build is flaky: 2-15 % of launches (tools/dbg3.py) return 16 wrong
// values -- one wave of ntile 2 or 3, lanes 15/31/47/63, registers rr = 0..3, position 1 -- also with __syncthreads()
// barriers, without any DMA code (OPT 256) and without inline-asm waits.  The committed kernel (conv_wino.h) shows
// 0 / 600 under the same probe.  Root cause not found; do not ship without it.
template <class C, int WH>
__device__ __forceinline__ void conv_wino8_body(const ConvParams& p, float* lds) {
    constexpr int KC = C::KC, PS = C::PS, RS = C::RS, GPC = C::GPC, PF = C::PF;
    constexpr int NP8 = 4;                                   // staging passes of 64 pixel slots
    constexpr int wh = WH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1;

    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x;
    bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int b = bx / p.tiles_y;
    const int oy0 = ty * C::TH, ox0 = tx * C::TW;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;

    // Residual tile (128 px x 128 ch): fetched by LDS-DMA (no VGPRs, no waiting) right behind the first halo loads and
    // added in the epilogue -- measured before this change: the 32 residual loads ahead of the first MFMA cost 8 % of a
    // four-chunk launch (every CU fetching at once, matrix pipe idle).  Layout: 64 pixel pairs (one DMA instruction each:
    // lane = (pixel of the pair, channel quad)) at a stride of 1 KiB + 32 B, so the epilogue's reads -- lane = channel,
    // the two lane halves 4 pairs apart -- fall into different banks.
    constexpr int RPAIR = 264;
    constexpr int LDS_MAIN = 2 * C::BUF_FLOATS > 16384 ? 2 * C::BUF_FLOATS : 16384;
    float* const lds_res = lds + LDS_MAIN;
    const bool use_res = p.residual != nullptr && !(C::OPT & 256);      // OPT 256 (ablation): no residual
    // Raw barrier: with an LDS-DMA possibly in flight hipcc turns __syncthreads() into vmcnt(0) + s_barrier, which
    // would drain the weight ring and the halo loads at every chunk.  LDS accesses are ordered by the explicit
    // lgkmcnt(0); the DMA is drained once (vmcnt(0)) before the barrier that precedes the epilogue.
    auto wg_barrier = [&]() {
        if constexpr (C::OPT & 512) __syncthreads();      // debugging aid
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto dma_residual = [&]() {      // wave w: 16 pixels of tile row w >> 1, two per instruction
        const int row = wave >> 1, col0 = (wave & 1) * 16;
        int ch = blockIdx.y * 128 + (lane & 31) * 4;
        if (ch + 4 > p.Cout) ch = 0;                                   // lanes past cout: any valid address (never used)
        if (oy0 + C::TH <= p.Hout && ox0 + C::TW <= p.Wout) {          // whole tile inside the image: scalar offsets only
            const __amdgpu_buffer_rsrc_t rr = buf_rsrc(p.residual + (((size_t)b * p.Hout + oy0 + row) * p.Wout + ox0 + col0) * p.Cout);
            const unsigned voff = ((unsigned)(lane >> 5) * (unsigned)p.Cout + (unsigned)ch) * 4u;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (__attribute__((address_space(3))) void*)(lds_res + (row * 16 + (col0 >> 1) + i) * RPAIR),
                                                         16, voff, (unsigned)(2 * i) * (unsigned)p.Cout * 4u, 0, 0);
        } else {
            const int oy = min(oy0 + row, p.Hout - 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ox = min(ox0 + col0 + 2 * i + (lane >> 5), p.Wout - 1);      // clamped: masked at store
                const float* g = p.residual + (((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + ch;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lds_res + (row * 16 + (col0 >> 1) + i) * RPAIR), 16, 0, 0);
            }
        }
    };

    // ---- halo loader: 64 pixel slots x 8 channel quads.  Passes 0..2 = halo rows 2i, 2i+1 x columns 0..31, pass 3 =
    // the two leftover columns of all six rows (12 slots) ----
    static_assert(C::HH == 6 && C::HW == 34, "loader geometry is written for the 4 x 32 tile");
    const int q = tid & 7, pl = tid >> 3;                    // pl in 0..63
    const int prow = pl >> 5, pcol = pl & 31;
    const int ixa = ix0 + pcol;
    const bool colok = ixa >= 0 && ixa < Wv;
    const int colx = colok ? (ixa >> p.ups) : 0;
    const int iy3 = iy0 + (pl >> 1), ix3 = ix0 + 32 + (pl & 1);
    const bool ok3 = pl < 12 && iy3 >= 0 && iy3 < Hv && ix3 >= 0 && ix3 < Wv;
    const int soff3 = ok3 ? ((b * p.Hin + (iy3 >> p.ups)) * p.Win + (ix3 >> p.ups)) : 0;
    const int wbase = prow * RS + pcol * PS + q * 4;          // + 2*i*RS
    const int wbase3 = (pl >> 1) * RS + (32 + (pl & 1)) * PS + q * 4;
    auto row_src = [&](int i) -> int {                        // per-thread (2 rows per pass): source row base or -1
        const int iy = iy0 + 2 * i + prow;
        return (iy >= 0 && iy < Hv) ? (b * p.Hin + (iy >> p.ups)) * p.Win : -1;
    };
    f32x4 pre[NP8];
    f32x4 gsc, gsh;
    bool cvalid;
    auto issue_loads = [&](int chunk) {
        const int s = chunk >= p.nchunk0;
        const int cl = (s ? chunk - p.nchunk0 : chunk) * KC + q * 4;
        const int cs = p.csrc[s];
        cvalid = cl < cs;
        const float* __restrict__ base = p.src[s] + (cvalid ? cl : 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rs = row_src(i);
            pre[i] = *reinterpret_cast<const f32x4*>(base + (size_t)((rs < 0 ? 0 : rs) + colx) * cs);
        }
        pre[3] = *reinterpret_cast<const f32x4*>(base + (size_t)soff3 * cs);
        if (p.gn) {
            const int cg = cvalid ? (s ? p.csrc[0] : 0) + cl : 0;
            gsc = *reinterpret_cast<const f32x4*>(p.gscale + (size_t)b * p.ctot + cg);
            gsh = *reinterpret_cast<const f32x4*>(p.gshift + (size_t)b * p.ctot + cg);
        }
    };
    auto write_pass = [&](float* buf, int i) {
        f32x4 v = pre[i];
        const bool ok = cvalid && (i < 3 ? (colok && row_src(i) >= 0) : ok3);
        if (p.gn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], gsc[e], gsh[e]);
        }
        if (p.silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f<false>(v[e]);
        }
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < 3) *reinterpret_cast<f32x4*>(&buf[wbase + 2 * i * RS]) = v;
        else if (pl < 12) *reinterpret_cast<f32x4*>(&buf[wbase3]) = v;
    };

    // ---- operand addressing: lane = (Winograd tile li, channel half lh); this wave reads patch rows wh..wh+2 ----
    const int li = lane & 31, lh = lane >> 5;
    const int wty = li >> 4, wtx = li & 15;
    const int a_base = (2 * wty + wh) * RS + (2 * wtx) * PS + lh * 4;
    const int ntile = blockIdx.y * 4 + wn;
    const bool nvalid = ntile * 32 < p.Cout;
    // packed weights [ntile][chunk][s][e][xiq][lane][4]: this wave uses xiq = 2*wh, 2*wh + 1
    const float* __restrict__ wp = p.w + ((size_t)(nvalid ? ntile : 0) * p.nchunks) * 16384 + wh * 512;
    const unsigned lane4 = lane * 4;
    const int n = ntile * 32 + li;
    const bool nok = nvalid && n < p.Cout;
    const bool full_tile = oy0 + C::TH <= p.Hout && ox0 + C::TW <= p.Wout;      // wave-uniform

    // acc[2*x + j]: x = local V row (global row 2*wh + x), j = column
    f32x16 acc[8];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    if (nok) {
        // bias + temb enter through the accumulator init: with Y = A^T M A, M[0][0] = a, M[0][3] = -a, M[3][0] = -a,
        // M[3][3] = a adds exactly a to the 2x2 output.  Half 0 owns row 0 of M, half 1 row 3.
        float add = p.bias ? p.bias[n] : 0.f;
        if (p.temb) add += p.temb[(size_t)b * p.temb_ld + n];
        // (accumulator indices must be compile-time: a runtime-indexed register array is placed in scratch memory)
        float addv = add;
        if constexpr (C::OPT & 1024) asm volatile("" : "+v"(addv));      // debugging aid: opaque value
        if constexpr (wh == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[0][r] = addv;
                acc[3][r] = -addv;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[4][r] = -addv;
                acc[7][r] = addv;
            }
        }
    }

    // weight ring: this wave consumes 2 fragments per group g = (s, e): f = 2*g + k, k = local xiq
    constexpr int RF = 2 * PF;
    f32x4 ring[RF];
    auto foff = [](int f) { return (f >> 1) * 1024 + (f & 1) * 256; };     // float offset of fragment f inside a chunk
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(wp);                       // fragment address = scalar offset + lane * 16 B
    const unsigned lane16 = lane * 16;
#pragma unroll
    for (int f = 0; f < RF; ++f) ring[f] = buf_load4(wrs, lane16, foff(f) * 4u);

    using f32x2 = __attribute__((ext_vector_type(2))) float;
    auto load_half = [&](const float* buf, int hh, f32x2 (&dst)[12]) {      // patch rows wh..wh+2, channels of half hh
        int ab = a_base;
        asm volatile("" : "+v"(ab));
        __builtin_assume((ab & 1) == 0);
        const int coff = (hh >> 1) * 8 + (hh & 1) * 2;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
                dst[a * 4 + bb] = *reinterpret_cast<const f32x2*>(&buf[ab + a * RS + bb * PS + coff]);
    };
    // the two V rows of this wave, for BOTH channels of a half at once (the channel pair a ds_read_b64 delivers is a
    // 64-bit register pair, so every add of B^T d B is one v_pk_add_f32 = two results per VALU issue):
    // rows (0,1) from patch rows (0,1,2); rows (2,3) from patch rows (1,2,3).  Same operation tree as the scalar form.
    // (OPT 32: the adds as inline-asm v_pk_add_f32, which hipcc's post-RA peephole cannot split back into two
    // scalar adds when they sit in the shadow of an MFMA.)
    auto padd = [](f32x2 a, f32x2 b) -> f32x2 {
        if constexpr (C::OPT & 32) {
            f32x2 r;
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            return r;
        } else return a + b;
    };
    auto psub = [](f32x2 a, f32x2 b) -> f32x2 {
        if constexpr (C::OPT & 32) {
            f32x2 r;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            return r;
        } else return a - b;
    };
    auto transform2 = [&](const f32x2 (&d)[12], f32x2 (&v)[8]) {
        f32x2 r0[4], r1[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x2 u0 = d[0 + bb], u1 = d[4 + bb], u2 = d[8 + bb];
            if constexpr (wh == 0) {   // t0 = u0, t1 = u1, t2 = u2:  r[0] = t0 - t2, r[1] = t1 + t2
                r0[bb] = psub(u0, u2);
                r1[bb] = padd(u1, u2);
            } else {                 // t1 = u0, t2 = u1, t3 = u2:  r[2] = t2 - t1, r[3] = t1 - t3
                r0[bb] = psub(u1, u0);
                r1[bb] = psub(u0, u2);
            }
        }
        v[0] = psub(r0[0], r0[2]); v[1] = padd(r0[1], r0[2]); v[2] = psub(r0[2], r0[1]); v[3] = psub(r0[1], r0[3]);
        v[4] = psub(r1[0], r1[2]); v[5] = padd(r1[1], r1[2]); v[6] = psub(r1[2], r1[1]); v[7] = psub(r1[1], r1[3]);
    };

    // One chunk = 8 halves (channel pairs per lane) x 2 groups of 8 MFMAs.  Even group of half h: issue the LDS patch
    // reads of half h+1 (single patch buffer: it was consumed one group earlier); odd group: transform them into the
    // other operand set.  The second wave of the SIMD covers the LDS latency; a group is 512 MFMA cycles.
    auto chunk_body = [&](const int chunk, const bool more) __attribute__((always_inline)) {
        const float* cur = lds + (chunk & 1) * C::BUF_FLOATS;
        float* nxt = lds + ((chunk + 1) & 1) * C::BUF_FLOATS;
        const unsigned wcb = (unsigned)chunk * 65536u;      // byte offset of this chunk's fragments
        f32x2 dh[12];
        f32x2 vv[2][8];
        auto make_operands = [&](f32x2 (&v)[8]) {
            if constexpr (C::OPT & 8) {          // OPT 8 (ablation): no transform adds
#pragma unroll
                for (int x = 0; x < 8; ++x) v[x] = dh[x];
            } else transform2(dh, v);
        };
        load_half(cur, 0, dh);
        make_operands(vv[0]);
#pragma unroll
        for (int g = 0; g < GPC; ++g) {
            const int hh = g >> 1;
            if (hh + 1 < 8) {
                if ((g & 1) == 0) {
                    if (!(C::OPT & 2) || hh == 0) load_half(cur, hh + 1, dh);      // OPT 2 (ablation): one more half-load per chunk only
                } else make_operands(vv[(hh + 1) & 1]);
            }
            if (more && !(C::OPT & 4)) {      // OPT 4: no staging
                constexpr int G0 = GPC - NP8 - 1;
                if (g >= G0 && g < G0 + NP8) write_pass(nxt, g - G0);
                if (g == G0 + NP8 && chunk + 2 < p.nchunks) issue_loads(chunk + 2);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int f = g * 2 + k;
                const f32x4 bq = ring[f % RF];
                if constexpr (!(C::OPT & 1))     // OPT 1: no weight refills
                    ring[f % RF] = buf_load4(wrs, lane16, wcb + (f + RF < 32 ? foff(f + RF) : 16384 + foff(f + RF - 32)) * 4u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[k * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[hh & 1][k * 4 + j][g & 1], bq[j], acc[k * 4 + j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- pipeline ----
    issue_loads(0);
    if (use_res) dma_residual();          // behind the first halo loads in the (in-order) vmcnt queue, ahead of nothing urgent
#pragma unroll
    for (int i = 0; i < NP8; ++i) write_pass(lds, i);
    if (p.nchunks > 1) issue_loads(1);
    wg_barrier();

    if (!nvalid) {
        for (int chunk = 0; chunk < p.nchunks; ++chunk) {
            if (chunk + 1 < p.nchunks) {
#pragma unroll
                for (int i = 0; i < NP8; ++i) write_pass(lds + ((chunk + 1) & 1) * C::BUF_FLOATS, i);
                if (chunk + 2 < p.nchunks) issue_loads(chunk + 2);
            }
            if (use_res && chunk + 1 == p.nchunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the residual tile has landed
            wg_barrier();
        }
        if constexpr (!(C::OPT & 16)) wg_barrier();     // matches the exchange barrier below
        return;
    }
    for (int chunk = 0; chunk + 1 < p.nchunks; ++chunk) {
        chunk_body(chunk, true);
        wg_barrier();
    }
    chunk_body(p.nchunks - 1, false);
    if constexpr (C::OPT & 16) {   // OPT 16 (timing ablation): no output transform / exchange / stores
        float keep = 0.f;
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[x][r];
        if (keep == 123.456f) p.out[0] = keep;
        return;
    }
    if (use_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the residual tile has landed (long ago)
    wg_barrier();            // every wave is done with the halo buffers: they become the exchange image

    // ---- partial output transform: this half's rows of M -> partial Y (linear), then swap halves through LDS ----
    // tmp[0][j] = M0j + M1j + M2j, tmp[1][j] = M1j - M2j - M3j.  half 0 (rows 0,1): (M0j + M1j, M1j); half 1 (rows 2,3): (M2j, -M2j - M3j)
    // exchange image: xch[wave][k = 0..31][lane]; registers 0..7 are finished by half 0, 8..15 by half 1.
    float* xch = lds;
    float mine[32];      // the 8 registers this wave finishes: partial y[4] each
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float t[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m0 = acc[0 + j][r], m1 = acc[4 + j][r];          // local rows 0, 1
            if constexpr (wh == 0) {
                t[0][j] = m0 + m1;
                t[1][j] = m1;
            } else {
                t[0][j] = m0;
                t[1][j] = -m0 - m1;
            }
        }
        float y[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            y[a * 2 + 0] = t[a][0] + t[a][1] + t[a][2];
            y[a * 2 + 1] = t[a][1] - t[a][2] - t[a][3];
        }
        const bool keep = (r >> 3) == wh;       // compile-time: registers 0..7 are finished by half 0, 8..15 by half 1
        const int rr = r & 7;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (keep) mine[rr * 4 + k] = y[k];
            else xch[((wave * 32) + rr * 4 + k) * 64 + lane] = y[k];     // the partner finishes this register
        }
    }
    wg_barrier();
    {
        const int partner = wave ^ 1;
#pragma unroll
        for (int k = 0; k < 32; ++k) mine[k] += xch[((partner * 32) + k) * 64 + lane];
    }
    // Here a lane still owns ONE channel (n) at 32 pixels: residual add and GroupNorm sums happen before the quad
    // transposes -- 32 conflict-free ds_read_b32 and one double pair per lane instead of four.
    // register r = 8*wh + rr -> tile (rr & 3) + 8*(rr >> 2) + 4*lh of tile row wh; value k -> pixel (2*wh + (k >> 1), 2*tilecol + (k & 1))
    // (conditions hoisted: one straight-line loop per case, no per-value branches or EXEC juggling)
    double s1 = 0.0, s2 = 0.0;
    if (use_res) {
        const int rbase = (wn * 32 + li) + (4 * lh) * RPAIR;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                mine[rr * 4 + k] += lds_res[rbase + ((2 * wh + (k >> 1)) * 16 + (rr & 3) + 8 * (rr >> 2)) * RPAIR + (k & 1) * 128];
    }
    if (p.stats && full_tile) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const double dv = (double)mine[k];
            s1 += dv;
            s2 = fma(dv, dv, s2);
        }
    } else if (p.stats) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool inside = oy0 + 2 * wh + (k >> 1) < p.Hout && ox0 + 2 * ((rr & 3) + 8 * (rr >> 2) + 4 * lh) + (k & 1) < p.Wout;
                const double dv = inside ? (double)mine[rr * 4 + k] : 0.0;      // branch-free: a zero adds nothing to either sum
                s1 += dv;
                s2 = fma(dv, dv, s2);
            }
    }

    // ---- packed stores (quad transposes) of this wave's 8 tile-registers x 4 positions ----
    int eoy0 = oy0, eox0 = ox0, elh = lh;
    asm volatile("" : "+s"(eoy0), "+s"(eox0), "+v"(elh));
    const int q4 = li & 3;
    const int cq = ntile * 32 + (li & ~3);
    const bool quad_ok = cq < p.Cout;
    const bool vec_ok = (p.out_ld & 3) == 0 && cq + 4 <= p.out_ld;
    const bool fast_store = full_tile && vec_ok && ntile * 32 + 32 <= p.Cout;      // wave-uniform
    if (fast_store) {
        // scalar base per (kk, pos) + one per-lane offset: tile = 8*(2*wh + kk) + q4 + 4*lh -> row wh, column 8*kk + q4 + 4*lh
        const __amdgpu_buffer_rsrc_t ro = buf_rsrc(p.out + (((size_t)b * p.Hout + eoy0 + 2 * wh) * p.Wout + eox0) * p.out_ld);
        const unsigned voff = ((unsigned)(2 * (q4 + 4 * elh)) * (unsigned)p.out_ld + (unsigned)cq) * 4u;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                float x[4] = {mine[(4 * kk + 0) * 4 + pos], mine[(4 * kk + 1) * 4 + pos], mine[(4 * kk + 2) * 4 + pos], mine[(4 * kk + 3) * 4 + pos]};
                quad_transpose(x, q4);
                const unsigned pix = (unsigned)(pos >> 1) * (unsigned)p.Wout + (unsigned)(16 * kk + (pos & 1));
                buf_store4(ro, voff, pix * (unsigned)p.out_ld * 4u, f32x4{x[0], x[1], x[2], x[3]});
            }
        }
    } else
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {            // blocks of 4 registers: global registers 8*wh + 4*kk + i
        const int tile = 8 * (2 * wh + kk) + q4 + 4 * elh;
        const int oy = eoy0 + 2 * (tile >> 4), ox = eox0 + 2 * (tile & 15);
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            float x[4] = {mine[(4 * kk + 0) * 4 + pos], mine[(4 * kk + 1) * 4 + pos], mine[(4 * kk + 2) * 4 + pos], mine[(4 * kk + 3) * 4 + pos]};
            quad_transpose(x, q4);
            const int py = oy + (pos >> 1), px = ox + (pos & 1);
            if (quad_ok && py < p.Hout && px < p.Wout) {
                const size_t pix = ((size_t)b * p.Hout + py) * p.Wout + px;
                if (vec_ok) *reinterpret_cast<f32x4*>(p.out + pix * p.out_ld + cq) = f32x4{x[0], x[1], x[2], x[3]};
                else
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (cq + c < p.Cout) p.out[pix * p.out_ld + cq + c] = x[c];
            }
        }
    }
    if (p.stats) {
        // slot = (tile, half): two slots per spatial tile (cdx_conv_stats_slots accounts for it)
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lh == 0 && nok) {
            const int slot = (ty * p.tiles_x + tx) * 2 + wh;
            const int nslots = p.tiles_y * p.tiles_x * 2;
            double* o = p.stats + (((size_t)b * nslots + slot) * p.Cout + n) * 2;
            o[0] = s1;
            o[1] = s2;
        }
    }
}

template <class C>
__global__ __launch_bounds__(512, 2) void conv_wino8_kernel(const ConvParams p) {
    // halo double buffer / exchange image (64 KiB) + residual tile (64 pixel pairs x 1056 B)
    constexpr int LDS_FLOATS = (2 * C::BUF_FLOATS > 16384 ? 2 * C::BUF_FLOATS : 16384) + 64 * 264;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    // wave-uniform: both arms execute the same number of barriers
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) conv_wino8_body<C, 1>(p, lds);
    else conv_wino8_body<C, 0>(p, lds);
}

