// WORK IN PROGRESS -- NOT COMPILED INTO libcdx.so.
// Persistent form of the 8-wave Winograd kernel (one workgroup per CU walks tiles; next tile's halo / residual fetched
// under the current tile; stores drain under the next tile).  State when parked: compiles to 256 VGPRs with ~170 spilled
// VGPRs and ~80 spilled SGPRs (loader state, ring and in-flight loads live across the epilogue), and its parity failure
// (16 wrong values per bad launch) was the store-data hazard described in DESIGN.md section 8: its epilogue interleaves
// quad transposes and 16-byte stores, so the rule "registers read by a store are not written again" must be applied
// before it is revived.  Expected gain if the spills are removed: the tile turnaround only (~3-5 % of the kernel); the
// LDS-DMA residual path measured no gain on the non-persistent kernel.
// ------------------------------------------------------------------------------------------------------------
// Persistent 8-wave form (experiment).  Measured on the non-persistent 8-wave kernel:
// of a 47 us four-chunk tile, ~4 us is the residual tile being fetched before the first MFMA and ~5 us the output
// stores + workgroup turnaround -- memory phases during which the CU's matrix pipe idles, because the register file
// holds exactly one workgroup.  Here one workgroup per CU walks tiles item = blockIdx.x, += gridDim.x, and the chunk
// pipeline runs straight through tile boundaries:
//   * the next tile's first halo chunk is staged under this tile's last chunk (same double buffer);
//   * the next tile's residual tile (128 px x 128 ch = 64 KiB) arrives by LDS-DMA (global_load_lds: no VGPRs) during
//     this tile, and enters the accumulators at the next tile's start from LDS;
//   * this tile's stores drain under the next tile's MFMAs; the weight ring never empties (same weights again).
// A tile boundary therefore costs the output transform + exchange + store ISSUE only.
// Addressing is buffer-form throughout (scalar resource + per-lane offset computed once per tile + scalar offset per
// chunk): no per-access VALU.  LDS: 2 x 30 KiB halo + 64 KiB residual + 32 KiB exchange (two rounds) = 156 KiB.
template <class C, int WH>
__device__ __forceinline__ void conv_wino8p_body(const ConvParams& p, float* lds) {
    constexpr int KC = C::KC, PS = C::PS, RS = C::RS, GPC = C::GPC, PF = C::PF;
    constexpr int NP8 = 4;
    constexpr int wh = WH;
    float* const lds_res = lds + 2 * C::BUF_FLOATS;
    float* const xch = lds_res + 128 * 128;

    // Raw barrier: with an LDS-DMA possibly in flight hipcc turns __syncthreads() into vmcnt(0) + s_barrier, which
    // would drain the weight ring and the halo loads at every chunk.  LDS accesses are ordered by the explicit
    // lgkmcnt(0); the DMA is ordered by later in-order vmcnt waits of its issuing wave plus these barriers.
    auto wg_barrier = [&]() {
        if constexpr (C::OPT & 1) __syncthreads();      // debugging aid
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // one statement: nothing can be scheduled in between
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1;
    const int tiles_per_image = p.tiles_x * p.tiles_y;
    const int total = tiles_per_image * p.B;
    const int stride = gridDim.x;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;

    // ---- halo loader: 64 pixel slots x 8 channel quads.  Passes 0..2 = halo rows 2i, 2i+1 x columns 0..31, pass 3 =
    // the two leftover columns of all six rows (12 slots).  Per tile and source: 4 byte offsets + a validity mask. ----
    static_assert(C::HH == 6 && C::HW == 34, "loader geometry is written for the 4 x 32 tile");
    const int q = tid & 7, pl = tid >> 3;
    const int prow = pl >> 5, pcol = pl & 31;
    const int wbase = prow * RS + pcol * PS + q * 4;
    const int wbase3 = (pl >> 1) * RS + (32 + (pl & 1)) * PS + q * 4;
    int lb = 0;                     // image of the loader's tile (scalar)
    unsigned voff[NP8];             // byte offset of this thread's slot inside the image, per pass (includes q*16)
    unsigned okmask = 0;            // bit i: the slot of pass i lies inside the image
    auto set_loader = [&](int item, int s) {
        int t = item;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        lb = t / p.tiles_y;
        const unsigned cs = (unsigned)p.csrc[s];
        const int iy0 = ty * C::TH - 1, ix0 = tx * C::TW - 1;
        const int ixa = ix0 + pcol;
        const bool colok = ixa >= 0 && ixa < Wv;
        const int colx = colok ? (ixa >> p.ups) : 0;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int iy = iy0 + 2 * i + prow;
            const bool ok = colok && iy >= 0 && iy < Hv;
            const unsigned pix = ok ? (unsigned)((iy >> p.ups) * p.Win + colx) : 0u;
            voff[i] = (pix * cs + (unsigned)(q * 4)) * 4u;
            okmask |= ok ? (1u << i) : 0u;
        }
        const int iy3 = iy0 + (pl >> 1), ix3 = ix0 + 32 + (pl & 1);
        const bool ok3 = pl < 12 && iy3 >= 0 && iy3 < Hv && ix3 >= 0 && ix3 < Wv;
        const unsigned pix3 = ok3 ? (unsigned)((iy3 >> p.ups) * p.Win + (ix3 >> p.ups)) : 0u;
        voff[3] = (pix3 * cs + (unsigned)(q * 4)) * 4u;
        okmask |= ok3 ? 8u : 0u;
    };
    f32x4 pre[NP8];
    f32x4 gsc, gsh;
    bool cvalid;
    auto issue_loads = [&](int chunk) {
        const int s = chunk >= p.nchunk0;
        const int crel = (s ? chunk - p.nchunk0 : chunk) * KC;
        const int cs = p.csrc[s];
        cvalid = crel + q * 4 < cs;
        // one resource per (source, image); lanes past the channel count read the next pixel or, past the image, 0
        const size_t img = (size_t)p.Hin * p.Win * cs;
        const __amdgpu_buffer_rsrc_t rs = buf_rsrc(p.src[s] + (size_t)lb * img, (unsigned)(img * 4));
#pragma unroll
        for (int i = 0; i < NP8; ++i) pre[i] = buf_load4(rs, voff[i], (unsigned)crel * 4u);
        if (p.gn) {
            const unsigned cg = (unsigned)((s ? p.csrc[0] : 0) + crel) * 4u;
            gsc = buf_load4(buf_rsrc(p.gscale + (size_t)lb * p.ctot, (unsigned)p.ctot * 4u), (unsigned)q * 16u, cg);
            gsh = buf_load4(buf_rsrc(p.gshift + (size_t)lb * p.ctot, (unsigned)p.ctot * 4u), (unsigned)q * 16u, cg);
        }
    };
    auto write_pass = [&](float* buf, int i) {
        f32x4 v = pre[i];
        const bool ok = cvalid && ((okmask >> i) & 1u);
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
    // the issue cursor walks (item, chunk) in execution order, one chunk ahead of the staging, two ahead of the MFMAs
    int is_item = blockIdx.x, is_chunk = 0;
    auto issue_next = [&]() {          // precondition: is_item < total
        if (is_chunk == 0 || is_chunk == p.nchunk0) set_loader(is_item, is_chunk >= p.nchunk0 ? 1 : 0);
        issue_loads(is_chunk);
        if (++is_chunk == p.nchunks) {
            is_chunk = 0;
            is_item += stride;
        }
    };

    // ---- residual tile of an item -> LDS by LDS-DMA: wave w copies 16 pixels of tile row w >> 1 (2 per instruction) ----
    auto dma_residual = [&](int item) {
        int t = item;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int row = wave >> 1, col0 = (wave & 1) * 16;
        const int oy = min(ty * C::TH + row, p.Hout - 1);
        int ch = blockIdx.y * 128 + (lane & 31) * 4;
        if (ch + 4 > p.Cout) ch = 0;                                   // lanes past cout: any valid address (never used)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ox = min(tx * C::TW + col0 + 2 * i + (lane >> 5), p.Wout - 1);
            const float* g = p.residual + (((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + ch;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds_res + (row * 32 + col0 + 2 * i) * 128),
                                             16, 0, 0);
        }
    };

    // ---- operand addressing: lane = (Winograd tile li, channel half lh); this wave reads patch rows wh..wh+2 ----
    const int li = lane & 31, lh = lane >> 5;
    const int wty = li >> 4, wtx = li & 15;
    const int a_base = (2 * wty + wh) * RS + (2 * wtx) * PS + lh * 4;
    const int ntile = blockIdx.y * 4 + wn;
    const bool nvalid = ntile * 32 < p.Cout;
    const int n = ntile * 32 + li;
    const bool nok = nvalid && n < p.Cout;
    // packed weights [ntile][chunk][s][e][xiq][lane][4]: this wave uses xiq = 2*wh, 2*wh + 1
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(p.w + ((size_t)(nvalid ? ntile : 0) * p.nchunks) * 16384 + wh * 512);
    const unsigned lane16 = lane * 16;
    const float bias_n = (nok && p.bias) ? p.bias[n] : 0.f;
    auto tile_add = [&](int item) -> float {      // bias + temb of the tile's image for this lane's channel
        const int b = item / tiles_per_image;
        return (nok && p.temb) ? bias_n + p.temb[(size_t)b * p.temb_ld + n] : bias_n;
    };

    f32x16 acc[8];
    constexpr int RF = 2 * PF;
    f32x4 ring[RF];
    auto foff = [](int f) { return (f >> 1) * 1024 + (f & 1) * 256; };     // float offset of fragment f inside a chunk

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
    // the two V rows of this wave for channel c of the half in `d` (same operation tree as conv_wino8_body)
    auto transform = [&](const f32x2 (&d)[12], int c, float (&v)[8]) {
        float r0[4], r1[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float u0 = d[0 + bb][c], u1 = d[4 + bb][c], u2 = d[8 + bb][c];
            if constexpr (wh == 0) {
                r0[bb] = u0 - u2;
                r1[bb] = u1 + u2;
            } else {
                r0[bb] = u1 - u0;
                r1[bb] = u0 - u2;
            }
        }
        v[0] = r0[0] - r0[2]; v[1] = r0[1] + r0[2]; v[2] = r0[2] - r0[1]; v[3] = r0[1] - r0[3];
        v[4] = r1[0] - r1[2]; v[5] = r1[1] + r1[2]; v[6] = r1[2] - r1[1]; v[7] = r1[1] - r1[3];
    };

    int vpar = 0;   // LDS buffer parity of the chunk being computed
    // One chunk of MFMAs; `stage` = a chunk follows in this workgroup's stream (written into the other buffer during
    // the second half), `issue` = and one after that (its global loads start).
    // Register diet (this kernel keeps the loader state, the ring and the next chunk's loads alive across tile
    // boundaries): ONE patch buffer and scalar operand sets.  Group 2h+1 issues the LDS reads of half h+1 first, runs
    // its first four MFMAs, then transforms channel 0 of the new half (the partner wave covers the rest of the LDS
    // latency); group 2h+2 transforms channel 1.
    auto chunk_body = [&](const int chunk, const bool stage, const bool issue) __attribute__((always_inline)) {
        const float* cur = lds + vpar * C::BUF_FLOATS;
        float* nxt = lds + (vpar ^ 1) * C::BUF_FLOATS;
        const unsigned wcb = (unsigned)chunk * 65536u;                       // byte offset of this chunk's fragments
        const unsigned wnb = chunk + 1 < p.nchunks ? wcb + 65536u : 0u;      // after the last chunk: chunk 0 again (next tile)
        f32x2 dh[12];
        float vv[2][8];
        load_half(cur, 0, dh);
        transform(dh, 0, vv[0]);
#pragma unroll
        for (int g = 0; g < GPC; ++g) {
            const int hh = g >> 1;
            if ((g & 1) == 0) transform(dh, 1, vv[1]);                      // channel 1 of this half, for group g + 1
            else if (hh + 1 < 8) load_half(cur, hh + 1, dh);                // patch of the next half (dh is free now)
            if (stage) {
                constexpr int G0 = GPC - NP8 - 1;
                if (g >= G0 && g < G0 + NP8) write_pass(nxt, g - G0);
                if (g == G0 + NP8 && issue) issue_next();
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int f = g * 2 + k;
                const f32x4 bq = ring[f % RF];
                ring[f % RF] = buf_load4(wrs, lane16, f + RF < 32 ? wcb + foff(f + RF) * 4u : wnb + foff(f + RF - 32) * 4u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[k * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[g & 1][k * 4 + j], bq[j], acc[k * 4 + j], 0, 0, 0);
                if (k == 0 && (g & 1) && hh + 1 < 8) {
                    __builtin_amdgcn_sched_barrier(0);                        // reads + 4 MFMAs first, then the transform
                    transform(dh, 0, vv[0]);                                 // channel 0 of the next half, for group g + 1
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto transform2 = [&](const f32x2 (&d)[12], f32x2 (&v)[8]) {           // see conv_wino8_body
        f32x2 r0[4], r1[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x2 u0 = d[0 + bb], u1 = d[4 + bb], u2 = d[8 + bb];
            if constexpr (wh == 0) {
                r0[bb] = u0 - u2;
                r1[bb] = u1 + u2;
            } else {
                r0[bb] = u1 - u0;
                r1[bb] = u0 - u2;
            }
        }
        v[0] = r0[0] - r0[2]; v[1] = r0[1] + r0[2]; v[2] = r0[2] - r0[1]; v[3] = r0[1] - r0[3];
        v[4] = r1[0] - r1[2]; v[5] = r1[1] + r1[2]; v[6] = r1[2] - r1[1]; v[7] = r1[1] - r1[3];
    };

    auto chunk_body_pair = [&](const int chunk, const bool stage, const bool issue) __attribute__((always_inline)) {
        const float* cur = lds + vpar * C::BUF_FLOATS;
        float* nxt = lds + (vpar ^ 1) * C::BUF_FLOATS;
        const unsigned wcb = (unsigned)chunk * 65536u;                       // byte offset of this chunk's fragments
        const unsigned wnb = chunk + 1 < p.nchunks ? wcb + 65536u : 0u;      // after the last chunk: chunk 0 again (next tile)
        f32x2 dh[12];
        f32x2 vv[2][8];
        load_half(cur, 0, dh);
        transform2(dh, vv[0]);
#pragma unroll
        for (int g = 0; g < GPC; ++g) {
            const int hh = g >> 1;
            if (hh + 1 < 8) {
                if ((g & 1) == 0) load_half(cur, hh + 1, dh);
                else transform2(dh, vv[(hh + 1) & 1]);
            }
            if (stage) {
                constexpr int G0 = GPC - NP8 - 1;
                if (g >= G0 && g < G0 + NP8) write_pass(nxt, g - G0);
                if (g == G0 + NP8 && issue) issue_next();
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int f = g * 2 + k;
                const f32x4 bq = ring[f % RF];
                ring[f % RF] = buf_load4(wrs, lane16, f + RF < 32 ? wcb + foff(f + RF) * 4u : wnb + foff(f + RF - 32) * 4u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[k * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[hh & 1][k * 4 + j][g & 1], bq[j], acc[k * 4 + j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- stream prologue: first tile's chunk 0 staged, chunk 1 (or the next tile's chunk 0) in flight ----
    int item = blockIdx.x;           // grid.x <= total
    const int my_chunks = ((total - 1 - (int)blockIdx.x) / stride + 1) * p.nchunks;   // chunks in this workgroup's stream
    int done = 0;                    // chunks computed so far
    const int dma_chunk = p.nchunks > 1 ? 1 : 0;
    issue_next();
#pragma unroll
    for (int i = 0; i < NP8; ++i) write_pass(lds, i);
    if (my_chunks > 1) issue_next();
    if (p.residual) dma_residual(item);
    else
        for (int i = tid * 4; i < 128 * 128; i += 2048) *reinterpret_cast<f32x4*>(lds_res + i) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < RF; ++f) ring[f] = buf_load4(wrs, lane16, foff(f) * 4u);
    float add_cur = tile_add(item), add_next = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();

    if (!nvalid) {
        // a wave whose 32 channels lie beyond cout only helps with staging, DMA and barriers
        for (; item < total; item += stride) {
            const bool next_item = item + stride < total;
            if (item != (int)blockIdx.x && done + 1 < my_chunks) issue_next();
            if (p.nchunks == 1) wg_barrier();
            for (int chunk = 0; chunk < p.nchunks; ++chunk) {
                if (chunk == dma_chunk && next_item && p.residual) dma_residual(item + stride);
                if (done + 1 < my_chunks) {
#pragma unroll
                    for (int i = 0; i < NP8; ++i) write_pass(lds + (vpar ^ 1) * C::BUF_FLOATS, i);
                    if (done + 2 < my_chunks && chunk + 1 < p.nchunks) issue_next();
                }
                ++done;
                vpar ^= 1;
                wg_barrier();
            }
            wg_barrier();
            wg_barrier();
            wg_barrier();
        }
        return;
    }

    for (; item < total; item += stride) {
        // ---- accumulator init: bias + temb + residual (from the LDS copy) injected through M (see conv_wino_kernel);
        // half 0 owns M[0][0] = R00 and M[0][3] = -R01; half 1 owns M[3][0] = -R10 and M[3][3] = R11 ----
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tile = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int py = 2 * (tile >> 4) + wh, px = 2 * (tile & 15);
            const float v0 = ((C::OPT & 2) ? 0.f : lds_res[(py * 32 + px) * 128 + wn * 32 + li]) + add_cur;
            const float v1 = ((C::OPT & 2) ? 0.f : lds_res[(py * 32 + px + 1) * 128 + wn * 32 + li]) + add_cur;
            if constexpr (wh == 0) {
                acc[0][r] = v0;
                acc[3][r] = -v1;
            } else {
                acc[4][r] = -v0;
                acc[7][r] = v1;
            }
        }
        const bool next_item = item + stride < total;
        if (item != (int)blockIdx.x && done + 1 < my_chunks) issue_next();
        if (p.nchunks == 1) wg_barrier();     // every wave has consumed the residual copy before the next DMA lands
        for (int chunk = 0; chunk < p.nchunks; ++chunk) {
            // Residual of the NEXT tile: issued one barrier after every wave has consumed the current copy, and with a
            // whole tile of younger, waited-for loads ahead of its first use (vmcnt retires in order): no drain needed.
            if (chunk == dma_chunk && next_item && p.residual) dma_residual(item + stride);
            if (chunk == p.nchunks - 1 && next_item) add_next = tile_add(item + stride);
            // (the loads of the chunk after next are not started under a tile's LAST chunk: their registers would be
            // live across the epilogue; they start right after the next tile's accumulator init instead)
            const bool stage = done + 1 < my_chunks, issue = done + 2 < my_chunks && chunk + 1 < p.nchunks;
            if constexpr (C::OPT & 4) chunk_body_pair(chunk, stage, issue);      // debugging aid: the pair-transform schedule
            else chunk_body(chunk, stage, issue);
            ++done;
            vpar ^= 1;
            wg_barrier();
        }

        // ---- partial output transform: this half's rows of M -> partial Y (linear), halves swapped through LDS in two
        // rounds of 4 registers, each round stored at once.  tmp[0][j] = M0j + M1j + M2j, tmp[1][j] = M1j - M2j - M3j;
        // half 0 (rows 0,1) contributes (M0j + M1j, M1j), half 1 (rows 2,3) (M2j, -M2j - M3j).
        // Registers 0..7 are finished by half 0, 8..15 by half 1.  GroupNorm sums are taken BEFORE the quad transposes,
        // where all 16 values of a lane belong to its own channel n (one double pair per lane). ----
        int t = item;
        const int tx = t % p.tiles_x;
        t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int oy0 = ty * C::TH, ox0 = tx * C::TW;
        const int q4 = li & 3;
        const int cq = ntile * 32 + (li & ~3);
        const bool quad_ok = cq < p.Cout;
        const bool vec_ok = (p.out_ld & 3) == 0 && cq + 4 <= p.out_ld;
        const bool full_tile = oy0 + C::TH <= p.Hout && ox0 + C::TW <= p.Wout;
        const bool fast_store = full_tile && (p.out_ld & 3) == 0 && ntile * 32 + 32 <= p.Cout;      // wave-uniform
        // fast path: scalar base per (kk, pos) + one per-lane offset: tile = 8*(2*wh + kk) + q4 + 4*lh -> row wh, column 8*kk + q4 + 4*lh
        const __amdgpu_buffer_rsrc_t ro = buf_rsrc(p.out + (((size_t)b * p.Hout + oy0 + 2 * wh) * p.Wout + ox0) * p.out_ld);
        const unsigned so = ((unsigned)(2 * (q4 + 4 * lh)) * (unsigned)p.out_ld + (unsigned)cq) * 4u;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float mine[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = r & 7;
                if ((rr >> 2) != kk) continue;
                float tt[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float m0 = acc[0 + j][r], m1 = acc[4 + j][r];
                    if constexpr (wh == 0) {
                        tt[0][j] = m0 + m1;
                        tt[1][j] = m1;
                    } else {
                        tt[0][j] = m0;
                        tt[1][j] = -m0 - m1;
                    }
                }
                float y[4];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    y[a * 2 + 0] = tt[a][0] + tt[a][1] + tt[a][2];
                    y[a * 2 + 1] = tt[a][1] - tt[a][2] - tt[a][3];
                }
                const bool keep = (r >> 3) == wh;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (keep) mine[(rr & 3) * 4 + k] = y[k];
                    else xch[((wave * 16) + (rr & 3) * 4 + k) * 64 + lane] = y[k];
                }
            }
            wg_barrier();
            {
                const int partner = wave ^ 1;
#pragma unroll
                for (int k = 0; k < 16; ++k) mine[k] += xch[((partner * 16) + k) * 64 + lane];
            }
            if (kk == 0) wg_barrier();      // the partner has read round 0 before round 1 overwrites it
            if (p.stats) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tile = 8 * (2 * wh + kk) + i + 4 * lh;
                    const int oy = oy0 + 2 * (tile >> 4), ox = ox0 + 2 * (tile & 15);
#pragma unroll
                    for (int pos = 0; pos < 4; ++pos) {
                        if (full_tile || (oy + (pos >> 1) < p.Hout && ox + (pos & 1) < p.Wout)) {
                            const double dv = (double)mine[i * 4 + pos];
                            s1 += dv;
                            s2 = fma(dv, dv, s2);
                        }
                    }
                }
            }
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                float x[4] = {mine[0 * 4 + pos], mine[1 * 4 + pos], mine[2 * 4 + pos], mine[3 * 4 + pos]};
                quad_transpose(x, q4);
                if (fast_store) {
                    const unsigned pix = (unsigned)(pos >> 1) * (unsigned)p.Wout + (unsigned)(16 * kk + (pos & 1));
                    buf_store4(ro, so, pix * (unsigned)p.out_ld * 4u, f32x4{x[0], x[1], x[2], x[3]});
                } else {
                    const int tile = 8 * (2 * wh + kk) + q4 + 4 * lh;
                    const int py = oy0 + 2 * (tile >> 4) + (pos >> 1), px = ox0 + 2 * (tile & 15) + (pos & 1);
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
        add_cur = add_next;
    }
}

template <class C>
__global__ __launch_bounds__(512, 2) void conv_wino8p_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * C::BUF_FLOATS + 128 * 128 + 8 * 16 * 64];
    // wave-uniform: both arms execute the same number of barriers
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) conv_wino8p_body<C, 1>(p, lds);
    else conv_wino8p_body<C, 0>(p, lds);
}

template <class C>
inline int conv_wino8p_launch(const ConvParams& p, hipStream_t stream) {
    const int total = p.tiles_x * p.tiles_y * p.B;
    const int ny = ceil_div(p.Cout, C::BN);
    int nx = 256 / ny;               // one workgroup per CU (256 CUs): the register file holds exactly one
    if (nx < 1) nx = 1;
    if (nx > total) nx = total;
    hipLaunchKernelGGL(conv_wino8p_kernel<C>, dim3(nx, ny), dim3(512), 0, stream, p);
    return check_launch();
}

