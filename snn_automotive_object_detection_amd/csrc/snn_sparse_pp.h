// Ping-pong form of the FAT conv + LIF (round 6; VERDICT r5 item 1b: "an explicit two-phase ping-pong of the two co-resident waves per SIMD").
// MEASURED NEGATIVE: bit-identical to the FAT conv and 14 % SLOWER (profiles/r6_pingpong.txt, with the interval timeline this file's SNN_EXP_TIMELINE stamps
// produce).  Compiled only into -DSNN_PINGPONG builds (tools/ab_build.sh); the product library does not contain it.  What the design assumed and what the
// hardware did is in the last paragraph.
// Included by snn_kernels.hip behind snn_sparse.h (its staging layout, loop body and register LIF are reused verbatim).
//
// What round 5 measured on k_gemm_lif_sparse<true, 1, FAT> (profiles/r5_sparse_whatif_final.txt, r5_fat_timelines.txt, r6_conv_stalls.txt): the
// matrix instructions of a launch take 1.27 ms at the clock it holds, everything else 0.96 ms, and the two overlap to 1.79 ms - each SIMD holds two
// waves of two INDEPENDENT work-groups whose phases (copies + fragment set-up | 108 matrix instructions | barrier) meet by chance: whenever both
// are outside their matrix phase the pipe idles (28 % of the launch), and no single ingredient is worth more than 6 %.
//
// Here the two waves of a SIMD belong to ONE work-group of 8 waves (512 threads, up to 256 registers, one work-group per CU) and alternate by
// construction.  Half h = wave >> 2 (waves h * 4 .. h * 4 + 3 sit on SIMDs 0 .. 3) is a complete FAT-conv work-group of round 5: its own tile
// (64 positions x 64 columns, wave = all planes of 16 positions), its own ring in LDS.  A step of a half is two phases separated by WORK-GROUP
// barriers:
//     Y (memory)  the step's A fragments and first weight fragments LDS -> registers, then the LDS-DMA copies of the steps ahead
//     X (matrix)  the step's 108 matrix instructions (+ the remaining weight-fragment reads), then the wait for the next step's copies
// and half 1 runs one barrier interval behind half 0, so every interval pairs one half's X with the other's Y: a SIMD's matrix pipe always has
// exactly one wave feeding it, the other wave's memory phase (a quarter of its length) hides completely, and nothing competes for issue slots.
// Ring per half: THREE A slots (8 KB each: the row words come from HBM / the Infinity Cache and are requested two steps ahead, since an interval
// is half of what a step was) and two B slots (24 KB each: weight planes out of the L2) - 72 KB, 148 KB per work-group.
// The work-groups are PERSISTENT (grid = CUs): each walks its share of the XCD's tile list, two adjacent tiles (same positions, neighbouring
// column blocks) at a time - with one work-group per CU a relaunch per tile would leave the pipe idle through every prologue.  The LIF runs in
// registers (sp_lif_regs) at the end of a tile; both halves' epilogues fall together (one interval apart), the only stretch without matrix work.
// Same instructions in the same order per accumulator as the FAT shape: bit-identical spike planes (tests/test_gpu_sparse.py).
//
// What happened (profiles/r6_pingpong.txt): an interval takes ~2930 cycles against the 1728 of its 108 matrix instructions.  X alone is 2230-2480 cycles - ONE wave issues
// this stream at 20.7-23 cycles per matrix instruction (18.8 without any fragment read in X; the instruction type does not matter), because an in-order wave stalls at every
// s_waitcnt on a fragment and 11 % of its steps carry the secondary plane's pass (+ ~1080 cycles) - and the barrier behind X adds ~510 cycles of skew: 8 waves in lock
// step wait for the slowest.  Y (1460-1530 cycles: 14 LDS-DMA instructions at ~100 cycles of issue each) hides as designed.  Two INDEPENDENT waves per SIMD (the FAT conv)
// reach 20.6 cycles per instruction per SIMD by filling each other's stalls by chance; a pipe fed by one wave at a time cannot.
#pragma once

#define PP_A_SLOTS 3
#define PP_A_BYTES (SP_A_ARR * SP_ROWS * 4)                      // one step's row arrays of a half (4 waves x 8 M-tile slots x 16 rows): 8 KB
#define PP_HALF (PP_A_SLOTS * PP_A_BYTES + 2 * SP_B_BYTES)       // 72 KB
#define PP_LDS (G3_LUT_BYTES + 2 * PP_HALF)                      // 148 KB
#define PP_A_LOADS 8                                             // LDS-DMA instructions of one A stage per wave (two passes x four arrays)

template <int NS>                                                // sparse M-tiles per wave (two dense ones in front): T = NS + 3
__global__ __launch_bounds__(512, 2) void k_conv_lif_pp(const SparseConvArgs args) {
    constexpr int ND = 2, MTS = SP_MT_FAT, NT = 4, ROWS = SP_ROWS, NPASS = 2, BD = NS >= 6 ? 1 : 2;     // (T = 9: no registers for a third weight-fragment buffer)
    static_assert(ND + NS <= MTS && (3 * NT) % (BD + 1) == 0, "slots; fragment ring carries over");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned char* const lut = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 2, wm = wave & 3;                     // half, row-wave of the half
    const int lr = lane & 15, lg = lane >> 4;
    unsigned char* const ring = smem + G3_LUT_BYTES + h * PP_HALF;
    const uint32_t ring_base = smem_base + G3_LUT_BYTES + h * PP_HALF;
    if (tid < 256) {
        uint4 q;
        q.x = bf16_pair(tid, 0); q.y = bf16_pair(tid, 1); q.z = bf16_pair(tid, 2); q.w = bf16_pair(tid, 3);
        *reinterpret_cast<uint4*>(lut + tid * 16) = q;
    }
    const int pb = args.pb, M = args.M, Kc = args.Kc, Np = args.Np;
    const int n_steps = Kc / 2;
    const void* a_base[SP_A_ARR];
#pragma unroll
    for (int j = 0; j < SP_A_ARR; ++j) a_base[j] = sgpr_ptr(reinterpret_cast<const char*>(args.enc) + (size_t)j * args.Pe * 4);
    const int cw2_s = __builtin_amdgcn_readfirstlane(args.Cw / 2);
    const uint32_t a_dst = ring_base + wm * MTS * 64;                                              // + A slot, pass * 256
    const uint32_t b_dst = ring_base + PP_A_SLOTS * PP_A_BYTES + wm * 1024;                        // + B slot
    const unsigned long long b_chunk = (unsigned long long)Np * 64, b_plane = args.plane_elems * 2;
    const unsigned char* const a_rd = ring + (wm * MTS * 16 + lr) * 4;                             // + A slot, M-tile slot * 64
    const unsigned char* const b_rd = ring + PP_A_SLOTS * PP_A_BYTES + lr * G3_ROWB + ((lg ^ G3_SWZ(lr)) << 4);   // + B slot
    __syncthreads();                                            // the table

    // ---- this work-group's tiles: XCD x = blockIdx % 8 owns xcd_cpx column blocks on a contiguous range of row tiles (k_gemm_lif_sparse's order);
    // the work-groups of an XCD take its tile list in pairs j = 2 pr + h: same row tile (same positions: the second half's row words are L2 hits),
    // neighbouring column blocks
    const int x = blockIdx.x & 7, wslot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int cpx = args.xcd_cpx, groups = args.n_blocks / cpx;
    const int pairs = (args.xcd_contig * cpx + 1) / 2;
    for (int pr = wslot; pr < pairs; pr += wpx) {
        const int j = 2 * pr + h;
        const int nb = __builtin_amdgcn_readfirstlane((x % groups) * cpx + j % cpx);
        const int mb = __builtin_amdgcn_readfirstlane((x / groups) * args.xcd_contig + j / cpx);
        const bool live = j / cpx < args.xcd_contig && mb < args.n_tiles;      // (half-uniform; a dead half keeps the barriers)
        const int m0 = mb * pb;
        // ---- per-tile staging state (k_gemm_lif_sparse, FAT conv on 4 x 1 waves: lane L of pass ps stages row L & 15 of M-tile slot 4 ps + (L >> 4))
        uint32_t voff[NPASS], inc[NPASS], tap_fix[NPASS], row_fix[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int xs = 4 * ps + (lane >> 4);
            const int xplane = args.mt_plane[wm][xs];
            const bool xused = xplane >= 0, xdense = xused && xplane < args.nd;
            const int lp = min(args.mt_j[wm][xs] * 16 + (lane & 15), pb - 1);
            const int p = min(max(m0, 0) + (xused ? lp : 0), M - 1);
            int l = 0;
            while (l + 1 < args.n_levels && p >= args.lv[l + 1].pos_base) ++l;
            const int H = args.lv[l].H, W = args.lv[l].W;
            const int local = p - args.lv[l].pos_base;
            const int n = local / (H * W), rem = local % (H * W);
            const int y = rem / W, xx = rem % W;
            const uint32_t row0 = (uint32_t)args.lv[l].tile_begin + (uint32_t)((n * (H + 2) + y) * (W + 2) + xx);      // tap (-1, -1)
            const uint32_t Pe = args.Pe;
            const int Cw2 = args.Cw / 2;
            if (xdense || !xused) {
                const int t = xused ? xplane : 0;
                voff[ps] = (uint32_t)(((size_t)t * args.Cw * Pe + row0) * 4);
                inc[ps] = 2 * Pe * 4;
            } else {
                const uint32_t delta = (uint32_t)((const char*)args.cmp - (const char*)args.enc);
                voff[ps] = delta + (uint32_t)(((size_t)(xplane - args.nd) * Cw2 * SP_A_ARR * Pe + row0) * 4);
                inc[ps] = SP_A_ARR * Pe * 4;
            }
            tap_fix[ps] = 4u - (uint32_t)Cw2 * inc[ps];         // next tap of the row: one position on, back to channel word 0
            row_fix[ps] = (uint32_t)((W + 2 - 3) * 4);          // after the third tap of a row: one padded image row down
        }
        int f_c = 0, f_tap = 0;
        auto stage_a = [&](const uint32_t slot_off) __attribute__((always_inline)) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const uint32_t d = __builtin_amdgcn_readfirstlane(a_dst + slot_off + ps * 256);
                asm volatile("s_mov_b32 m0, %5\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1\n\t"
                             "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2\n\t"
                             "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dword %0, %3\n\t"
                             "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dword %0, %4"
                             :: "v"(voff[ps]), "s"(a_base[0]), "s"(a_base[1]), "s"(a_base[2]), "s"(a_base[3]), "s"(d), "s"((uint32_t)(ROWS * 4))
                             : "memory", "scc");
                voff[ps] += inc[ps];
            }
            f_c = __builtin_amdgcn_readfirstlane(f_c + 1);
            if (f_c == cw2_s) {
                f_c = 0;
                f_tap = __builtin_amdgcn_readfirstlane(f_tap + 1);
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) voff[ps] += tap_fix[ps] + (f_tap == 3 ? row_fix[ps] : 0u);
                if (f_tap == 3) f_tap = 0;
            }
        };
        // B staging: 24 pieces of 1 KB per step (2 chunks x 3 planes x 4 blocks of 16 columns); wave wm of the half copies pieces wm, wm + 4, ..
        const int brow = wm * 16 + (lane >> 2);
        const uint32_t b_off = (uint32_t)((nb * 64 + brow) * 64 + (((lane & 3) ^ G3_SWZ(brow)) << 4));
        unsigned long long s_ptr = (unsigned long long)args.wpk;                  // chunk 2 * step, plane 0
        auto stage_b = [&](const uint32_t slot_off) __attribute__((always_inline)) {
            const uint32_t d = __builtin_amdgcn_readfirstlane(b_dst + slot_off);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int piece = wm + 4 * i;                                      // wave-uniform; piece & 3 = wm = its block of 16 columns
                const int c2 = piece / 12, pl = (piece % 12) / 4;
                glds16(sgpr_ptr(reinterpret_cast<const void*>(s_ptr + c2 * b_chunk + pl * b_plane)), b_off, d + c2 * (3 * 64 * G3_ROWB) + pl * (64 * G3_ROWB));
            }
            s_ptr += 2 * b_chunk;
        };

        f32x4 acc[MTS][NT];
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // the step's A fragments and its first weight fragments, loaded in the Y phase
        bfv8 p_ad[ND][2], p_as[NS];
        int p_ix[NS];
        unsigned long long p_sec[2] = {0, 0};
        bfv16 bbuf[BD + 1];
        // (all occupancy bytes first, then all table fragments: left to itself hipcc chains byte -> fragment nine times, nine LDS round trips in a row)
        auto load_a_all = [&](const uint32_t off) __attribute__((always_inline)) {
            uint32_t byt[ND * 2 + NS];
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) byt[d * 2 + c2] = *reinterpret_cast<const uint8_t*>(a_rd + off + c2 * (ROWS * 4) + d * 64 + lg);
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const unsigned char* r = a_rd + off + (ND + q) * 64;
                byt[ND * 2 + q] = *reinterpret_cast<const uint8_t*>(r + lg);
                p_ix[q] = (int)*reinterpret_cast<const uint16_t*>(r + (1 + (lg >> 1)) * (ROWS * 4) + 2 * (lg & 1));
            }
            uint32_t o2[(NS + 3) / 4];
#pragma unroll
            for (int b4 = 0; b4 < (NS + 3) / 4; ++b4) o2[b4] = *reinterpret_cast<const uint32_t*>(a_rd + off + 3 * (ROWS * 4) + min(ND + 4 * b4 + lg, MTS - 1) * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) p_ad[d][c2] = *reinterpret_cast<const bfv8*>(lut + (byt[d * 2 + c2] << 4));
#pragma unroll
            for (int q = 0; q < NS; ++q) p_as[q] = *reinterpret_cast<const bfv8*>(lut + (byt[ND * 2 + q] << 4));
#pragma unroll
            for (int b4 = 0; b4 < (NS + 3) / 4; ++b4)
                p_sec[b4] = __ballot(o2[b4] != 0u) & (NS - 4 * b4 >= 4 ? ~0ull : ((1ull << (16 * (NS - 4 * b4))) - 1ull));
        };
        auto load_b = [&](bfv16& dst, const uint32_t off, const int gn) __attribute__((always_inline)) {
            const bfv8 lo = *reinterpret_cast<const bfv8*>(b_rd + off + (2 - gn % 3) * (64 * G3_ROWB) + (gn / 3) * 16 * G3_ROWB);
            const bfv8 hi = *reinterpret_cast<const bfv8*>(b_rd + off + 3 * 64 * G3_ROWB + (2 - gn % 3) * (64 * G3_ROWB) + (gn / 3) * 16 * G3_ROWB);
            dst = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        };
        // X: the step's matrix instructions (k_gemm_lif_sparse's do_part, weight fragments BD groups ahead: the first BD arrive from the Y phase)
        auto x_phase = [&](const uint32_t oa, const uint32_t ob) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 3 * NT; ++g) {
#ifndef SNN_EXP_PP_NOBREAD                               // (timing experiment: the weight fragments stay what the Y phase loaded)
                if (g + BD < 3 * NT) load_b(bbuf[(g + BD) % (BD + 1)], ob, g + BD);
#endif
                const bfv16 bb = bbuf[g % (BD + 1)];
                const bfv8 b0 = __builtin_shufflevector(bb, bb, 0, 1, 2, 3, 4, 5, 6, 7), b1 = __builtin_shufflevector(bb, bb, 8, 9, 10, 11, 12, 13, 14, 15);
#ifdef SNN_EXP_PP_NOMFMA
                asm volatile("" :: "v"(bb));
#elif defined(SNN_EXP_PP_ALLDENSE)                       // (timing experiment: the same number of matrix instructions, all of the dense type)
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    acc[d][g / 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p_ad[d][0], b0, acc[d][g / 3], 0, 0, 0);
                    acc[d][g / 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p_ad[d][1], b1, acc[d][g / 3], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    acc[ND + q][g / 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p_as[q], (q & 1) ? b1 : b0, acc[ND + q][g / 3], 0, 0, 0);
#elif defined(SNN_EXP_PP_ALLSPARSE)                      // (timing experiment: ... all of the structured-sparse type)
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    acc[d][g / 3] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(p_ad[d][0], bb, acc[d][g / 3], p_ix[0], 0, 0);
                    acc[d][g / 3] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(p_ad[d][1], bb, acc[d][g / 3], p_ix[1], 0, 0);
                }
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    acc[ND + q][g / 3] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(p_as[q], bb, acc[ND + q][g / 3], p_ix[q], 0, 0);
#else
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    acc[d][g / 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p_ad[d][0], b0, acc[d][g / 3], 0, 0, 0);
                    acc[d][g / 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p_ad[d][1], b1, acc[d][g / 3], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    acc[ND + q][g / 3] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(p_as[q], bb, acc[ND + q][g / 3], p_ix[q], 0, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((p_sec[0] | p_sec[1]) != 0ull) {               // (rare) the secondary plane of the M-tiles that have one in this step
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    if (((p_sec[q >> 2] >> (16 * (q & 3))) & 0xffffull) == 0ull) continue;
                    const unsigned char* r = a_rd + oa + (ND + q) * 64 + 3 * (ROWS * 4);
                    const uint32_t occ = *reinterpret_cast<const uint8_t*>(r + lg);
                    const bfv8 a2 = *reinterpret_cast<const bfv8*>(lut + (occ << 4));
                    const int i2 = 0xeeee;                       // every nibble: positions (2, 3)
#pragma unroll
                    for (int g = 0; g < 3 * NT; ++g) {
                        const bfv8 c0 = *reinterpret_cast<const bfv8*>(b_rd + ob + (2 - g % 3) * (64 * G3_ROWB) + (g / 3) * 16 * G3_ROWB);
                        const bfv8 c1 = *reinterpret_cast<const bfv8*>(b_rd + ob + 3 * 64 * G3_ROWB + (2 - g % 3) * (64 * G3_ROWB) + (g / 3) * 16 * G3_ROWB);
                        bfv16 bb;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { bb[i] = c0[i]; bb[8 + i] = c1[i]; }
                        acc[ND + q][g / 3] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a2, bb, acc[ND + q][g / 3], i2, 0, 0);
                    }
                }
            }
        };
#ifdef SNN_EXP_PP_NOBAR                                  // (timing experiments - wrong results: what does each ingredient of an interval cost?)
#define PP_BARRIER() do { asm volatile("" ::: "memory"); } while (0)
#else
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
#ifdef SNN_EXP_TIMELINE                                  // diagnostic build: shader-clock stamps (s_memtime) of work-group 0's first tile pair, five per step and half
#define PP_STAMP(i) do { if (blockIdx.x == 0 && pr == wslot && (tid & 255) == 0 && s < 64) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                                                   args.tl[((size_t)h * 64 + s) * 8 + (i)] = t_; } } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#endif

        if (h == 1) PP_BARRIER();                               // half 1 runs one interval behind half 0
        // ---- P: the first copies of the tile (A(0), B(0), A(1)); A(0) and B(0) must have landed before Y_0
        if (live) {
            stage_a(0);
            stage_b(0);
            if (n_steps > 1) stage_a(PP_A_BYTES);
            if (n_steps > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        PP_BARRIER();
        uint32_t sa = 0, sa2 = 2 * PP_A_BYTES;                  // A slot of step s, of step s + 2
        for (int s = 0; s < n_steps; ++s) {
            const uint32_t ob = (uint32_t)((s & 1) * SP_B_BYTES), ob_n = (uint32_t)(((s + 1) & 1) * SP_B_BYTES);
            // ---- Y_s: this step's fragments into registers, then the copies of the steps ahead (B(s + 1) first: the wait behind X_s leaves A(s + 2) in flight)
            PP_STAMP(0);
            if (live) {
#ifndef SNN_EXP_PP_NOY
                load_a_all(sa);
#pragma unroll
                for (int g0 = 0; g0 < BD; ++g0) load_b(bbuf[g0], ob, g0);
#endif
#ifndef SNN_EXP_PP_NOCOPY
                if (s + 1 < n_steps) stage_b(ob_n);
                if (s + 2 < n_steps) stage_a(sa2);
#endif
            }
            PP_STAMP(1);
            PP_BARRIER();
            PP_STAMP(2);
            // ---- X_s
            if (live) {
                x_phase(sa, ob);
                asm volatile("" ::: "memory");
                PP_STAMP(3);
#ifndef SNN_EXP_PP_NOWAIT
                if (s + 2 < n_steps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            }
            PP_STAMP(4);
            PP_BARRIER();
            sa = sa + PP_A_BYTES == PP_A_SLOTS * PP_A_BYTES ? 0u : sa + PP_A_BYTES;
            sa2 = sa2 + PP_A_BYTES == PP_A_SLOTS * PP_A_BYTES ? 0u : sa2 + PP_A_BYTES;
        }
        // ---- epilogue: the LIF in registers (sp_lif_regs), each wave for its own 16 positions; no LDS, no barrier
        if (live) {
            const int T = args.T;
            uint32_t mine[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
            const bool counting = args.cnt_row != nullptr;
#ifdef SNN_EXP_PP_NOEPI
            for (int r = 0; r < 4; ++r) mine[r] = __float_as_uint(acc[r][0][0] + acc[r + 1][1][1]);
#else
            if (counting) sp_lif_regs<NS + 3, 1, NT, MTS, true>(acc, args.p, lane, mine, cnt);
            else sp_lif_regs<NS + 3, 1, NT, MTS, false>(acc, args.p, lane, mine, cnt);
#endif
            const int t_mine = 1 + ((lane & 15) >> 1), word = nb * 2 + (lane & 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lp = 16 * wm + 4 * lg + r, pos = m0 + lp;
                if (pos >= M || lp >= pb) continue;
                uint32_t* dst;
                if (args.out_split) dst = args.spk + ((size_t)(word >> 2) * M + pos) * 4 + (word & 3);
                else dst = args.spk + (size_t)pos * (Np >> 5) + word;
                if (t_mine < T) dst[(size_t)t_mine * args.spk_stride] = mine[r];
                if ((lane & 15) < 2) dst[0] = 0u;                                        // step 0: no spike
                if (counting && (lane & 15) == 0 && cnt[r]) atomicAdd(args.cnt_row + pos, cnt[r]);
            }
        }
        if (h == 0) PP_BARRIER();                               // (pairs with half 1's barrier behind its last X)
#undef PP_BARRIER
#undef PP_STAMP
    }
}
