// exact bf16x3 family (default): k_gemm_bf16x3 - the four big contractions on the bf16 matrix cores with T-in-tile LIF
// fusion (K3b) - and its weight packer.  Included by snn_kernels.hip.
#pragma once

// ------------------------------------------------------------------------------------------------
// K3b: exact bf16x3 spike GEMM on the bf16 matrix cores (16x the fp32 MFMA rate, 3 MFMAs per product).
//
// Spikes are exactly {0,1} and every fp32 weight is exactly hi + mid + lo with three bf16 values, so
//   A_bits x W  ==  A_bf16 x W_hi + A_bf16 x W_mid + A_bf16 x W_lo     (every product exact)
// with fp32 accumulation inside v_mfma_f32_32x32x16_bf16.  Measured against fp64 the result is as accurate
// as the fp32 MFMA chain (tools/bf16x3_numerics.hip: rms error 4.1e-8 vs 4.2e-8 at K=2304).
//
// cur[M][ldo] = A_bits[M][K] x W[K][N];   work-group = 8 waves = 256 rows x 128 columns, wave = 64 x 64
// = 4 x 4 tiles of v_mfma_f32_16x16x32_bf16 (one 32-deep chunk = one k-step; in an LDS-fed loop with this
// kernel's traffic the 16x16x32 shape sustains 2.47 PF against 1.90 PF for 32x32x16: tools/mfma_probe4.hip).
// Per 32-deep chunk: B = 3 planes x 128 x 32 bf16 (24 KB) copied global->LDS, A = 256 spike words expanded
// to bf16 in LDS (two 16-bit halves per row, one per thread); both double-buffered, one barrier per chunk.
// LDS rows are 64 B (no padding) with an XOR swizzle of the 16-B units: conflict-free ds_read_b128 fragment
// reads at 80 KB of LDS per work-group, so TWO work-groups share a CU and one's barrier/staging phase hides
// behind the other's MFMAs.
// CONV = true: row m = (t, position) and the chunk (tap, channel word) is gathered straight from the
// encoder bit-planes (9 taps, zero outside the image) - the un-fused time-batched 3x3 convolution.
// ------------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// lane `t` (a compile-time constant) of my0 / my1 <- the two halves of a wave ballot.  The ballot is an SGPR pair (often VCC itself)
// written by the v_cmp right before, and on gfx940 / gfx950 a vector instruction reading an SGPR that a vector instruction wrote needs
// 2 wait states in between (LLVM: VALUWriteSGPRVALURead) - which the hazard recognizer cannot add inside inline asm: without them the
// straight-line epilogue stored stale spike words (tests/test_gpu_period_planes.py; tests/test_code_object.py checks the code object).
// s_nop 2 = 3 wait states.
#define G3_KEEP_BALLOT(my0, my1, b, t)                                                                          \
    asm volatile("s_nop 2\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"                          \
                 : "+v"(my0), "+v"(my1) : "s"((uint32_t)(b)), "s"((uint32_t)((b) >> 32)), "n"(t))

// LDS-DMA: 16 B per lane straight from global memory (wave-uniform 64-bit base + the lane's 32-bit byte offset)
// into LDS at (wave-uniform byte address in M0) + 16*lane; no VGPR destination, completion is counted on vmcnt.
// Issued as inline asm on purpose: once hipcc sees an LDS-DMA in flight it degrades every LDS wait of the loop to
// s_waitcnt lgkmcnt(0) and puts vmcnt(0) in front of every ds_write (possible alias); hidden from it, the fragment
// reads keep their exact counted waits.  The kernel waits for the DMA by hand (vmcnt(0) before the chunk barrier).
// a wave-uniform pointer as an SGPR pair (inline asm "s" operands are not legalised by hipcc)
__device__ __forceinline__ const void* sgpr_ptr(const void* p) {
    const unsigned long long x = (unsigned long long)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
// Three pieces (the three weight planes of a chunk) per call; s_nop 4 / s_nop 0: SGPR -> VMEM-base and M0 -> LDS-DMA
// wait states, which hipcc's hazard recogniser does not insert inside inline asm.
// M0: every asm statement below writes M0 itself right before the copy that reads it and leaves nothing in it that is used
// later, so all that is required of the compiler is that IT keeps no value of its own in M0 across these statements.  LLVM
// reserves M0 on AMDGPU (it is not allocatable, and a clobber entry for it is ignored with a warning - hence none here) and
// only materialises it immediately in front of its own M0 readers (movrel, sendmsg, GWS, LDS-DMA builtins): none of those
// exists in this code object.  tests/test_code_object.py disassembles the built library and asserts exactly that: the only
// instructions that touch M0 are these s_mov_b32, each followed by its global_load_lds within the same asm statement.
__device__ __forceinline__ void glds16(const void* p0, uint32_t voff, uint32_t d0) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(p0), "s"(d0) : "memory");
}
__device__ __forceinline__ void glds16x3(const void* p0, const void* p1, const void* p2, uint32_t voff,
                                         uint32_t d0, uint32_t d1, uint32_t d2) {
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3"
                 :: "v"(voff), "s"(p0), "s"(p1), "s"(p2), "s"(d0), "s"(d1), "s"(d2) : "memory");
}

// Work-group tile: the 8 waves form (8 / WN) x WN; a wave is 16*MT rows x 64 columns.
//   WN = 2: 64*MT rows x 128 columns (256 x 128 at MT = 4): 24 KB of weight planes per chunk
//   WN = 1: 128*MT rows x 64 columns (512 x 64):            12 KB per chunk for the same MFMA work - half the L2 -> LDS
//           copies per FLOP (the 256 x 128 tile pulls 14 B/clk/CU = 7 TB/s chip-wide out of the L2).  Measured equal.
#define G3_BM(wn, mt) ((8 / (wn)) * 16 * (mt))
#define G3_BN(wn) (64 * (wn))
#define G3_ROWB 64                                  // bytes per LDS weight row: 32 bf16, four 16-B units, XOR-swizzled
#define G3_AW_BYTES(wn) (G3_BM(wn, 4) * 4)          // raw spike words of one chunk (one per row): 1 or 2 KB
#define G3_B_BYTES(wn) (3 * G3_BN(wn) * G3_ROWB)    // three weight planes of one chunk: 24 or 12 KB
#define G3_SLOT(wn) (G3_AW_BYTES(wn) + G3_B_BYTES(wn))   // one ring slot = one 32-deep chunk
#define G3_LUT_BYTES 4096                           // byte -> 8 bf16 (0 / 1.0) expansion table
#define G3_STATE_BYTES (512 * 64)                   // register-fused variant: 16 LIF state values per thread
#define G3_LDS(nb, wn) (G3_LUT_BYTES + (nb) * G3_SLOT(wn))   // table at offset 0, then the ring (3 slots: 80896 / 47104 B)
// unit u (= k-group 8u..8u+7) of weight row r lives at physical unit u ^ swz(r), swz = [0,3,2,1][(r >> 2) & 3].  A
// 16x16x32 fragment read has lane l on row l&15, unit l>>4; the four 16-lane groups of a ds_read_b128
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) then each hit 16 distinct 16-B slots of the 256-B bank row.
#define G3_SWZ(r) ((0 - ((r) >> 2)) & 3)

struct Gemm3Args {
    const uint32_t* A;           // fc: [M][Kw] spike words;  conv: encoder planes [T][P][Cw]
    const uint16_t* wpk;         // [3][Kc][Np][32] bf16
    float* out;                  // [M][ldo]
    unsigned long long plane_elems;     // Kc*Np*32
    unsigned long long enc_stride;      // conv: words per time plane
    int M, Kc, Np, ldo, n_blocks;
    int Cw, P_total, n_levels;          // conv only
    // conv + LIF: spikes leave as bit-planes.  G3_CONV_LIF_REG: rows are positions, the T loop runs inside;
    // G3_CONV_LIF_TILE: a 256-row tile = the Tc time steps t0 .. t0+Tc-1 of pb = 256/Tc positions (row = (t - t0)*pb + position).
    // The LIF epilogue always runs T steps; the input current of a step outside the window reads as +0 (dead time steps:
    // lif_feed_forward_step integrates the current of step t AFTER that step's membrane update, so the current of the last
    // step never reaches a spike - and a layer fed by a LIF layer sees no spike at step 0; launchers: lif_windows)
    int T, pb, Tc, t0;
    uint32_t* spk;
    unsigned long long spk_stride;
    // spike-rate side outputs of the T-in-tile epilogues (nullable; zeroed by the caller): conv: spikes per (level, image) slot
    // l * max_n + n; linear layers: spikes per row (RoI)
    unsigned long long* cnt_img;
    uint32_t* cnt_row;
    int max_n;
    // word-major operand planes (the heads' internal format): A is [T][word][row] instead of [T][row][word], a_step = rows per
    // word plane (conv: padded positions of all levels; linear: M).  A chunk's spike words of a tile are then a few contiguous
    // runs (128 B per time step of a conv tile) instead of one 4-byte piece per row at the row pitch - the 4-bytes-per-line
    // gather cost 6 % (conv) / 11 % (fc6) of the kernel in clock.  out_wm: the linear-layer LIF epilogue writes
    // its spike planes word-major as well (fc6 -> fc7).
    int wm, out_wm;
    unsigned long long a_step;
    // out_split (conv + LIF tile epilogue): spike planes in blocks of four words, [T][word / 4][position][4] - a work-group's
    // 128 columns are then ONE 16-byte piece per (t, position) and consecutive positions are adjacent (512-byte runs per time
    // step and tile) instead of half of every 32-byte row: whole 32-byte sectors leave the L2 (WRITE_SIZE 98 -> 49 MB per
    // launch at the Cityscapes pyramid).  Read by k_li_heads_mfma (LiHeadsArgs.half_split).
    int out_split;
    int xcd_classes, n_tiles;    // XCD-aware block order (0: plain row-major order), row tiles of the launch
    // T-in-tile modes, periods != 0 (snn_common.h: PERIOD PLANES): the A planes are the encoder's period planes e_n, row group n - 1 of
    // a tile accumulates u_n = W e_n, and the LIF epilogue forms the current of step t as the sum of the row groups whose bit is
    // set in div[t] (n - 1 for every divisor n <= Tc of t + 1), added in ascending n.  t0 = 0 in this mode.
    int periods;
    uint32_t div[SNN_MAX_STEPS];
    int epi_general;             // SNN_EPI_GENERAL=1 (tests): the LIF tile epilogue's general form even where a straight-line instance exists
    int xcd_contig, xcd_cpx;     // xcd_contig > 0 (conv launches with 2 or 4 column blocks): row tiles per XCD; XCD x = blockIdx % 8 runs xcd_cpx
                                 // (1 or 2) column blocks on a CONTIGUOUS range of row tiles, see the kernel
    // T-in-tile modes: the LAST n_short row-waves of the work-group multiply MT - 1 M-tiles instead of MT ("short" waves: the
    // last 16 of their rows do not exist), so a tile has 16 n_short fewer rows.  With n_short = half the row-waves every
    // SIMD hosts one full and one short wave: tile heights between the MT steps (512 / 448 / 384 / 320 / 256 rows on the
    // 8 x 1 grid), which is what lets a launch of a few hundred work-groups come out at a whole number of rounds per CU
    // (launchers: g3_pick_tile).  Logical tile row (what t * pb + position counts) = physical row - 16 * (short waves before it).
    int n_short;
    NeuronP p;
    ConvLevelDev lv[SNN_MAX_LEVELS];
};

// ---- spike-rate reduction fused into the LIF epilogues (rpn.py:163-172, faster_rcnn.py:556-557: the rates are spike COUNTS / (T * neurons)).
// The wave ballot of a time step already is the spike word pair, so a position's count is a popcount of words the epilogue
// holds anyway; the owner wave of a position keeps it in LDS (pos_cnt[pi]) and after the last pass the work-group sends
// ONE integer atomic per (level, image) slot present in the tile (conv; lanes with equal slots are combined in the wave
// first) or one per row (linear layers).  Integer atomics: the totals do not depend on the order of arrival.
#define G3_CNT_BYTES 2048                           // pos_cnt: one uint32 per tile position (pb <= 512)
template <bool CONV, class Args>
__device__ __forceinline__ void tile_counts_flush(const Args& a, const uint32_t* pos_cnt, int m0, int pb, int tid) {
    const int lane = tid & 63;
    for (int i = tid; i < ((pb + 63) & ~63); i += (int)blockDim.x) {     // whole waves take a round together
        const int pos = m0 + i;
        const bool live = i < pb && pos < a.M;
        const uint32_t c = live ? pos_cnt[i] : 0u;
        if (!CONV) {
            if (live && c) atomicAdd(a.cnt_row + pos, c);
            continue;
        }
        int slot = -1;
        if (live) {
            int l = 0;
            while (l + 1 < a.n_levels && pos >= a.lv[l + 1].pos_base) ++l;
            slot = l * a.max_n + (pos - a.lv[l].pos_base) / (a.lv[l].H * a.lv[l].W);
        }
        unsigned long long rem = __ballot(live);
        while (rem) {                               // one round per distinct slot among the wave's positions (normally one)
            const int leader = __ffsll(rem) - 1;
            const int sl = __shfl(slot, leader);
            const bool mine = live && slot == sl;
            uint32_t v = mine ? c : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == leader && v) atomicAdd(a.cnt_img + sl, (unsigned long long)v);
            rem &= ~__ballot(mine);
        }
    }
}

__device__ __forceinline__ uint32_t bf16_pair(uint32_t w, int j) {      // bits 2j, 2j+1 -> two bf16 (0 / 1.0)
    const uint32_t t = (w >> (2 * j)) & 3u;
    return ((t | (t << 15)) & 0x10001u) * 0x3F80u;
}

// FUSE (conv only): M = positions; per time step the 9*Cw chunks are accumulated, then the LIF update runs
// on the accumulators in registers and only spike bits are written.
//
// LDS: a 4-KB table byte -> 8 bf16, a ring of NB slots (one 32-deep chunk each: 256 raw spike words + 3 weight
// planes), and for FUSE 16 LIF state values per thread.  Staging runs ahead, so the slot of chunk c+1 is already
// complete during chunk c and the first fragments of chunk c+1 are read from LDS BEFORE the barrier that ends chunk
// c - the matrix pipe does not drain at the barrier.
// NB = 3 (80 KB, two work-groups per CU): the weight planes of chunk c+2 are copied during chunk c and must have
// landed at its end (s_waitcnt vmcnt(0)).  NB = 4 (fused variant, which owns its CU): the planes of chunk c+3 are
// copied during chunk c and need to land only by the end of chunk c+1 (s_waitcnt vmcnt(3): the three youngest
// operations, this chunk's copies, stay in flight across the barrier) - no L2 latency is ever waited for.
// The A (spike) fragment of a lane is 8 consecutive k of one row = ONE BYTE of that row's spike word: it is fetched
// as table[byte] by a single ds_read_b128.  No expanded spike image is ever built (the first version spent 36 VALU
// instructions + 2 ds_write_b128 per thread and chunk on it, 10 % of the kernel: every VALU instruction beside
// v_mfma_f32_16x16x32_bf16 competes for the SIMD's vector issue, which the MFMAs alone hold half of the time).
// PD = fragment prefetch distance inside a chunk, in groups of 4 MFMAs (one weight fragment per group).
enum { G3_FC = 0, G3_CONV = 1, G3_CONV_LIF_REG = 2, G3_CONV_LIF_TILE = 3, G3_FC_LIF_TILE = 4 };
// LIF_TILE epilogue: the current tile goes through LDS in two passes of CG = 32*WN columns, row pitch CG + 4 floats
// (conflict-free for the accumulator writes and the column reads)
#define G3_TILE_CG(wn) (32 * (wn))
#define G3_TILE_BYTES(wn) (G3_BM(wn, 4) * (G3_TILE_CG(wn) + 4) * 4)     // 69632 / 73728 B

// MT = 16-row M-tiles per wave: the work-group tile is 64*MT rows (256; 192 / 128 only where a small problem spreads
// better over the CUs that way - per unit of work the smaller tiles are slower: fc6 1.03 / 1.07 / 1.21 ms at MT 4 / 3 / 2).
// MT = 8 ("fat waves"): the SAME 256 x 128 tile and LDS image as MT = 4 / WN = 2, run by FOUR waves of 128 x 64 (2 x 2,
// 256 threads, up to 256 registers): every weight fragment read from LDS feeds 8 MFMAs instead of 4 - 37 % fewer LDS read
// bytes per MFMA, which is energy, i.e. clock, on this power-limited loop (timing build with every second weight-fragment
// read skipped: conv+LIF -4.3 %, in-kernel clock 1.99 -> 2.05 GHz).  Two such work-groups share a CU (2 waves per SIMD).
// (Round 4 also ran fc6 + fc7 as ONE launch with per-row-tile counters and agent-scope release / acquire hand-offs between the two
// layers' work-groups: bit-identical, 1 % slower - profiles/r4_det_pair.txt.  Removed in round 5: a measured negative, and its
// bounded-wait failure path could only trap.)
template <int MODE, int NB, int MT, int WN>
__device__ __forceinline__ void gemm_bf16x3_body(const Gemm3Args& args, const int bid) {
    constexpr bool CONV = MODE == G3_CONV || MODE == G3_CONV_LIF_REG || MODE == G3_CONV_LIF_TILE;
    constexpr bool FUSE = MODE == G3_CONV_LIF_REG, TILE = MODE == G3_CONV_LIF_TILE || MODE == G3_FC_LIF_TILE;
    static_assert(((MT >= 2 && MT <= 4) || MT == 8) && (MT == 4 || !FUSE), "M-tiles per wave");
    static_assert(WN == 1 || (WN == 2 && true), "waves along N");
    static_assert(WN == 2 || !FUSE, "the register-fused variant keeps the 4 x 2 wave grid");
    static_assert(MT != 8 || (WN == 2 && NB == 3), "fat waves: 2 x 2 wave grid on the 3-slot ring");
#ifdef SNN_EXP_TIMELINE     // diagnostic build: wall-clock stamps (s_memrealtime, 100 MHz) of the work-group's phases + where it ran
    unsigned long long tl_entry = 0, tl_loop0 = 0, tl_loop1 = 0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_entry) :: "memory");
#endif
    constexpr int NW = MT == 8 ? 4 : 8;                                   // waves per work-group
    constexpr int BM = (NW / WN) * 16 * MT, BN = G3_BN(WN), WROWS = 16 * MT;   // rows, columns per work-group; rows per wave
    constexpr int AW_BYTES = G3_AW_BYTES(WN);
    static_assert(NB == 3 || NB == 4, "ring depth");
    constexpr int SLOT = G3_SLOT(WN);
    constexpr int PD = (CONV && !FUSE) ? 2 : 3, RING = PD + 1;  // 12 groups per chunk: RING must divide 12 (the 128-register conv
                                                               // rows have 128 registers: one fragment less in flight)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned char* const lut = smem;                // table at LDS offset 0: a fragment address is just (byte << 4)
    unsigned char* const ring = smem + G3_LUT_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    constexpr int NWM = NW / WN;                                          // row-waves of the work-group
    const int first_short = NWM - ((TILE && MT != 8) ? args.n_short : 0);  // row-waves >= first_short are short
    const bool short_wave = __builtin_amdgcn_readfirstlane((int)(wm >= first_short)) != 0;
    // Plain row-major tile order, column block fastest: work-group b runs on XCD b % 8 (round-robin dispatch), so with
    // 2 (or 4, 8) column blocks every XCD only ever sees ONE weight panel - half of the 3.5 MB of conv weight planes,
    // which then stay resident in its 4-MB L2 beside the streaming spike planes.  Both re-orderings tried (each XCD a
    // contiguous eighth of the tiles; both column blocks of a tile on one XCD) put all panels on every XCD and
    // multiplied the L2 fills: FETCH_SIZE 102 -> 323 / 535 MB per launch at unchanged kernel time (profiles/r1_h_*).
    // xcd_classes > 0 (launch_gemm3, four or more column blocks): work-group b = XCD x = b % 8, j = b / 8 takes column block
    // 2 (x % groups) + j % 2 of row tile (j / 2) * classes + x / groups - an XCD sees TWO adjacent weight panels (as much L2 as
    // the one panel of the 256 x 128 tile) and both of them for the same row tiles, so a tile's spike rows come into half as
    // many L2s.  Tiles past the end (the grid is padded to a multiple of `classes` tiles) leave at once.
    int nb = bid % args.n_blocks;
    int mb = bid / args.n_blocks;
    if (args.xcd_classes) {
        const int groups = 8 / args.xcd_classes, x = bid & 7, j = bid >> 3;
        nb = 2 * (x % groups) + (j & 1);
        mb = (j >> 1) * args.xcd_classes + x / groups;
        if (mb >= args.n_tiles) return;
    }
    // xcd_contig (the 3x3 convolution: 2 column blocks): XCD x still sees ONE weight panel (column block x % n_blocks), but its row
    // tiles are a contiguous range of the launch (the 8 / n_blocks XCDs of a column block take a quarter each) instead of every
    // fourth tile.  A tile shares two thirds of its spike rows with the tiles one image row up and down (the 3x3 halo): with
    // tiles of 32 positions those were 12 tiles away, i.e. on the same XCD by accident of 384 = 12 x 32; with the 36-position
    // tiles of the 7-step window they landed on other XCDs and every XCD fetched the halo rows again (FETCH_SIZE 83 -> 172 MB per
    // launch, L2 misses x 1.9: profiles/r3_a_*).  In a contiguous range the neighbours run on the same XCD a few slots apart.
    // With 4 column blocks (the 512 x 64 tile) an XCD takes TWO of them (xcd_cpx = 2: work-groups j, j + 1 of the XCD are the two
    // column blocks of one row tile - 1.77 MB of weight panels in its L2, as before, and every spike row still fetched by 2 XCDs, not 4).
    if (args.xcd_contig) {
        const int x = bid & 7, j = bid >> 3, cpx = args.xcd_cpx, groups = args.n_blocks / cpx;
        nb = (x % groups) * cpx + j % cpx;
        mb = (x / groups) * args.xcd_contig + j / cpx;
        if (j / cpx >= args.xcd_contig || mb >= args.n_tiles) return;
    }
    nb = __builtin_amdgcn_readfirstlane(nb);     // (block-uniform by construction: said explicitly for the LDS-DMA asm's SGPR operands)
    mb = __builtin_amdgcn_readfirstlane(mb);
    const int m0 = TILE ? mb * args.pb : mb * BM;            // first row (TILE: first position) of the tile
    const int Kc = args.Kc, Np = args.Np, M = args.M;

    if (tid < 256) {                                // table entry e: element j = bit j of e as bf16
        uint4 q;
        q.x = bf16_pair(tid, 0); q.y = bf16_pair(tid, 1); q.z = bf16_pair(tid, 2); q.w = bf16_pair(tid, 3);
        *reinterpret_cast<uint4*>(lut + tid * 16) = q;
    }

    // ---- A staging: the spike words of a chunk (one per tile row) go straight into the ring slot by LDS-DMA
    // (global_load_lds_dword: lane L of wave w lands at slot + 256 w + 4 L = row 64 w + L; source = wave-uniform 64-bit base in
    // SGPRs + the lane's 32-bit byte offset; no VGPR destination, no ds_write, counted on vmcnt with the weight planes).
    // Convolution rows read encoder planes that carry a ONE-POSITION ZERO HALO around every image (the encoder writes them
    // that way): every tap of every position is a plain read, there is no border logic in the kernel.
    // Round 1 fetched the words into registers (global_load_dword under a per-lane tap-validity mask) and stored them with
    // ds_write; its stream counters lived in vector registers behind exec-mask branches: ~25 vector instructions per wave
    // and chunk, 12-15 % of the kernel (timing build without the stream: conv+LIF 2.99 -> 2.61 ms, fc6 1.03 -> 0.87 ms).
    const bool a_role = wave * 64 < BM;
    const int xrow = tid & (G3_BM(WN, 4) - 1);      // physical tile row this thread stages
    // logical row (short waves: their last 16 physical rows are holes)
    const int xw = xrow / WROWS, xshort = max(0, xw - first_short);
    const bool xhole = xw >= first_short && xrow % WROWS >= WROWS - 16;
    const int xl = xrow - 16 * xshort;
    const int xt = TILE ? xl / args.pb : 0;         // TILE: time step of the row
    const int xm = TILE ? ((xt < args.Tc && !xhole) ? m0 + xl % args.pb : M) : m0 + xrow;
    const bool a_wm = args.wm != 0;                   // word-major planes: a row's consecutive words are a_step words apart
    const int row_words = a_wm ? 1 : args.Cw;         // conv: words from one (padded) position to the next
    uint32_t a_off = 0;                             // bytes: fc row / conv centre tap, channel word 0
    uint32_t a_pitch = 0;                           // conv: bytes per (padded) image row of the lane's pyramid level
    if (CONV) {
        if (xm < M) {
            const int t = FUSE ? 0 : (TILE ? xt + args.t0 : xm / args.P_total), p = (FUSE || TILE) ? xm : xm % args.P_total;
            int l = 0;
            while (l + 1 < args.n_levels && p >= args.lv[l + 1].pos_base) ++l;
            const int H = args.lv[l].H, W = args.lv[l].W;
            const int local = p - args.lv[l].pos_base;
            const int n = local / (H * W), rem = local % (H * W);
            const int y = rem / W, x = rem % W;
            const size_t prow = (size_t)args.lv[l].tile_begin + ((size_t)n * (H + 2) + y + 1) * (W + 2) + x + 1;   // padded row
            a_off = (uint32_t)(((size_t)t * args.enc_stride + prow * row_words) * 4);
            a_pitch = (uint32_t)((W + 2) * row_words * 4);
        } else {                                    // unused tile rows: position (0, 0) of level 0, image 0 - its taps are in range
            a_off = (uint32_t)((args.lv[0].W + 3) * row_words * 4);
            a_pitch = (uint32_t)((args.lv[0].W + 2) * row_words * 4);
        }
    } else if (TILE) {                              // fc rows of the spike planes [T][M][Kc] / [T][Kc][M]; unused tile rows read row 0
        a_off = xm >= M ? 0u : a_wm ? (uint32_t)(((size_t)(xt + args.t0) * Kc * M + xm) * 4) : (uint32_t)(((size_t)(xt + args.t0) * M + xm) * Kc * 4);
    } else {
        a_off = (uint32_t)((size_t)min(xm, M - 1) * (a_wm ? 1 : Kc) * 4);
    }
    // The fetch stream walks the chunk sequence (t, tap dy, tap dx, channel word) as a running scalar pointer: inside a tap
    // row the wave-uniform word offset dx*Cw + cc just increments; every 3*Cw chunks the lanes step one image row down
    // (one v_add), every Kc chunks one time step on.  Past the last chunk the stream wraps to the start (staged, never
    // multiplied).  The counters are wave-uniform: __builtin_amdgcn_readfirstlane says so, or hipcc keeps them in vector
    // registers and branches on them through the exec mask.
    // (s_nop 4: an SGPR written by SALU / v_readfirstlane needs 5 wait states before a VMEM instruction reads it as its base
    // address or M0, and hipcc's hazard recogniser does not look into inline asm.)
    const int n_steps = FUSE ? args.Tc : 1;                // (register-fused variant: currents of steps 0 .. Tc-1)
    const int row_chunks = __builtin_amdgcn_readfirstlane(3 * args.Cw);
    // scalar steps of the stream, in words: to the next channel / K word of the same rows; (conv) at the end of a tap's Cw words
    // on to the next tap of the row; at the end of a tap row back to its first tap; from the centre tap to tap (-1, -1)'s column
    const int w_step = __builtin_amdgcn_readfirstlane(a_wm ? (int)args.a_step : 1);
    const int tap_adj = __builtin_amdgcn_readfirstlane(a_wm ? 1 - args.Cw * (int)args.a_step : 0);
    const int row_back = __builtin_amdgcn_readfirstlane(a_wm ? 3 : 3 * args.Cw);
    const int tap_back = __builtin_amdgcn_readfirstlane(CONV ? row_words : 0);
    const int cw_s = __builtin_amdgcn_readfirstlane(args.Cw);
    const uint32_t* f_ptr = static_cast<const uint32_t*>(sgpr_ptr(args.A - tap_back));
    uint32_t f_voff = CONV ? a_off - a_pitch : a_off;               // lane: row offset of tap row dy
    int f_t = 0, f_kc = 0, f_j = 0, f_c = 0;
    const uint32_t a_dst = smem_base + G3_LUT_BYTES + wave * 256;   // + slot offset
    auto stage_a = [&](uint32_t slot_off) __attribute__((always_inline)) {
#ifndef SNN_EXP_NO_FETCH                            // (timing only: no spike-word stream at all)
        if (a_role) {
            const uint32_t d = __builtin_amdgcn_readfirstlane(a_dst + slot_off);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1" :: "v"(f_voff), "s"(f_ptr), "s"(d) : "memory");
        }
#endif
        f_ptr = static_cast<const uint32_t*>(sgpr_ptr(f_ptr + w_step));
        if (CONV) {
            f_c = __builtin_amdgcn_readfirstlane(f_c + 1);
            if (f_c == cw_s) {                      // next tap of the row (row-major planes: the words just continue, tap_adj = 0)
                f_c = 0;
                f_ptr = static_cast<const uint32_t*>(sgpr_ptr(f_ptr + tap_adj));
            }
            f_j = __builtin_amdgcn_readfirstlane(f_j + 1);
            if (f_j == row_chunks) {                // next tap row
                f_j = 0;
                f_ptr = static_cast<const uint32_t*>(sgpr_ptr(f_ptr - row_back));
                f_voff += a_pitch;
            }
        }
        f_kc = __builtin_amdgcn_readfirstlane(f_kc + 1);
        if (f_kc == Kc) {
            f_kc = 0; f_j = 0; f_c = 0;
            if (FUSE) { f_t = __builtin_amdgcn_readfirstlane(f_t + 1); if (f_t == n_steps) f_t = 0; }
            f_ptr = static_cast<const uint32_t*>(sgpr_ptr(args.A + (FUSE ? (size_t)f_t * args.enc_stride : 0) - tap_back));
            f_voff = CONV ? a_off - a_pitch : a_off;
        }
    };

    // ---- B staging: LDS-DMA, one 1-KiB piece (16 rows of one plane) per instruction; the 3 * BN/16 pieces of a chunk
    // go round the 8 waves (piece p = wave, wave + 8, wave + 16: all of them on row block wave % (BN/16)).  Lane L
    // lands in physical unit L&3 of row (L>>2) of the piece, so it fetches logical unit (L&3) ^ swz(row): the swizzle
    // is applied on the SOURCE address, the LDS image stays lane-linear ----
    constexpr int RBLK = BN / 16, NPIECE = 3 * RBLK;            // row blocks per plane, pieces per chunk (24 / 12)
    const int brow = (wave % RBLK) * 16 + (lane >> 2);
    const int bcol = min(nb * BN + brow, Np - 1);               // columns past Np: any valid row (never stored)
    const uint32_t b_off = (uint32_t)(bcol * 64 + (((lane & 3) ^ G3_SWZ(brow)) << 4));     // bytes within a chunk plane
    // fat waves (4 per work-group): a wave also copies row block wave + 4 (16 rows = 1 KB further; the swizzle, a function of
    // (row >> 2) & 3, is the same)
    const uint32_t b_off2 = (uint32_t)(min(nb * BN + brow + 64, Np - 1) * 64 + (((lane & 3) ^ G3_SWZ(brow)) << 4));
    const unsigned long long b_chunk = (unsigned long long)Np * 64, b_plane = args.plane_elems * 2;   // bytes
    unsigned long long s_ptr = (unsigned long long)args.wpk;   // weight stream: plane 0 of the next chunk (scalar)
    int s_kc = 0;
    const uint32_t b_dst = smem_base + G3_LUT_BYTES + AW_BYTES + (wave % RBLK) * 1024;      // + slot offset, plane
    auto stage_next = [&](uint32_t slot_off) __attribute__((always_inline)) {
        const uint32_t d = __builtin_amdgcn_readfirstlane(b_dst + slot_off);                // wave-uniform LDS address
        if (WN == 2) {
            glds16x3(sgpr_ptr(reinterpret_cast<const void*>(s_ptr)), sgpr_ptr(reinterpret_cast<const void*>(s_ptr + b_plane)),
                     sgpr_ptr(reinterpret_cast<const void*>(s_ptr + 2 * b_plane)), b_off,
                     d, d + BN * G3_ROWB, d + 2 * BN * G3_ROWB);
            if (NW == 4)
                glds16x3(sgpr_ptr(reinterpret_cast<const void*>(s_ptr)), sgpr_ptr(reinterpret_cast<const void*>(s_ptr + b_plane)),
                         sgpr_ptr(reinterpret_cast<const void*>(s_ptr + 2 * b_plane)), b_off2,
                         d + 4096, d + 4096 + BN * G3_ROWB, d + 4096 + 2 * BN * G3_ROWB);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int piece = wave + 8 * j;                 // wave-uniform
                if (piece < NPIECE) {
                    const int pl = piece / RBLK;
                    glds16(sgpr_ptr(reinterpret_cast<const void*>(s_ptr + pl * b_plane)), b_off, d + pl * (BN * G3_ROWB));
                }
            }
        }
        s_ptr += b_chunk;
        if (++s_kc == Kc) { s_kc = 0; s_ptr = (unsigned long long)args.wpk; }
    };

    // 16x16 tiles: lane holds column lane&15, rows (lane>>4)*4 + reg of each tile
    const int lr = lane & 15, lg = lane >> 4;
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // A fragment: row (wm*64 + mt*16 + lr), k = 8*lg .. 8*lg+7  ->  table[byte lg of the row's spike word]
    const unsigned char* const w_rd = ring + (wm * WROWS + lr) * 4;                          // + slot offset, mt*64
    const int lg8 = 8 * lg;
    // the lane's byte of the row word is read as a byte (ds_read_u8; the four k-groups of a row hit one dword: no conflict):
    // fragment address = byte << 4, one shift instead of bfe + shift
    auto rd_w = [&](uint32_t slot_off, int mt) __attribute__((always_inline)) { return (uint32_t)*reinterpret_cast<const uint8_t*>(w_rd + lg + slot_off + mt * 64); };
    auto rd_a = [&](uint32_t w) __attribute__((always_inline)) { return *reinterpret_cast<const bf16x8*>(lut + (w << 4)); };
    // B fragment: row (tile*16 + lr), logical unit lg; swz depends on lr only
    const unsigned char* const b_rd = ring + AW_BYTES + (wn * 64 + lr) * G3_ROWB + ((lg ^ G3_SWZ(lr)) << 4);
    // group g of a chunk = (N-tile g/3, plane 2 - g%3): per accumulator the small terms first (lo, mid, hi)
    auto rd_b = [&](uint32_t slot_off, int g) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(b_rd + slot_off + (2 - g % 3) * (BN * G3_ROWB) + (g / 3) * 16 * G3_ROWB);
    };

    // LIF state of the fused variant, whole T loop: v (64 registers) and the synaptic current i (48 registers;
    // the 16 values of M-tile 3 live in LDS, private to the thread, touched once per time step - they are what
    // keeps the main loop free of scratch spills)
    f32x4 v[FUSE ? 4 : 1][FUSE ? 4 : 1], ci[FUSE ? 3 : 1][FUSE ? 4 : 1];
    f32x4* const ci_lds = reinterpret_cast<f32x4*>(smem + G3_LDS(NB, WN)) + tid;                    // [nt][512 threads]
    if (FUSE) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                v[mt][nt] = f32x4{args.p.v_leak, args.p.v_leak, args.p.v_leak, args.p.v_leak};
                if (mt < 3) ci[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                else ci_lds[nt * 512] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
    }
    const int n_total = n_steps * Kc;

    // ---- LIF step of the register-fused variant on the accumulators (then cleared): called after the K loop of every
    // current step, and with zero currents for the steps after the last one (dead time steps, Gemm3Args.Tc)
    auto lif_reg_step = [&](const int t) __attribute__((always_inline)) {
        // ---- LIF epilogue in registers.  A ballot over accumulator register (mt, nt, reg) holds, for each
        // of the 4 row groups rg, 16 channel bits of position mt*16 + rg*4 + reg; N-tiles (0,1) and (2,3)
        // pair up into the two 32-channel words of that position, which lane = position finally stores ----
        uint32_t my0 = 0, my1 = 0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int np = 0; np < 2; ++np) {                   // N-tile pair -> word np of the position
                f32x4 vd[2], d[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int nt = 2 * np + q;
                    if (mt < 3) {
                        lif_decay4(acc[mt][nt], v[mt][nt], ci[mt][nt], args.p, vd[q], d[q]);
                    } else {
                        f32x4 i3 = ci_lds[nt * 512];
                        lif_decay4(acc[mt][nt], v[mt][nt], i3, args.p, vd[q], d[q]);
                        ci_lds[nt * 512] = i3;
                    }
                    acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                  // two ballots live at a time
                    const bool z0 = d[0][r] > 0.0f, z1 = d[1][r] > 0.0f;
                    const unsigned long long b0 = __ballot(z0), b1 = __ballot(z1);
                    v[mt][2 * np][r] = z0 ? args.p.v_reset : vd[0][r];
                    v[mt][2 * np + 1][r] = z1 ? args.p.v_reset : vd[1][r];
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const uint32_t w = (uint32_t)((b0 >> (16 * rg)) & 0xffffull) | ((uint32_t)((b1 >> (16 * rg)) & 0xffffull) << 16);
                        // lane (mt*16 + rg*4 + r) keeps the two words of its position (rows >= M are never stored)
                        if (np == 0) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(my0) : "s"(w), "n"(mt * 16 + rg * 4 + r));
                        else         asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(my1) : "s"(w), "n"(mt * 16 + rg * 4 + r));
                    }
                }
            }
        {
            const int row = m0 + wm * 64 + lane;                  // lane = position within the wave's 64 rows
            const int word0 = (nb * BN + wn * 64) >> 5;
            uint32_t* dst = args.spk + (size_t)t * args.spk_stride + (size_t)row * (Np >> 5) + word0;
            if (row < M) {
                if (word0 * 32 < Np) dst[0] = my0;
                if ((word0 + 1) * 32 < Np) dst[1] = my1;
            }
        }
    };

    // Software pipeline over the chunk sequence c = (t, kc).  During chunk c:
    //   the spike word of chunk c+3 is fetched from global memory (register),
    //   the spike word of chunk c+2 (fetched during chunk c-1) and, by LDS-DMA, the weight planes of chunk c+2 go
    //   into ring slot (c-1) mod NB (spike words: slot of chunk c+2),
    //   the weight fragments of chunk c are read PD groups ahead of their MFMAs, the first ones of chunk c+1 and
    //   its spike words / table fragments at the end of chunk c (that slot has been complete since the last barrier).
    // One barrier per chunk; s_sched_barrier pins one fragment read + 4 MFMAs per group.
    {
#pragma unroll
        for (int j = 0; j < NB - 1; ++j) { stage_a(j * SLOT); stage_next(j * SLOT); }      // chunks 0 .. NB-2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the table
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
#ifdef SNN_EXP_TIMELINE
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_loop0) :: "memory");
#endif
#ifdef SNN_EXP_CLOCK       // diagnostic build: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the main loop
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (TILE && tid == 0) {
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk_c0), "=s"(clk_r0) :: "memory");
    }
#endif
    // The loop body is unrolled over one full period of the ring (NB slots) and of the fragment double buffer (2): every
    // LDS address in it is a base register + an immediate, no ring bookkeeping is left at run time.
    // Slots (byte offsets) of chunk c, of c+1, and of c+NB-1 (which receives its spike words and weight planes now).
    // MTA = M-tiles this wave multiplies: MT, or MT - 1 for the "short" waves of a tile with fewer rows (Gemm3Args.short_wm:
    // the last 16 rows of those waves do not exist; wave-uniform choice between two instances of the same loop, same
    // number of chunks and barriers in both).
    constexpr int UNR = (NB % 2) ? 2 * NB : NB;
    int kc = 0, t = 0;
    auto chunk_loop = [&](auto mta_c) __attribute__((always_inline)) {
        constexpr int MTA = decltype(mta_c)::value;
        bf16x8 af[2][MTA], bq[RING];
        uint32_t wq[MTA];
#pragma unroll
        for (int mt = 0; mt < MTA; ++mt) wq[mt] = rd_w(0, mt);
#pragma unroll
        for (int mt = 0; mt < MTA; ++mt) af[0][mt] = rd_a(wq[mt]);
#pragma unroll
        for (int g = 0; g < PD; ++g) bq[g] = rd_b(0, g);
        for (int c0 = 0; c0 < n_total; c0 += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (c0 + u >= n_total) break;
                const uint32_t o_cur = (uint32_t)((u % NB) * SLOT), o_nxt = (uint32_t)(((u + 1) % NB) * SLOT),
                               o_wr = (uint32_t)(((u + NB - 1) % NB) * SLOT);
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    const int gp = g + PD;
                    bq[gp % RING] = gp < 12 ? rd_b(o_cur, gp) : rd_b(o_nxt, gp - 12);
                    constexpr int AG0 = MT == 8 ? 4 : 8, WG0 = MT == 8 ? 0 : 4;      // groups that read the next chunk's A side
                    if (g == WG0) {
#pragma unroll
                        for (int mt = 0; mt < MTA; ++mt) wq[mt] = rd_w(o_nxt, mt);
                    }
                    if (g >= AG0 && g - AG0 < MTA) af[(u & 1) ^ 1][g - AG0] = rd_a(wq[g - AG0]);
                    if (g == 2) {
                        stage_a(o_wr);
#ifndef SNN_EXP_NO_GLDS
                        stage_next(o_wr);
#endif
                    }
#pragma unroll
                    for (int mt = 0; mt < MTA; ++mt)
                        acc[mt][g / 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][mt], bq[g % RING], acc[mt][g / 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifndef SNN_EXP_NO_BARRIER
                // weight planes landed (vmcnt), spike words written (lgkmcnt), everyone done reading slot `sl`.
                // The s_waitcnt builtin (not inline asm) so that hipcc's own wait-count bookkeeping knows the prefetched
                // fragments have arrived; the empty asm statements are compiler fences (neither builtin orders memory
                // accesses for hipcc, which otherwise moves LDS reads across the barrier).
                asm volatile("" ::: "memory");
                // NB = 4: this chunk's own copies (the youngest vector-memory operations of the wave: its spike-word copy if it
                // has rows to stage + 3 weight pieces, or 2 / 1 on the 8 x 1 wave grid) may stay in flight across the barrier
                if (NB == 3) __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0) lgkmcnt(0)
                else if (WN == 2) { if (a_role) __builtin_amdgcn_s_waitcnt(0x0074); else __builtin_amdgcn_s_waitcnt(0x0073); }
                else if (wave + 8 < NPIECE) { if (a_role) __builtin_amdgcn_s_waitcnt(0x0073); else __builtin_amdgcn_s_waitcnt(0x0072); }
                else { if (a_role) __builtin_amdgcn_s_waitcnt(0x0072); else __builtin_amdgcn_s_waitcnt(0x0071); }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#endif
                const bool step_done = ++kc == Kc;
                if (step_done) kc = 0;
                if (FUSE && step_done) {
                    lif_reg_step(t);
                    ++t;
                }
            }
        }
    };
    if constexpr (TILE && MT != 8) {
        if (short_wave) chunk_loop(std::integral_constant<int, MT - 1>{});
        else chunk_loop(std::integral_constant<int, MT>{});
    } else {
        chunk_loop(std::integral_constant<int, MT>{});
    }
    if (FUSE) {
        for (; t < args.T; ++t) lif_reg_step(t);
        return;
    }
#ifdef SNN_EXP_TIMELINE
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_loop1) :: "memory");
#endif
#ifdef SNN_EXP_CLOCK
    if (TILE && tid == 0) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        unsigned long long* o = reinterpret_cast<unsigned long long*>(args.spk + (size_t)args.T * args.spk_stride) + (size_t)bid * 2;
        o[0] = c1 - clk_c0; o[1] = r1 - clk_r0;
    }
#endif
    if (TILE) {
        // ---- LIF over the T time steps held in this tile.  The accumulators are the complete input currents
        // cur[t][position][column] of pb positions; in two passes of CG = 32*WN columns they go through LDS (the ring
        // is free now), where each thread runs neurons over t and the wave ballot of a step is the spike word(s):
        //   WN = 2 (CG = 64): lane = column, wave w takes positions w, w+8, ...; ballot = the word pair of (t, position)
        //   WN = 1 (CG = 32): lane = (position parity, column), wave w takes the position pairs; ballot = one word of
        //                     the even position (low half) and one of the odd position (high half)
        constexpr int CG = G3_TILE_CG(WN), PITCH = CG + 4;
        float* const tile = reinterpret_cast<float*>(smem);
        uint32_t* const pos_cnt = reinterpret_cast<uint32_t*>(smem + G3_TILE_BYTES(WN));    // behind the tile image
        const bool counting = args.cnt_img != nullptr || args.cnt_row != nullptr;
        const int pb = args.pb, T = args.T, t0 = args.t0, t1 = args.t0 + args.Tc;      // currents of steps t0 .. t1-1 are in the tile
        // input current of step t for the neuron whose tile column starts at src (row group g at src + g * pb * PITCH): +0 outside
        // the window; the row group of the step, or - period planes - the sum of the row groups of the divisors of t + 1
        const int group_stride = pb * PITCH;
        const bool periods = args.periods != 0;
        // (period planes: u_1 - row group 0 - is a term of every step, u_2 of every second, u_3 of every third: the caller reads the three
        // once per neuron and passes them in; the larger divisors, one or two steps each, are read where they are needed)
        auto tile_current = [&](const float* src, const int t, const float u1, const float u2, const float u3) __attribute__((always_inline)) {
            float cur = 0.0f;
            if (t >= t0 && t < t1) {
                if (!periods) {
                    cur = src[(size_t)(t - t0) * group_stride];
                } else {
                    uint32_t m = __builtin_amdgcn_readfirstlane(args.div[t]);        // wave-uniform; divisors in ascending order
                    cur = u1;
                    if (m & 2u) cur = __fadd_rn(cur, u2);
                    if (m & 4u) cur = __fadd_rn(cur, u3);
                    m &= ~7u;
                    while (m) {
                        const int g = __builtin_ctz(m);
                        m &= m - 1;
                        cur = __fadd_rn(cur, src[(size_t)g * group_stride]);
                    }
                }
            }
            return cur;
        };
        // The first G3_NPRE steps of a neuron run with NO memory access inside the recurrence: the row groups they need are requested
        // from LDS up front (one latency instead of one per step - the recurrence used to wait for an LDS read and, in period mode, a
        // scalar load of the divisor mask in every step, ~250 cycles each, while the matrix pipe had only the co-resident work-group's
        // waves to run), the period sums run over the compile-time divisors of t + 1 in ascending order (the order of tile_current:
        // bit-identical), and the steps are unrolled behind wave-uniform guards.  Steps from G3_NPRE on (T > 16: outside the paper's
        // grid) take the step-by-step form - more prefetched steps would not fit beside the accumulators that wait for the second
        // column pass.  Lane t keeps the ballot of step t.
        constexpr int NPRE = 16;
        auto lif_neuron = [&](const float* src, uint32_t& my0, uint32_t& my1, uint32_t& cnt_lo, uint32_t& cnt_hi, auto count_c) __attribute__((always_inline)) {
            constexpr bool COUNT = decltype(count_c)::value;  // rate counters wanted (four scalar instructions per step otherwise wasted)
            float cs[NPRE];
            float u1 = 0.0f, u2 = 0.0f, u3 = 0.0f;
            if (periods) {                                   // (t0 == 0: set_periods)
                float ug[NPRE];
#pragma unroll
                for (int g = 0; g < NPRE; ++g) ug[g] = g < t1 ? src[(size_t)g * group_stride] : 0.0f;
#pragma unroll
                for (int t = 0; t < NPRE; ++t) {
                    float c = 0.0f;
                    if (t < t1) {                            // wave-uniform: a scalar branch, not a select per step
                        c = ug[0];
#pragma unroll
                        for (int n = 2; n <= t + 1; ++n)
                            if ((t + 1) % n == 0) c = __fadd_rn(c, ug[n - 1]);
                    }
                    cs[t] = c;
                }
                u1 = ug[0]; u2 = ug[1]; u3 = ug[2];
            } else {
#pragma unroll
                for (int t = 0; t < NPRE; ++t) cs[t] = (t >= t0 && t < t1) ? src[(size_t)(t - t0) * group_stride] : 0.0f;
            }
            // Every vector instruction of the epilogue is matrix-pipe time lost (a SIMD runs either; the epilogue's ~1800 instructions per
            // wave and tile ARE the kernel's 15 % of idle pipe, tools/wg_timeline.py), so the steps are as short as exactness allows:
            //  * step 0 starts from v = v_leak, i = 0: v_dec = v_leak + ca*((v_leak - v_leak) + 0) = v_leak and i = (0 + cb*0) + cur_0
            //    = 0 + cur_0 exactly, no spike unless v_leak - v_th > 0 (wave-uniform; then the general step runs);
            //  * v_leak == 0 (Norse's default, the reference's): (0 - v) + i == i - v and (v_dec - v_th > 0) == (v_dec > v_th), both
            //    exact in IEEE arithmetic with gradual underflow (the encoders' identities, snn_common.h) - 8 instead of 10 operations.
            float vv = args.p.v_leak, ii = 0.0f;
            const bool quiet0 = !(__fsub_rn(args.p.v_leak, args.p.v_th) > 0.0f);
            const bool zr = args.p.v_leak == 0.0f;
            auto steps_from = [&](auto zr_c, auto first_c) __attribute__((always_inline)) {
                constexpr bool ZR = decltype(zr_c)::value;
                constexpr int FIRST = decltype(first_c)::value;
#pragma unroll
                for (int t = FIRST; t < NPRE; ++t) {
                    if (t < T) {
                        bool z;
                        if (ZR) {
                            const float v_dec = __fadd_rn(vv, __fmul_rn(args.p.ca, __fsub_rn(ii, vv)));
                            const float i_dec = __fadd_rn(ii, __fmul_rn(args.p.cb, ii));
                            z = v_dec > args.p.v_th;
                            vv = z ? args.p.v_reset : v_dec;
                            ii = __fadd_rn(i_dec, cs[t]);
                        } else {
                            z = lif_step(cs[t], vv, ii, args.p);
                        }
                        const unsigned long long b = __ballot(z);
                        G3_KEEP_BALLOT(my0, my1, b, t);
                        if (COUNT) {
                            cnt_lo += __popc((uint32_t)b);
                            cnt_hi += __popc((uint32_t)(b >> 32));
                        }
                    }
                }
            };
            if (quiet0 && T > 0) {
                ii = __fadd_rn(0.0f, cs[0]);                  // (step 0: no spike, v stays v_leak)
                if (zr) steps_from(std::true_type{}, std::integral_constant<int, 1>{});
                else steps_from(std::false_type{}, std::integral_constant<int, 1>{});
            } else {
                steps_from(std::false_type{}, std::integral_constant<int, 0>{});
            }
            for (int t = NPRE; t < T; ++t) {
                const bool z = lif_step(tile_current(src, t, u1, u2, u3), vv, ii, args.p);
                const unsigned long long b = __ballot(z);
                my0 = lane == t ? (uint32_t)b : my0;
                my1 = lane == t ? (uint32_t)(b >> 32) : my1;
                if (COUNT) {
                    cnt_lo += __popc((uint32_t)b);
                    cnt_hi += __popc((uint32_t)(b >> 32));
                }
            }
        };
        // A given (T, window) as straight-line code: no step / row-group
        // guards, no counters (the general form spends two scalar instructions per guard and four per step on the rate counts, and a
        // wave issues one instruction - scalar or vector - per four cycles at best).  Period planes, v_leak == 0, no spike at step 0.
        auto lif_neuron_fixed = [&](const float* src, uint32_t& my0, uint32_t& my1, auto ts_c, auto tcs_c) __attribute__((always_inline)) {
            constexpr int TS = decltype(ts_c)::value, TCS = decltype(tcs_c)::value;
            float ug[TCS];
#pragma unroll
            for (int g = 0; g < TCS; ++g) ug[g] = src[(size_t)g * group_stride];
            float vv = 0.0f, ii = 0.0f;
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                float c = 0.0f;
                if (t < TCS) {
                    c = ug[0];
#pragma unroll
                    for (int n = 2; n <= t + 1; ++n)
                        if ((t + 1) % n == 0) c = __fadd_rn(c, ug[n - 1]);
                }
                if (t == 0) { ii = __fadd_rn(0.0f, c); continue; }
                const float v_dec = __fadd_rn(vv, __fmul_rn(args.p.ca, __fsub_rn(ii, vv)));
                const float i_dec = __fadd_rn(ii, __fmul_rn(args.p.cb, ii));
                const bool z = v_dec > args.p.v_th;
                vv = z ? args.p.v_reset : v_dec;
                ii = __fadd_rn(i_dec, c);
                const unsigned long long b = __ballot(z);
                G3_KEEP_BALLOT(my0, my1, b, t);
            }
        };
        // (instantiated for T = 4 ... 16 with the window this kernel's launches use without rate outputs: conv T - 1 steps, fc6 T - 2 -
        // the paper's grid, metrics_for_different_timesteps.py:30-33; anything else takes the general form)
        constexpr int FIXED_D = CONV ? 1 : 2;
        const bool fixed_ok = periods && !counting && args.p.v_leak == 0.0f && !(__fsub_rn(args.p.v_leak, args.p.v_th) > 0.0f) && t0 == 0;
        const bool fixed_cfg = fixed_ok && !args.epi_general && T >= 4 && T <= 16 && t1 == T - FIXED_D;     // block-uniform
        auto lif_neuron_fixed_T = [&](const float* src, uint32_t& my0, uint32_t& my1) __attribute__((always_inline)) {
            switch (T) {
#define G3_FIXED_CASE(n) case n: lif_neuron_fixed(src, my0, my1, std::integral_constant<int, n>{}, std::integral_constant<int, n - FIXED_D>{}); break;
                G3_FIXED_CASE(4) G3_FIXED_CASE(5) G3_FIXED_CASE(6) G3_FIXED_CASE(7) G3_FIXED_CASE(8) G3_FIXED_CASE(9) G3_FIXED_CASE(10)
                G3_FIXED_CASE(11) G3_FIXED_CASE(12) G3_FIXED_CASE(13) G3_FIXED_CASE(14) G3_FIXED_CASE(15) G3_FIXED_CASE(16)
#undef G3_FIXED_CASE
            default: break;
            }
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // staged-ahead copies of chunks past the end have landed
#ifdef SNN_EXP_TIMELINE
        unsigned long long tl_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TL_STAMP(i) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_ph[i]) :: "memory")
        TL_STAMP(0);
#else
#define TL_STAMP(i)
#endif
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            __syncthreads();                               // ring reads done / previous pass consumed
            if (h == 0) TL_STAMP(1); else TL_STAMP(4);
            if (WN == 1 || wn == h) {
                const int lrow0 = wm * WROWS - 16 * max(0, wm - first_short);    // logical row of the wave's first row
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (mt == MT - 1 && short_wave) continue;                   // (rows that do not exist)
#pragma unroll
                    for (int nq = 0; nq < CG / 16; ++nq) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float val;
                            if (WN == 1) val = h == 0 ? acc[mt][nq][r] : acc[mt][2 + nq][r];
                            else val = acc[mt][nq][r];
                            tile[(lrow0 + mt * 16 + lg * 4 + r) * PITCH + nq * 16 + lr] = val;
                        }
                    }
                }
            }
            __syncthreads();
            if (h == 0) TL_STAMP(2); else TL_STAMP(5);
            const int word0 = (nb * BN + h * CG) >> 5;     // first output word of this pass
            if (word0 * 32 >= Np) continue;                // block-uniform
            if (WN == 2) {
                const bool two = (word0 + 1) * 32 < Np;
                for (int pi = wave; pi < pb; pi += NW) {   // wave-uniform
                    const int pos = m0 + pi;
                    if (pos >= M) break;
                    uint32_t my0 = 0, my1 = 0;             // lane t keeps the word pair of time step t
                    const float* src = tile + pi * PITCH + lane;
                    uint32_t cnt = 0;                      // wave-uniform: spikes of this position in this pass
                    uint32_t cnt_hi = 0;
                    if (fixed_cfg) lif_neuron_fixed_T(src, my0, my1);
                    else if (counting) lif_neuron(src, my0, my1, cnt, cnt_hi, std::true_type{});
                    else lif_neuron(src, my0, my1, cnt, cnt_hi, std::false_type{});
                    if (two) cnt += cnt_hi;
                    if (lane < T) {
                        if (!CONV && args.out_wm) {                    // word-major planes [T][word][row] (fc6 -> fc7)
                            uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + (size_t)word0 * M + pos;
                            dst[0] = my0;
                            if (two) dst[M] = my1;
                        } else if (CONV && args.out_split) {
                            uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + ((size_t)(word0 >> 2) * M + pos) * 4 + (word0 & 3);
                            dst[0] = my0;
                            if (two) dst[1] = my1;                      // word0 is even: the pair stays inside its block of four
                        } else {
                            uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + (size_t)pos * (Np >> 5) + word0;
                            dst[0] = my0;
                            if (two) dst[1] = my1;
                        }
                    }
                    if (counting && lane == 0) pos_cnt[pi] = (h == 0 ? 0u : pos_cnt[pi]) + cnt;
                }
            } else {
                const int par = lane >> 5, col = lane & 31;
                for (int pp = wave; 2 * pp < pb; pp += 8) {            // wave-uniform: position pair pp
                    const int pi = 2 * pp + par;                       // this half-wave's position
                    const bool live = pi < pb && m0 + pi < M;
                    if (m0 + 2 * pp >= M) break;
                    uint32_t my0 = 0, my1 = 0;             // lane t keeps the words of (t, even position), (t, odd position)
                    const float* src = tile + (live ? pi : 2 * pp) * PITCH + col;
                    uint32_t cnt0 = 0, cnt1 = 0;           // wave-uniform: spikes of the even / odd position in this pass
                    if (fixed_cfg) lif_neuron_fixed_T(src, my0, my1);
                    else if (counting) lif_neuron(src, my0, my1, cnt0, cnt1, std::true_type{});
                    else lif_neuron(src, my0, my1, cnt0, cnt1, std::false_type{});
                    const bool odd_ok = 2 * pp + 1 < pb && m0 + 2 * pp + 1 < M;
                    if (lane < T) {
                        if (!CONV && args.out_wm) {
                            uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + (size_t)word0 * M + (m0 + 2 * pp);
                            dst[0] = my0;
                            if (odd_ok) dst[1] = my1;
                        } else if (CONV && args.out_split) {
                            uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + ((size_t)(word0 >> 2) * M + m0 + 2 * pp) * 4 + (word0 & 3);
                            dst[0] = my0;
                            if (odd_ok) dst[4] = my1;
                        } else {
                            uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + (size_t)(m0 + 2 * pp) * (Np >> 5) + word0;
                            dst[0] = my0;
                            if (odd_ok) dst[Np >> 5] = my1;
                        }
                    }
                    if (counting && lane == 0) {
                        pos_cnt[2 * pp] = (h == 0 ? 0u : pos_cnt[2 * pp]) + cnt0;
                        if (odd_ok) pos_cnt[2 * pp + 1] = (h == 0 ? 0u : pos_cnt[2 * pp + 1]) + cnt1;
                    }
                }
            }
            if (h == 0) TL_STAMP(3); else TL_STAMP(6);
        }
        if (counting) {
            __syncthreads();
            tile_counts_flush<CONV>(args, pos_cnt, m0, pb, tid);
        }
#ifdef SNN_EXP_TIMELINE
        if (tid == 0) {
            unsigned long long tl_exit;
            uint32_t hw, xcc;
            asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                         : "=s"(tl_exit), "=s"(hw), "=s"(xcc) :: "memory");
            unsigned long long* o = reinterpret_cast<unsigned long long*>(args.spk + (size_t)args.T * args.spk_stride) + (size_t)bid * 16;
            o[0] = tl_entry; o[1] = tl_loop0; o[2] = tl_loop1; o[3] = tl_exit; o[4] = hw; o[5] = xcc;
            for (int i = 0; i < 7; ++i) o[8 + i] = tl_ph[i];
        }
#endif
        return;
    }
    // ---- store currents: per instruction 4 rows x 16 columns (64-B row segments) ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nb * BN + wn * 64 + nt * 16 + lr;
            if (col >= Np) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * WROWS + mt * 16 + lg * 4 + r;
                if (m < M) args.out[(size_t)m * args.ldo + col] = acc[mt][nt][r];
            }
        }
}

template <int MODE, int NB, int MT, int WN>
__global__ __launch_bounds__(MT == 8 ? 256 : 512, (MODE == G3_CONV_LIF_REG || MT == 8) ? 2 : 4) void k_gemm_bf16x3(const Gemm3Args args) {
    gemm_bf16x3_body<MODE, NB, MT, WN>(args, (int)blockIdx.x);
}

// fp32 weights -> three bf16 planes [3][Kc][Np][32]  (hi = rn(w), mid = rn(w - hi), lo = rn(w - hi - mid): exact)
__device__ __forceinline__ uint16_t f2bf_rn(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }

// Is the three-plane split of w exact, i.e. does (lo + mid) + hi - the order the kernels accumulate the planes in - give back w?
// It does for every finite fp32 value whose lowest set mantissa bit is >= 2^-133 (the finest bf16 subnormal) and whose high plane
// does not round up to infinity (|w| < 2^128 - 2^119): 3 x 8 significant bits cover the 24 of an fp32, and each residual is exact
// (Sterbenz).  What is left: fp32 subnormals / tiny normals with bits below 2^-133 (inexact), values next to FLT_MAX (hi = inf),
// and non-finite weights (the integer rounding of f2bf_rn turns a NaN into +-inf or 0).  status[0] += inexact finite weights,
// status[1] += non-finite weights, status[2] += weights with a plane that is a non-zero bf16 SUBNORMAL (exact, but only if the
// matrix cores keep subnormal operands: tests/test_gpu_stages.py::test_one_hot_spike_gemm_returns_the_weights_bitwise probes that).
__global__ void k_bf16x3_split_check(const float* __restrict__ w, size_t n, uint32_t* __restrict__ status) {
    uint32_t inexact = 0, nonfinite = 0, subnormal = 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const float x = w[idx];
        if ((__float_as_uint(x) & 0x7f800000u) == 0x7f800000u) { ++nonfinite; continue; }
        const uint16_t hi = f2bf_rn(x);
        const float r1 = __fsub_rn(x, bf2f(hi));
        const uint16_t mid = f2bf_rn(r1);
        const float r2 = __fsub_rn(r1, bf2f(mid));
        const uint16_t lo = f2bf_rn(r2);
        const float back = __fadd_rn(__fadd_rn(bf2f(lo), bf2f(mid)), bf2f(hi));
        if (!(back == x)) ++inexact;
        auto sub = [](uint16_t b) { return (b & 0x7f80u) == 0 && (b & 0x007fu) != 0; };
        if (sub(hi) || sub(mid) || sub(lo)) ++subnormal;
    }
    if (inexact) atomicAdd(status + 0, inexact);
    if (nonfinite) atomicAdd(status + 1, nonfinite);
    if (subnormal) atomicAdd(status + 2, subnormal);
}

__global__ void k_pack_bf16x3(const float* __restrict__ src, uint16_t* __restrict__ dst, int mode, int K, int N,
                              int Kc, int Np, int Cin, int Cp) {
    const size_t plane = (size_t)Kc * Np * 32;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < plane; idx += (size_t)gridDim.x * blockDim.x) {
        const int kk = idx & 31;
        const size_t rest = idx >> 5;
        const int n = (int)(rest % Np);
        const int kc = (int)(rest / Np);
        const int k = kc * 32 + kk;
        float w = 0.0f;
        if (n < N) {
            if (mode == PACK_CONV3X3) {
                const int tap = k / Cp, ci = k % Cp;
                if (ci < Cin) w = src[((size_t)n * Cin + ci) * 9 + tap];
            } else if (mode == PACK_LINEAR_PERM) {         // reduction index k' = s * Cp + c  <-  source column c * Cin + s  (Cin = S inner
                if (k < K) w = src[(size_t)n * K + (size_t)(k % Cp) * Cin + k / Cp];        // elements per channel, Cp = channels)
            } else if (k < K) {
                w = src[(size_t)n * K + k];
            }
        }
        const uint16_t hi = f2bf_rn(w);
        const float r1 = __fsub_rn(w, bf2f(hi));
        const uint16_t mid = f2bf_rn(r1);
        const float r2 = __fsub_rn(r1, bf2f(mid));
        const uint16_t lo = f2bf_rn(r2);
        dst[idx] = hi; dst[plane + idx] = mid; dst[2 * plane + idx] = lo;
    }
}
