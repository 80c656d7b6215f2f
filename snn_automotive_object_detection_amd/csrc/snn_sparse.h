// Structured-sparse matrix cores for the SPARSE period planes of the RPN's shared 3x3 convolution and the detector's fc6 (rounds 4-5).  Included by snn_kernels.hip
// after snn_bf16x3.h (Gemm3-style staging helpers, f2bf / bf2f, G3_SWZ).
//
// With period planes (snn_common.h) a conv row tile holds Tc row groups u_n = W e_n, and for n >= 3 the planes are nearly empty
// (densities 0.039, 0.019, 0.011, 0.007, 0.005 on the bench's pyramid) - yet each costs a full dense MFMA pass.  gfx950's
// v_smfmac_f32_16x16x64_bf16 multiplies an A operand with at most TWO non-zeros per four consecutive k at 64 k per instruction:
// 1.82 x the dense rate sustained in this kernel's LDS-fed shape (tools/sparse_probe.hip, profiles/r4_sparse_probe.txt).  So:
//   * planes e_1, e_2 (densities 0.25 / 0.12: nibbles with three spikes are common) stay on the dense v_mfma_f32_16x16x32_bf16;
//   * planes e_n, n >= 3, are COMPRESSED (by the encoder launches themselves since round 5 - snn_encode.h -, else by k_compress_planes): per 16 k of a row 8 value slots (occupied or not: the value is 1.0) +
//     8 two-bit positions.  The A fragment of a lane is table[occupancy byte] - the dense kernels' byte -> 8 bf16 table - plus 16 index
//     bits; operand layout and encoding were established on the hardware (sparse_probe A1 / A4);
//   * a nibble with three or four spikes of ONE period (rare: ~0.16 per position over the five sparse planes) keeps its first two in
//     the compressed plane; the others go into a SECONDARY compressed plane (two more value slots per nibble, at the constant
//     positions 2 and 3 - where a third / fourth spike can only be: one more dword per row and step, all four bits of a nibble
//     covered).  A wave whose 16 rows have an empty secondary block in a 64-k step (the rule: ~98 % of the (M-tile, step) pairs on the
//     bench's pyramid) skips it; otherwise it issues the structured-sparse instruction a second time for
//     that M-tile, right behind the step's other products.  Exactness is unchanged: every product is spike x (hi + mid + lo), fp32 sums
//     in a fixed order.  (Until late in round 4 the third and fourth spikes were per-tile fix-up lists applied in the epilogue, with a
//     dense fallback launch for inputs that overflowed them: 3.7 % of the conv launch, and a slow path for adversarial inputs.  The
//     secondary plane needs neither lists nor fallback, and reads its entries through the same 3x3 tap walk as everything else.)
//
// Row tile = pb positions x Tc planes as M-tiles of 16 rows (plane t, positions 16 j .. 16 j + 15), assigned to M-tile SLOTS of the
// waves (table in SparseConvArgs, dense ones first); a wave's K loop is instantiated for its (dense, sparse) counts.  K runs in steps
// of 64 (two 32-deep chunks of the packed weights, whose LDS image is the dense kernels'); ring = two step slots, one barrier per step,
// two work-groups per CU in every shape.  Shapes (k_gemm_lif_sparse<CONV, WN, FAT>; the launcher's planner picks, snn_kernels.hip):
//   8-wave (512 threads, 128 registers)    8 x 1 waves x 4 slots (conv) or 4 x 2 x 6 (linear layers); LIF through a tile image in LDS,
//                                          straight-line instances T = 5 .. 16, general (run-time) form for the linear layers beyond
//   FAT (round 5; 256 threads, <= 256 regs) four waves with twice the slots: half (conv 2 x 2: a quarter) of the weight-fragment reads per
//                                          matrix instruction, fragments requested two groups ahead.  Linear layers: 2 x 2 x 12.  Conv:
//                                          4 x 1 x 8 for T = 7 .. 9 (tiles of 64 positions), 2 x 2 x 16 for T = 12 .. 16 (tiles of 32) - every
//                                          (row-)wave holds ALL planes of its own 16 positions, so the LIF runs in REGISTERS (sp_lif_regs):
//                                          no tile image, no epilogue barrier; also for fc6 where a row-wave holds all planes of its RoIs
// All shapes give the same bits: every accumulator sees the same matrix instructions in the same order, and the LIF forms are the same
// operations in the same order (tests/test_gpu_sparse.py compares the spike planes in the workspace).
#pragma once

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfv16 __attribute__((ext_vector_type(16)));

#define SP_MT 4                                     // M-tile slots per wave on the 8 x 1 wave grid (all waves span the 64 columns)
#define SP_MT2 6                                    // ... per ROW-wave on the 4 x 2 grid (two waves of 32 columns share a row-wave's slots)
#define SP_MT2_FAT 12                               // ... per row-wave of the FAT shape (linear layers): 2 x 2 waves of up to 256 registers, 32 columns per wave
#define SP_MT_FAT 8                                 // ... per wave of the FAT conv, T <= 9: 4 waves x all 64 columns, every wave ALL planes of its own 16 positions (register LIF)
#define SP_MT2_FAT_CONV 16                          // ... per row-wave of the FAT conv, T = 12 .. 16: 2 x 2 waves, a row-wave = all planes of its 16 positions, 32 columns per wave
#define SP_MTMAX 16
#define SP_ROWS (8 * SP_MT * 16)                    // physical tile rows (512; 4 x 6 x 16 = 384 on the 4 x 2 grid)
// (SP_A_ARR and sp_nibble_code live in snn_common.h: the RPN encoder writes the same compressed layout)
#define SP_A_BYTES (SP_A_ARR * SP_ROWS * 4)
#define SP_B_BYTES (2 * 3 * 64 * G3_ROWB)           // two chunks x three weight planes x 64 columns
#define SP_SLOT (SP_A_BYTES + SP_B_BYTES)           // one 64-k step: 32 KB
#define SP_LDS (G3_LUT_BYTES + 2 * SP_SLOT)         // 68 KB: two work-groups per CU
#ifdef SNN_EXP_SP_NO_BREAD                            // (timing experiment, see the K loop)
#define SNN_EXP_BSEL(g) 0
#else
#define SNN_EXP_BSEL(g) (g)
#endif
#ifndef SP_PRE_A
#define SP_PRE_A 1
#endif
#ifndef SP_PRE_B
#define SP_PRE_B 0                                  // (1: FAT shapes request a step's first weight fragments right behind the previous step's barrier -
                                                    // measured 0.9 % / 1.2 % SLOWER on the conv / fc6, profiles/r5_preb_ab.txt: off)
#endif
#ifndef SP_SEC_LATE
#define SP_SEC_LATE 1
#endif
#define SP_PITCH 36                                 // epilogue tile image: 32 columns + 4 floats of padding per row

struct SparseConvArgs {
    const uint32_t* enc;         // raw period planes, word-major [Tc][Cw][Pe] (zero halo); the dense planes are read from here
    const uint32_t* cmp;         // compressed planes [Tc - nd][Cw / 2][4][Pe]
    const uint16_t* wpk;         // [3][Kc][Np][32] bf16
    uint32_t* spk;               // spike planes out
    unsigned long long* tl;      // SNN_EXP_TIMELINE builds: 8 stamps per work-group
    // spike-rate side output (nullable; zeroed by the caller): spikes per row - RoI, or position (the conv's launcher sums them per
    // (level, image) afterwards: k_sum_pos_counts).  Integer atomics, one per (row, 32 columns): order-independent.
    uint32_t* cnt_row;
    unsigned long long plane_elems, spk_stride;
    unsigned int Pe;             // padded rows of a word plane
    int M, Kc, Np, Cw, n_blocks, n_tiles, n_levels;
    int T, Tc, nd, pb, q, out_split;
    signed char mt_plane[8][SP_MTMAX];   // plane of the row-wave's M-tile slot (-1: unused); dense planes (< nd) first
    unsigned char mt_j[8][SP_MTMAX];     // position block of the slot: local positions 16 j ..
    unsigned char w_nd[8], w_ns[8];      // dense / sparse M-tiles of the row-wave
    int xcd_contig, xcd_cpx;             // block order, as Gemm3Args
    // epi_general != 0: the LIF epilogue's general form (run-time T and window: the current of step t < Tc is the sum of the row groups in
    // div[t], ascending) - every (T, window) without a straight-line instance: linear layers at T > 16 and in spike-rate mode (window T - 1)
    int lif_regs;                // FAT shapes: every (row-)wave holds ALL planes of its own block of 16 positions / RoIs (slot s = plane s) and this
                                 // (T, window) has a register-LIF instance: the LIF runs in registers (sp_lif_regs), no tile image (host: sparse_plan_lif_regs)
    int epi_general;
    uint32_t div[SNN_MAX_STEPS];
    NeuronP p;
    ConvLevelDev lv[SNN_MAX_LEVELS];
};

// Reduction-index permutation of a linear layer's period planes (the detector's fc6).  The flattened RoI features run (channel, bin):
// k = c * S + s, so four consecutive k are four neighbouring BINS of one channel - strongly correlated values, hence often the same
// period: three or four spikes per nibble would be the rule in the sparse planes, not the exception.  With k' = s * C + c four consecutive
// k' are four CHANNELS at one bin (independent, as in the RPN's conv, whose reduction index is tap * C + channel).  The encoders keep
// writing planes in the reference's order; this kernel transposes the bits of every (plane, RoI) row, and fc6's weights are packed in
// the same order (snn_pack_linear_weight_bf16x3_perm): the contraction is the same sum in another order.
// Block = (32 RoIs, plane); word-major planes [T][Dw][R] in and out.  Thread task = (RoI, block of 32 channels): the channels' 32 x S bits are
// S words of the row, i.e. a 32 x S bit matrix (row = channel: S consecutive bits at bit offset j S) that leaves as its transpose (S words
// of 32 channel bits).  With S a compile-time constant: each channel's bits are cut out with two funnel shifts (v_alignbit) into a 32-bit
// and an (S - 32)-bit part, and the two 32 x 32 bit matrices are transposed in registers by the five-stage butterfly (Hacker's Delight
// 7-3, LSB-first form): ~1100 operations per task.  (First version: every bit gathered from LDS, 82 us for 2000 RoIs x 10 planes; second:
// one v_bfe + v_lshl_or per bit, 3136 operations per task, 38 us.)  Stores: 32 consecutive RoIs of one word.
__device__ __forceinline__ void bit_transpose32(uint32_t (&a)[32]) {          // out[s] bit j = in[j] bit s
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int jj = 16 >> st;
        const uint32_t m = st == 0 ? 0x0000ffffu : st == 1 ? 0x00ff00ffu : st == 2 ? 0x0f0f0f0fu : st == 3 ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (!(k & jj)) {
                const uint32_t t = ((a[k] >> jj) ^ a[k + jj]) & m;
                a[k] ^= t << jj;
                a[k + jj] ^= t;
            }
    }
}

#define PERM_CB 8                                   // channel blocks (of 32 channels) per pass: one per wave; LDS = PERM_CB * S * 33 words (51.7 KB at S = 49)
template <int S>
__global__ __launch_bounds__(256) void k_permute_planes(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int Dw, int R, int C) {
    static_assert(S > 32 && S <= 64, "two 32-column parts");
    __shared__ uint32_t pl[PERM_CB * S * 33];                 // [word of the pass][32 RoIs + 1]
    const int t = blockIdx.y, r0 = blockIdx.x * 32, rl = threadIdx.x & 31;
    const bool live = r0 + rl < R;
    const int cbn = C / 32;                                   // channel blocks = words per bin in the permuted order
    // passes of PERM_CB channel blocks (any channel count: round 5 - the whole row in LDS capped C at 320)
    for (int cb0 = 0; cb0 < cbn; cb0 += PERM_CB) {
        const int nw = min(PERM_CB, cbn - cb0) * S, wbase = cb0 * S;       // words [wbase, wbase + nw) of the row
        if (cb0) __syncthreads();
        // (the row's words are requested in batches of 7: one load in flight per thread made this kernel a chain of memory latencies)
        for (int w0 = threadIdx.x >> 5; w0 < nw; w0 += 8 * 7) {
            uint32_t v[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int w = w0 + 8 * i;
                v[i] = (live && w < nw) ? in[((size_t)t * Dw + wbase + w) * R + r0 + rl] : 0u;
            }
#pragma unroll
            for (int i = 0; i < 7; ++i)
                if (w0 + 8 * i < nw) pl[(w0 + 8 * i) * 33 + rl] = v[i];
        }
        __syncthreads();
        const int cl = threadIdx.x >> 5, cb = cb0 + cl;       // one channel block per wave-half
        if (cb >= cbn) continue;
        uint32_t w[S + 1];
#pragma unroll
        for (int i = 0; i < S; ++i) w[i] = pl[(cl * S + i) * 33 + rl];          // bits [32 cb S, 32 (cb + 1) S) of the row: channel j at bit j S + s
        w[S] = 0u;
        uint32_t lo[32], hi[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int o = j * S, i = o >> 5, sh = o & 31;                        // compile-time after unrolling
            lo[j] = sh ? __builtin_amdgcn_alignbit(w[i + 1], w[i], sh) : w[i];
            hi[j] = (sh ? __builtin_amdgcn_alignbit(w[i + 2 <= S ? i + 2 : S], w[i + 1], sh) : w[i + 1]) & ((1u << (S - 32)) - 1u);
        }
        bit_transpose32(lo);
        bit_transpose32(hi);
        if (live) {
#pragma unroll
            for (int s = 0; s < S; ++s)
                out[((size_t)t * Dw + (size_t)s * cbn + cb) * R + r0 + rl] = s < 32 ? lo[s] : hi[s - 32];
        }
    }
}

struct CompressArgs {
    const uint32_t* enc;
    uint32_t* cmp;
    unsigned int Pe;
    int Cw, nd;
};

// thread = (row, 64-k step w2 = blockIdx.y, sparse plane ts = blockIdx.z); row = padded position (conv) or RoI (linear layer).
// Out: four dwords per (row, step): primary occupancy / indices 0-1 / indices 2-3, then the secondary plane's occupancy (the third and
// fourth spike of a nibble: zero almost everywhere).  A third spike can only sit at bit 2 or 3 of its nibble and a fourth at bit 3, so
// the secondary slots are (value = leftover bit 2, position 2), (value = leftover bit 3, position 3): constant indices, nothing stored.
__global__ __launch_bounds__(256) void k_compress_planes(const CompressArgs a) {
    __shared__ uint16_t code[256];
    code[threadIdx.x] = sp_byte_code(threadIdx.x);
    __syncthreads();
    const unsigned int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= a.Pe) return;
    const int w2 = blockIdx.y, ts = blockIdx.z, t = a.nd + ts;
    uint32_t c4[4];
    sp_compress_pair(a.enc[((size_t)t * a.Cw + 2 * w2) * a.Pe + row], a.enc[((size_t)t * a.Cw + 2 * w2 + 1) * a.Pe + row], code, c4);
    uint32_t* out = a.cmp + ((size_t)ts * (a.Cw / 2) + w2) * SP_A_ARR * a.Pe + row;
    const size_t Pe = a.Pe;
    out[0] = c4[0]; out[Pe] = c4[1]; out[2 * Pe] = c4[2];
    out[3 * Pe] = c4[3];
}

// LIF over T steps of NP independent neurons per lane from the period sums in the LDS tile image: the straight-line form of
// k_gemm_bf16x3's epilogue (period planes, v_leak = 0, no spike at step 0, conv window T - 1), same operations in the same order per
// neuron.  NP = 2: the recurrence is one dependent chain of ~6 operations per step; two of them interleaved fill each other's latencies
// (LIF part of an epilogue pass 3.9 -> see profiles/r4_sparse_timeline.txt).
// COUNT: also the spikes per position (low / high half of the ballot = even / odd position of the pair): scalar popcounts.
template <int TS, int D, int NP, bool COUNT>
__device__ __forceinline__ void sp_lif_fixed(const float* const (&src)[NP], const int group_stride, const NeuronP& p, uint32_t (&my0)[NP], uint32_t (&my1)[NP],
                                             uint32_t (&cnt_lo)[NP], uint32_t (&cnt_hi)[NP]) {
    constexpr int TCS = TS - D;                          // currents of steps 0 .. T - 1 - D (conv: D = 1; fc6: D = 2 - dead time steps)
    float ug[NP][TCS];
#pragma unroll
    for (int u = 0; u < NP; ++u)
#pragma unroll
        for (int g = 0; g < TCS; ++g) ug[u][g] = src[u][(size_t)g * group_stride];
    float vv[NP], ii[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) { vv[u] = 0.0f; ii[u] = 0.0f; }
#pragma unroll
    for (int t = 0; t < TS; ++t) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            float c = 0.0f;
            if (t < TCS) {
                c = ug[u][0];
#pragma unroll
                for (int n = 2; n <= t + 1; ++n)
                    if ((t + 1) % n == 0) c = __fadd_rn(c, ug[u][n - 1]);
            }
            if (t == 0) { ii[u] = __fadd_rn(0.0f, c); continue; }
            const float v_dec = __fadd_rn(vv[u], __fmul_rn(p.ca, __fsub_rn(ii[u], vv[u])));
            const float i_dec = __fadd_rn(ii[u], __fmul_rn(p.cb, ii[u]));
            const bool z = v_dec > p.v_th;
            vv[u] = z ? p.v_reset : v_dec;
            ii[u] = __fadd_rn(i_dec, c);
            const unsigned long long b = __ballot(z);
            G3_KEEP_BALLOT(my0[u], my1[u], b, t);
            if (COUNT) { cnt_lo[u] += (uint32_t)__builtin_popcount((uint32_t)b); cnt_hi[u] += (uint32_t)__builtin_popcount((uint32_t)(b >> 32)); }
        }
    }
}

// The same recurrence with T and the window at run time (one neuron per lane): u_1 .. u_3 are read once (terms of every / every second /
// every third step), the larger divisors where a step needs them - the order of k_gemm_bf16x3's tile_current (ascending n), so the sums
// are those of the straight-line instances bit for bit.  For the (T, window) pairs outside sp_lif_fixed's grid: a linear layer's tile
// then holds 16 or 32 RoIs, i.e. one or two of these per wave and column pass behind a 196-step K loop.
template <bool COUNT>
__device__ __forceinline__ void sp_lif_general(const float* src, const int group_stride, const NeuronP& p, const int T, const int Tcs, const uint32_t* div,
                                               const int lane, uint32_t& my0, uint32_t& my1, uint32_t& cnt_lo, uint32_t& cnt_hi) {
    const float u1 = src[0];
    const float u2 = Tcs > 1 ? src[(size_t)group_stride] : 0.0f;
    const float u3 = Tcs > 2 ? src[(size_t)2 * group_stride] : 0.0f;
    float vv = 0.0f, ii = 0.0f;
    for (int t = 0; t < T; ++t) {
        float c = 0.0f;
        if (t < Tcs) {
            uint32_t m = __builtin_amdgcn_readfirstlane(div[t]);                   // wave-uniform; divisors in ascending order
            c = u1;
            if (m & 2u) c = __fadd_rn(c, u2);
            if (m & 4u) c = __fadd_rn(c, u3);
            m &= ~7u;
            while (m) {
                const int g = __builtin_ctz(m);
                m &= m - 1;
                c = __fadd_rn(c, src[(size_t)g * group_stride]);
            }
        }
        if (t == 0) { ii = __fadd_rn(0.0f, c); continue; }
        const float v_dec = __fadd_rn(vv, __fmul_rn(p.ca, __fsub_rn(ii, vv)));
        const float i_dec = __fadd_rn(ii, __fmul_rn(p.cb, ii));
        const bool z = v_dec > p.v_th;
        vv = z ? p.v_reset : v_dec;
        ii = __fadd_rn(i_dec, c);
        const unsigned long long b = __ballot(z);
        my0 = lane == t ? (uint32_t)b : my0;
        my1 = lane == t ? (uint32_t)(b >> 32) : my1;
        if (COUNT) { cnt_lo += (uint32_t)__builtin_popcount((uint32_t)b); cnt_hi += (uint32_t)__builtin_popcount((uint32_t)(b >> 32)); }
    }
}

// spike-rate mode of the sparse conv: spikes per (level, image) slot from the per-position counts; block = (slot, chunk of the image's
// positions), one integer atomic per block (round 5: one block per slot took 71 us for the 73 728 positions of a level-0 image)
#define POSCNT_CHUNKS 32
struct PosCountArgs { const uint32_t* cnt_pos; unsigned long long* cnt_img; int n_levels, max_n; ConvLevelDev lv[SNN_MAX_LEVELS]; };
__global__ __launch_bounds__(256) void k_sum_pos_counts(const PosCountArgs a) {
    __shared__ unsigned long long part[4];
    const int l = blockIdx.x / a.max_n, n = blockIdx.x % a.max_n;
    unsigned long long sum = 0;
    if (n < a.lv[l].N) {
        const int hw = a.lv[l].H * a.lv[l].W;
        const uint32_t* src = a.cnt_pos + a.lv[l].pos_base + (size_t)n * hw;
        for (int i = blockIdx.y * 256 + threadIdx.x; i < hw; i += 256 * POSCNT_CHUNKS) sum += src[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long tot = part[0] + part[1] + part[2] + part[3];
        if (tot) atomicAdd(a.cnt_img + blockIdx.x, tot);
    }
}

// LIF of the FAT conv in REGISTERS (round 5).  The wave's accumulators hold all TS - 1 period sums of its own 16 positions x 64 columns
// (slot s = plane s): lane (lg, lr) has, for r < 4 and nt < 4, the neuron (position 4 lg + r, column 16 nt + lr).  No tile image in LDS,
// no work-group barrier: the recurrence of sp_lif_fixed (same operations, same order of the divisor sums) runs on the four N-tiles of a
// row r at once (four independent chains), the ballot of N-tile nt holds in bits 16 lg .. 16 lg + 15 the half-word (columns 16 nt ..) of
// position 4 lg + r, and lane lr of group lg keeps the word (step 1 + (lr >> 1), columns 32 (lr & 1) ..) of its position - 16 (step, word)
// combinations per position for TS <= 9; step 0 never spikes (its plane is written as zeros).
template <int TS, int D, int NTL, int MTS_, bool COUNT>
__device__ __forceinline__ void sp_lif_regs(const f32x4 (&acc)[MTS_][NTL], const NeuronP& p, const int lane, uint32_t (&mine)[4], uint32_t (&cnt)[4]) {
    constexpr int TCS = TS - D;                          // conv: D = 1; fc6: D = 2 (dead time steps)
    static_assert(NTL == 4 || NTL == 2, "64 columns (two words per position) or 32 (one)");
    static_assert(TCS <= MTS_ && TS - 1 <= (NTL == 4 ? 8 : 16), "all planes of a block in one wave; 16 (step, word) lanes per position");
    const int sh = 16 * (lane >> 4), lr = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float vv[NTL], ii[NTL];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) { vv[nt] = 0.0f; ii[nt] = 0.0f; }
        uint32_t keep = 0, count = 0;
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            unsigned long long b[NTL];
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                b[nt] = 0;
                float c = 0.0f;
                if (t < TCS) {
                    c = acc[0][nt][r];
#pragma unroll
                    for (int n = 2; n <= t + 1; ++n)
                        if ((t + 1) % n == 0) c = __fadd_rn(c, acc[n - 1][nt][r]);
                }
                if (t == 0) { ii[nt] = __fadd_rn(0.0f, c); continue; }
                const float v_dec = __fadd_rn(vv[nt], __fmul_rn(p.ca, __fsub_rn(ii[nt], vv[nt])));
                const float i_dec = __fadd_rn(ii[nt], __fmul_rn(p.cb, ii[nt]));
                const bool z = v_dec > p.v_th;
                vv[nt] = z ? p.v_reset : v_dec;
                ii[nt] = __fadd_rn(i_dec, c);
                b[nt] = __ballot(z);
            }
            if (t == 0) continue;
            const uint32_t w0 = ((uint32_t)(b[0] >> sh) & 0xffffu) | ((uint32_t)(b[1] >> sh) << 16);
            uint32_t w1 = 0;
            if constexpr (NTL == 4) {                    // lane lr <-> (step 1 + (lr >> 1), word lr & 1)
                w1 = ((uint32_t)(b[2] >> sh) & 0xffffu) | ((uint32_t)(b[3] >> sh) << 16);
                keep = (lr >> 1) == t - 1 ? ((lr & 1) ? w1 : w0) : keep;
            } else {                                     // lane lr <-> step 1 + lr
                keep = lr == t - 1 ? w0 : keep;
            }
            if (COUNT) count += (uint32_t)__builtin_popcount(w0) + (uint32_t)__builtin_popcount(w1);     // (spike-rate mode: this position's spikes of the step)
        }
        mine[r] = keep;
        cnt[r] = count;
    }
}

// WN = waves along the 64 columns.  1: 8 row-waves x 4 slots, every wave reads the whole weight slot from LDS each step (192 KB per
// work-group and step).  2: 4 row-waves x 6 slots, a wave covers 32 columns and reads half of the slot (96 KB): the shape for launches whose
// matrix-pipe time per step is below what those LDS reads take - fc6, whose tiles hold 32 RoIs (tools: profiles/r4_sparse_timeline.txt).
// FAT (linear layers, WN = 2; round 5): the same tile run by FOUR waves - 2 x 2, twelve M-tile slots per row-wave, 256 threads with up to
// 256 registers per lane; still two work-groups per CU, i.e. two waves per SIMD from DIFFERENT work-groups.  Every weight fragment read
// from LDS feeds twice the matrix instructions (the what-if builds price those reads at 9 % of the detector head and 14 % of the conv
// launch, profiles/r5_sparse_whatif.txt), and the registers pay for a third weight-fragment buffer: fragments are requested two groups
// ahead of their matrix instructions (one group ahead the shape LOSES 2 %).  Measured (profiles/r5_fat_wave_ab.txt, same lease): detector
// head 0.780 -> 0.756 ms at T = 12.  For the conv (4 row-waves x 8 slots) the K loop gained 4.6 % and the LIF epilogue lost it again
// (one wave per SIMD and work-group issues a vector instruction every four cycles, two interleave at two): 1.955 -> 1.940 ms at T = 8,
// +2 % at T = 16 - not instantiated.  (The BIG shape tried first - 512 threads, 48 slots, ONE work-group per CU, 4-slot ring - was
// bit-identical and 34 % / 6 % SLOWER on the conv / detector head, profiles/r5_big_tile_ab.txt: two lock-stepped waves of one work-group per
// SIMD leave the pipe idle at every barrier.)
template <bool CONV, int WN, bool FAT = false>
__global__ __launch_bounds__(FAT ? 256 : 512, FAT ? 2 : 4) void k_gemm_lif_sparse(const SparseConvArgs args) {
    // (FAT: linear layers on 2 x 2 waves; the conv on 4 x 1 for T <= 9, on 2 x 2 beyond)
    constexpr int NWAVES = FAT ? 4 : 8;
    constexpr int MTS = FAT ? (WN == 1 ? SP_MT_FAT : CONV ? SP_MT2_FAT_CONV : SP_MT2_FAT) : WN == 1 ? SP_MT : SP_MT2, NT = 4 / WN;      // slots per (row-)wave, 16-column N-tiles per wave
    constexpr int ROWS = SP_ROWS;                                   // physical tile rows of a ring slot (row-waves x MTS x 16 <= 512 in every shape)
    static_assert((NWAVES / WN) * MTS * 16 <= SP_ROWS, "ring slot rows");
    constexpr int A_BYTES = SP_A_ARR * ROWS * 4, SLOT = A_BYTES + SP_B_BYTES;
    constexpr int NPASS = FAT ? 2 : 1;                              // A-staging passes of a wave (64 rows each) per step
#ifndef SP_FAT_BDEPTH
#define SP_FAT_BDEPTH 2
#endif
    constexpr int BDEPTH = FAT ? SP_FAT_BDEPTH : 1;                 // groups the weight-fragment reads run ahead of their matrix instructions
#ifdef SNN_EXP_TIMELINE     // diagnostic build: wall-clock stamps (s_memrealtime, 100 MHz) of the work-group's phases.  Each stamp is stored at once (thread 0): the
    // 512-thread shapes sit at their 128-register limit, and stamps kept in registers until the end made the round-5 builds spill (304 bytes per lane: a
    // K loop 35 % slower than the product's)
#define SP_TL_STAMP(i) do { if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                                                   args.tl[(size_t)blockIdx.x * 8 + (i)] = t_; } } while (0)
    SP_TL_STAMP(0);
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned char* const lut = smem;
    unsigned char* const ring = smem + G3_LUT_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;                       // row-wave, column-wave
    // block order: XCD x = blockIdx % 8 takes xcd_cpx column blocks on a contiguous range of row tiles (k_gemm_bf16x3: xcd_contig)
    // (xcd_contig == 0 - the linear layers: plain order, column block fastest: an XCD only ever sees two weight panels, and all
    // work-groups of a panel walk K together)
    int nb = blockIdx.x % args.n_blocks, mb = blockIdx.x / args.n_blocks;
    if (args.xcd_contig) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, cpx = args.xcd_cpx, groups = args.n_blocks / cpx;
        nb = (x % groups) * cpx + j % cpx;
        mb = (x / groups) * args.xcd_contig + j / cpx;
        if (j / cpx >= args.xcd_contig || mb >= args.n_tiles) return;
    }
    nb = __builtin_amdgcn_readfirstlane(nb);
    mb = __builtin_amdgcn_readfirstlane(mb);
    const int pb = args.pb, M = args.M, Kc = args.Kc, Np = args.Np;
    const int m0 = mb * pb;
    if (tid < 256) {
        uint4 q;
        q.x = bf16_pair(tid, 0); q.y = bf16_pair(tid, 1); q.z = bf16_pair(tid, 2); q.w = bf16_pair(tid, 3);
        *reinterpret_cast<uint4*>(lut + tid * 16) = q;
    }
    const int nd_w = __builtin_amdgcn_readfirstlane((int)args.w_nd[wm]), ns_w = __builtin_amdgcn_readfirstlane((int)args.w_ns[wm]);
    // ---- A staging: lane L of the wave stages row L & 15 of one M-tile slot of its row-wave.  WN = 1: slot L >> 4 (64 rows per wave);
    // WN = 2: the two column-waves of a row-wave take three slots each (lanes 0 .. 47).  FAT: two passes - WN = 1: the wave's eight slots
    // (64 + 64 lanes); WN = 2: half the row-wave's slots per column-wave (six: 64 + 32 lanes; eight - the conv: 64 + 64)
    const void* a_base[SP_A_ARR];
#pragma unroll
    for (int j = 0; j < SP_A_ARR; ++j) a_base[j] = sgpr_ptr(reinterpret_cast<const char*>(args.enc) + (size_t)j * args.Pe * 4);
    uint32_t voff[NPASS], inc[NPASS];                       // byte offset from args.enc of the lane's first dword; array j of the step is
                                                            // j word planes further (dense lanes use two, the rest lands in unused LDS)
    bool a_lane[NPASS];
    uint32_t tap_fix[NPASS], row_fix[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int xs = FAT ? (WN == 1 ? 4 * ps + (lane >> 4) : min((MTS / 2) * wn + 4 * ps + (lane >> 4), MTS - 1)) : WN == 1 ? (lane >> 4) : min(3 * wn + (lane >> 4), MTS - 1);
        a_lane[ps] = FAT ? (WN == 1 || lane < (ps == 0 ? 64 : (MTS / 2 - 4) * 16)) : (WN == 1 || lane < 48);
        const int xplane = args.mt_plane[wm][xs];
        const bool xused = xplane >= 0, xdense = xused && xplane < args.nd;
        const int lp = min(args.mt_j[wm][xs] * 16 + (lane & 15), pb - 1);
        const int p = min(m0 + (xused ? lp : 0), M - 1);
        uint32_t row0 = (uint32_t)p;                        // linear layer: the RoI
        int W = 0;
        if (CONV) {
            int l = 0;
            while (l + 1 < args.n_levels && p >= args.lv[l + 1].pos_base) ++l;
            const int H = args.lv[l].H;
            W = args.lv[l].W;
            const int local = p - args.lv[l].pos_base;
            const int n = local / (H * W), rem = local % (H * W);
            const int y = rem / W, x = rem % W;
            row0 = (uint32_t)args.lv[l].tile_begin + (uint32_t)((n * (H + 2) + y) * (W + 2) + x);      // tap (-1, -1)
        }
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
    // physical row (wm MTS + slot) 16 + r; a pass covers four slots
    const uint32_t a_dst = smem_base + G3_LUT_BYTES + (FAT ? (wm * MTS + (WN == 1 ? 0 : (MTS / 2) * wn)) * 64 : WN == 1 ? wave * 256 : (wm * MTS + 3 * wn) * 64);
    const int cw2_s = __builtin_amdgcn_readfirstlane(args.Cw / 2);
    int f_c = 0, f_tap = 0;
    auto stage_a = [&](const uint32_t slot_off) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const uint32_t d = __builtin_amdgcn_readfirstlane(a_dst + slot_off + ps * 256);
            if (a_lane[ps]) {
                asm volatile("s_mov_b32 m0, %5\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1\n\t"
                             "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2\n\t"
                             "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dword %0, %3\n\t"
                             "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dword %0, %4"
                             :: "v"(voff[ps]), "s"(a_base[0]), "s"(a_base[1]), "s"(a_base[2]), "s"(a_base[3]), "s"(d), "s"((uint32_t)(ROWS * 4))
                             : "memory", "scc");
            }
            voff[ps] += inc[ps];
        }
        if (!CONV) return;
        f_c = __builtin_amdgcn_readfirstlane(f_c + 1);
        if (f_c == cw2_s) {
            f_c = 0;
            f_tap = __builtin_amdgcn_readfirstlane(f_tap + 1);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) voff[ps] += tap_fix[ps] + (f_tap == 3 ? row_fix[ps] : 0u);
            if (f_tap == 3) f_tap = 0;
        }
    };

    // ---- B staging: 24 pieces of 1 KB per step (2 chunks x 3 planes x 4 blocks of 16 columns); wave w copies pieces w, w + NWAVES, ..
    const int brow = (wave & 3) * 16 + (lane >> 2);
    const uint32_t b_off = (uint32_t)((nb * 64 + brow) * 64 + (((lane & 3) ^ G3_SWZ(brow)) << 4));
    const unsigned long long b_chunk = (unsigned long long)Np * 64, b_plane = args.plane_elems * 2;
    unsigned long long s_ptr = (unsigned long long)args.wpk;                  // chunk 2 * step, plane 0
    const uint32_t b_dst = smem_base + G3_LUT_BYTES + A_BYTES + (wave & 3) * 1024;
    auto stage_b = [&](const uint32_t slot_off) __attribute__((always_inline)) {
        const uint32_t d = __builtin_amdgcn_readfirstlane(b_dst + slot_off);
#pragma unroll
        for (int i = 0; i < 24 / NWAVES; ++i) {
            const int piece = wave + NWAVES * i;                               // wave-uniform; piece & 3 = wave & 3 = its block of 16 columns
            const int c2 = piece / 12, pl = (piece % 12) / 4;
            glds16(sgpr_ptr(reinterpret_cast<const void*>(s_ptr + c2 * b_chunk + pl * b_plane)), b_off, d + c2 * (3 * 64 * G3_ROWB) + pl * (64 * G3_ROWB));
        }
        s_ptr += 2 * b_chunk;
    };

    const unsigned char* const a_rd = ring + (wm * MTS * 16 + lr) * 4;                      // + slot offset, M-tile slot * 64
    const unsigned char* const b_rd = ring + A_BYTES + (wn * NT * 16 + lr) * G3_ROWB + ((lg ^ G3_SWZ(lr)) << 4);
    f32x4 acc[MTS][NT];
#pragma unroll
    for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int n_steps = Kc / 2;
    stage_a(0); stage_b(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

#ifdef SNN_EXP_TIMELINE
    SP_TL_STAMP(1);
#endif
#if defined(SNN_EXP_SP_NO_AREAD) || defined(SNN_EXP_SP_NO_BREAD)
    const bfv8 exp_a = *reinterpret_cast<const bfv8*>(lut + ((lane * 5) & 255) * 16);
    const bfv8 exp_b0 = *reinterpret_cast<const bfv8*>(b_rd), exp_b1 = *reinterpret_cast<const bfv8*>(b_rd + 3 * 64 * G3_ROWB);
#endif
    auto step_loop = [&](auto nd_c, auto ns_c) __attribute__((always_inline)) {
        constexpr int ND = decltype(nd_c)::value, NS = decltype(ns_c)::value;
        // A AHEAD (FAT conv on 4 x 1 waves: a wave reads only the rows it staged itself): the next step's A fragments - occupancy bytes, table
        // fragments, indices, secondary ballots - are requested at the END of a step, once this wave's copies have landed and BEFORE the step
        // barrier, into the registers the step's own fragments have just left: the byte -> table-fragment chain runs while the wave would wait
        // at the barrier anyway, and the first matrix instruction behind it finds its operand
        constexpr bool PRE_A = SP_PRE_A && WN == 1 && FAT;     // (the 512-thread shapes have no register to spare: the 8 x 1 conv spills with it)
        bfv8 p_ad[ND > 0 ? ND : 1][2], p_as[NS > 0 ? NS : 1];
        int p_ix[NS > 0 ? NS : 1];
        unsigned long long p_sec[4] = {0, 0, 0, 0};
        auto load_a_all = [&](const uint32_t off) __attribute__((always_inline)) {
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const uint32_t byte = *reinterpret_cast<const uint8_t*>(a_rd + off + c2 * (ROWS * 4) + d * 64 + lg);
                    p_ad[d][c2] = *reinterpret_cast<const bfv8*>(lut + (byte << 4));
                }
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const unsigned char* r = a_rd + off + (ND + q) * 64;
                const uint32_t occ = *reinterpret_cast<const uint8_t*>(r + lg);
                p_as[q] = *reinterpret_cast<const bfv8*>(lut + (occ << 4));
                p_ix[q] = (int)*reinterpret_cast<const uint16_t*>(r + (1 + (lg >> 1)) * (ROWS * 4) + 2 * (lg & 1));
            }
#pragma unroll
            for (int b4 = 0; b4 < (NS + 3) / 4; ++b4) {
                const uint32_t o2 = *reinterpret_cast<const uint32_t*>(a_rd + off + 3 * (ROWS * 4) + min(ND + 4 * b4 + lg, MTS - 1) * 64);
                p_sec[b4] = __ballot(o2 != 0u) & (NS - 4 * b4 >= 4 ? ~0ull : ((1ull << (16 * (NS - 4 * b4))) - 1ull));
            }
        };
        // weight fragments of group g = (N-tile g / 3, plane 2 - g % 3: small terms first): both 32-deep chunks of the step as ONE
        // 16-element operand (the structured-sparse instruction's B; its halves are the dense instruction's B of chunk c, c + 1),
        // buffered by g modulo the depth - the loads land in the halves of the buffer a later group reads, no register copies
#ifndef SP_FAT_BDEPTH_LIN
#define SP_FAT_BDEPTH_LIN 5
#endif
#ifndef SP_FAT_BDEPTH_C41
#define SP_FAT_BDEPTH_C41 BDEPTH
#endif
        // (round 6) the FAT linear layers (2 x 2 waves, six groups per step, 211 registers at two ahead) have the registers to request ALL of a step's weight fragments at its
        // top - five groups ahead, six buffers: detector head 0.700 -> 0.685 ms at T_det = 12, 1.533 -> 1.519 at 24 (profiles/r6_fc6_bdepth_ab.txt), same bits
        constexpr int BD = (FAT && ND + NS > 12) ? 1 : (FAT && !CONV) ? SP_FAT_BDEPTH_LIN : (FAT && CONV && WN == 1) ? SP_FAT_BDEPTH_C41 : BDEPTH;   // (the largest row-waves have no registers for a third buffer)
        constexpr bool TWO_PART = FAT && ND + NS > 14;
        // B HEAD START (-DSP_PRE_B=1, off: measured slower; FAT shapes whose step is one part): the first BD groups' fragments of step s + 1
        // requested right behind the barrier of step s - ahead of the copies' issue and the secondary-plane ballots of the step's top
        // (3 NT is a multiple of BD + 1: the ring carries over from step to step)
        constexpr bool PRE_B = SP_PRE_B && FAT && !TWO_PART;
        static_assert((3 * NT) % (BD + 1) == 0, "fragment ring carries over");
        bfv16 bbuf[BD + 1];
        auto load_b = [&](bfv16& dst, const uint32_t off, const int gn) __attribute__((always_inline)) {
#ifdef SNN_EXP_SP_NO_BREAD                            // (timing experiment: the weight fragments stay what they were before the loop)
            const bfv8 lo = exp_b0, hi = exp_b1;
#else
            const bfv8 lo = *reinterpret_cast<const bfv8*>(b_rd + off + (2 - gn % 3) * (64 * G3_ROWB) + (gn / 3) * 16 * G3_ROWB);
            const bfv8 hi = *reinterpret_cast<const bfv8*>(b_rd + off + 3 * 64 * G3_ROWB + (2 - gn % 3) * (64 * G3_ROWB) + (gn / 3) * 16 * G3_ROWB);
#endif
            dst = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        };
        if constexpr (PRE_A) load_a_all(0);
        if constexpr (PRE_B) {
#pragma unroll
            for (int g0 = 0; g0 < BD; ++g0) load_b(bbuf[g0], 0u, g0);
        }
        for (int s = 0; s < n_steps; ++s) {
            const uint32_t o_cur = (uint32_t)((s & 1) * SLOT), o_nxt = (uint32_t)(((s + 1) & 1) * SLOT);
            if (s + 1 < n_steps) {
#ifndef SNN_EXP_SP_NO_A                             // (timing experiments: what do the copies cost - wrong results)
                stage_a(o_nxt);
#endif
#ifndef SNN_EXP_SP_NO_B
                stage_b(o_nxt);
#endif
            }
            // which of the sparse M-tiles hold a third spike of a nibble in this step?  Lane L looks at the secondary occupancy dword of
            // row L & 15 of sparse slot L >> 4 (more than four sparse slots: a second / third / fourth look)
            unsigned long long sec[4] = {0, 0, 0, 0};             // bits 16 q .. 16 q + 15 of word b = sparse slot 4 b + q
#ifndef SNN_EXP_SP_NO_AREAD
            if constexpr (PRE_A) {
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) sec[b4] = p_sec[b4];
            }
            // (round 6, FAT conv on 2 x 2 waves: the secondary occupancy words are REQUESTED here and looked at behind the step's products, where the answer is
            // first needed - the ballot right behind the read made every step open with a full LDS round trip: conv + LIF 3.789 -> 3.753 ms at T = 16; the
            // linear layers measured 0.4 % slower with it and keep the early look, profiles/r6_sec_late_ab.txt)
            constexpr bool SEC_LATE = SP_SEC_LATE && FAT && CONV && WN == 2;
            uint32_t o2_late[(NS + 3) / 4 > 0 ? (NS + 3) / 4 : 1];
            if constexpr (!PRE_A && SEC_LATE) {
#pragma unroll
                for (int b4 = 0; b4 < (NS + 3) / 4; ++b4)
                    o2_late[b4] = *reinterpret_cast<const uint32_t*>(a_rd + o_cur + 3 * (ROWS * 4) + min(ND + 4 * b4 + lg, MTS - 1) * 64);
            } else if constexpr (!PRE_A) {
#pragma unroll
                for (int b4 = 0; b4 < (NS + 3) / 4; ++b4) {
                    const uint32_t o2 = *reinterpret_cast<const uint32_t*>(a_rd + o_cur + 3 * (ROWS * 4) + min(ND + 4 * b4 + lg, MTS - 1) * 64);
                    sec[b4] = __ballot(o2 != 0u) & (NS - 4 * b4 >= 4 ? ~0ull : ((1ull << (16 * (NS - 4 * b4))) - 1ull));
                }
            }
#endif
            // The step's products, in one part or - the largest row-wave, 15 M-tiles: no registers for all A fragments at once - in two (each
            // reads the weight fragments; every accumulator still sees its instructions in the same order).  part = the dense M-tiles (if
            // DENSE) and the sparse slots Q0 .. Q1 - 1
            auto do_part = [&](auto q0_c, auto q1_c, auto dense_c) __attribute__((always_inline)) {
                constexpr int Q0 = decltype(q0_c)::value, Q1 = decltype(q1_c)::value, NQ = Q1 - Q0;
                constexpr bool DENSE = decltype(dense_c)::value && ND > 0;
                // A fragments of this part
                bfv8 ad[ND > 0 ? ND : 1][2], as[NQ > 0 ? NQ : 1];
                int ix[NQ > 0 ? NQ : 1];
#ifdef SNN_EXP_SP_NO_AREAD                            // (timing experiment: no LDS reads on the A side - wrong results)
#pragma unroll
                for (int d = 0; d < ND; ++d) { ad[d][0] = exp_a; ad[d][1] = exp_a; }
#pragma unroll
                for (int q = 0; q < NQ; ++q) { as[q] = exp_a; ix[q] = 0x4444; }
#else
                if constexpr (PRE_A) {                          // (requested at the end of the previous step)
#pragma unroll
                    for (int d = 0; d < ND; ++d) { ad[d][0] = p_ad[d][0]; ad[d][1] = p_ad[d][1]; }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) { as[q] = p_as[Q0 + q]; ix[q] = p_ix[Q0 + q]; }
                } else {
                    if (DENSE) {
#pragma unroll
                        for (int d = 0; d < ND; ++d)
#pragma unroll
                            for (int c2 = 0; c2 < 2; ++c2) {
                                const uint32_t byte = *reinterpret_cast<const uint8_t*>(a_rd + o_cur + c2 * (ROWS * 4) + d * 64 + lg);
                                ad[d][c2] = *reinterpret_cast<const bfv8*>(lut + (byte << 4));
                            }
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const unsigned char* r = a_rd + o_cur + (ND + Q0 + q) * 64;
                        const uint32_t occ = *reinterpret_cast<const uint8_t*>(r + lg);
                        as[q] = *reinterpret_cast<const bfv8*>(lut + (occ << 4));
                        ix[q] = (int)*reinterpret_cast<const uint16_t*>(r + (1 + (lg >> 1)) * (ROWS * 4) + 2 * (lg & 1));
                    }
                }
#endif
                // weight fragments: the ring of the step loop (BD groups ahead of their matrix instructions)
                if constexpr (!PRE_B) {
#pragma unroll
                    for (int g0 = 0; g0 < BD; ++g0) load_b(bbuf[g0], o_cur, g0);
                }
#pragma unroll
                for (int g = 0; g < 3 * NT; ++g) {
#ifndef SNN_EXP_SP_NO_BREAD
                    if (g + BD < 3 * NT) load_b(bbuf[(g + BD) % (BD + 1)], o_cur, g + BD);
#endif
                    const bfv16 bb = bbuf[SNN_EXP_BSEL(g) % (BD + 1)];
#ifndef SNN_EXP_SP_NO_MFMA                            // (timing experiment: everything but the matrix instructions)
                    if (DENSE) {
                        const bfv8 b0 = __builtin_shufflevector(bb, bb, 0, 1, 2, 3, 4, 5, 6, 7), b1 = __builtin_shufflevector(bb, bb, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
                        for (int d = 0; d < ND; ++d) {
                            acc[d][g / 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad[d][0], b0, acc[d][g / 3], 0, 0, 0);
                            acc[d][g / 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad[d][1], b1, acc[d][g / 3], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        acc[ND + Q0 + q][g / 3] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(as[q], bb, acc[ND + Q0 + q][g / 3], ix[q], 0, 0);
#else
                    asm volatile("" :: "v"(bb));
                    if (DENSE) {
#pragma unroll
                        for (int d = 0; d < ND; ++d) asm volatile("" :: "v"(ad[d][0]), "v"(ad[d][1]));
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) asm volatile("" :: "v"(as[q]), "v"(ix[q]));
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if constexpr (TWO_PART) {
                constexpr int QS = (NS - 2 * ND + 1) / 2;          // (two parts of about equal matrix work: a dense M-tile counts twice)
                do_part(std::integral_constant<int, 0>{}, std::integral_constant<int, QS>{}, std::true_type{});
                do_part(std::integral_constant<int, QS>{}, std::integral_constant<int, NS>{}, std::false_type{});
            } else {
                do_part(std::integral_constant<int, 0>{}, std::integral_constant<int, NS>{}, std::true_type{});
            }
#if !defined(SNN_EXP_SP_NO_AREAD)
            if constexpr (!PRE_A && SEC_LATE) {
#pragma unroll
                for (int b4 = 0; b4 < (NS + 3) / 4; ++b4)
                    sec[b4] = __ballot(o2_late[b4] != 0u) & (NS - 4 * b4 >= 4 ? ~0ull : ((1ull << (16 * (NS - 4 * b4))) - 1ull));
            }
#endif
#ifdef SNN_EXP_SP_NO_SEC                              // (timing experiment - wrong results: what does the secondary plane's second pass cost?)
            if (false) {
#else
            if (NS > 0 && (sec[0] | sec[1] | sec[2] | sec[3]) != 0ull) {   // (rare) the secondary plane of the M-tiles that have one in this step
#endif
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    if (((sec[q >> 2] >> (16 * (q & 3))) & 0xffffull) == 0ull) continue;
                    const unsigned char* r = a_rd + o_cur + (ND + q) * 64 + 3 * (ROWS * 4);
                    const uint32_t occ = *reinterpret_cast<const uint8_t*>(r + lg);
                    const bfv8 a2 = *reinterpret_cast<const bfv8*>(lut + (occ << 4));
                    const int i2 = 0xeeee;                   // every nibble: positions (2, 3)
#pragma unroll
                    for (int g = 0; g < 3 * NT; ++g) {
                        const bfv8 c0 = *reinterpret_cast<const bfv8*>(b_rd + o_cur + (2 - g % 3) * (64 * G3_ROWB) + (g / 3) * 16 * G3_ROWB);
                        const bfv8 c1 = *reinterpret_cast<const bfv8*>(b_rd + o_cur + 3 * 64 * G3_ROWB + (2 - g % 3) * (64 * G3_ROWB) + (g / 3) * 16 * G3_ROWB);
                        bfv16 bb;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { bb[i] = c0[i]; bb[8 + i] = c1[i]; }
                        acc[ND + q][g / 3] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a2, bb, acc[ND + q][g / 3], i2, 0, 0);
                    }
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0): the next step's copies have landed
#ifndef SNN_EXP_SP_NO_AREAD
            if constexpr (PRE_A) {
                if (s + 1 < n_steps) load_a_all(o_nxt);       // (this wave's own rows; the table-fragment reads stay in flight across the barrier)
            }
#endif
#ifndef SNN_EXP_SP_NO_BAR                             // (timing experiment: waves run ahead of each other's copies - wrong results)
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            if constexpr (PRE_B) {
                if (s + 1 < n_steps) {
#pragma unroll
                    for (int g0 = 0; g0 < BD; ++g0) load_b(bbuf[g0], o_nxt, g0);
                }
            }
        }
    };
// (dense, sparse) M-tile counts of a row-wave the FAT shapes have loop instances for (host: sparse_plan_wn checks against the same lists)
// FAT conv: every wave (2 dense, Tc - 2 sparse), T = 7 .. 9 (T = 5 / 6 measured 9.5 % / 1.3 % SLOWER than the 8-wave shape, whose tiles
// hold 128 / 80 positions there against the FAT conv's 64: profiles/r5_fat_conv_ab.txt)
#define SP_FAT1_INSTANCES {2, 5}, {2, 4}, {2, 6}
#define SP_FAT1_CASES SP_CASE(2, 5) SP_CASE(2, 4) SP_CASE(2, 6)
// FAT conv on 2 x 2 waves, T = 12 .. 16 (instances from T = 10: 10 / 11 measured slower than the 8-wave shape): every row-wave (2 dense, Tc - 2 sparse)
#define SP_FAT1B_INSTANCES {2, 7}, {2, 8}, {2, 9}, {2, 10}, {2, 11}, {2, 12}, {2, 13}
#define SP_FAT1B_CASES SP_CASE(2, 7) SP_CASE(2, 8) SP_CASE(2, 9) SP_CASE(2, 10) SP_CASE(2, 11) SP_CASE(2, 12) SP_CASE(2, 13)
#define SP_FAT2_INSTANCES {2, 8}, {2, 7}, {2, 9}, {2, 10}, {2, 6}, {2, 5}, {2, 4}, {2, 3}, {2, 2}, {1, 10}, {1, 11}, {1, 9}, {1, 8}, {1, 7}, {1, 6}, {1, 5}
#define SP_FAT2_CASES SP_CASE(2, 8) SP_CASE(2, 7) SP_CASE(2, 9) SP_CASE(2, 10) SP_CASE(2, 6) SP_CASE(2, 5) SP_CASE(2, 4) SP_CASE(2, 3) SP_CASE(2, 2) \
                      SP_CASE(1, 10) SP_CASE(1, 11) SP_CASE(1, 9) SP_CASE(1, 8) SP_CASE(1, 7) SP_CASE(1, 6) SP_CASE(1, 5)
#define SP_CASE(ND_, NS_) if (nd_w == ND_ && ns_w == NS_) step_loop(std::integral_constant<int, ND_>{}, std::integral_constant<int, NS_>{}); else
    if constexpr (FAT && CONV && WN == 1) {
        SP_FAT1_CASES { step_loop(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); }
    } else if constexpr (FAT && CONV) {
        SP_FAT1B_CASES { step_loop(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); }
    } else if constexpr (FAT) {
        SP_FAT2_CASES { step_loop(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); }
    } else if constexpr (WN == 1) {
        SP_CASE(1, 3) SP_CASE(1, 2) SP_CASE(2, 2) SP_CASE(0, 4) SP_CASE(0, 3) SP_CASE(1, 1) SP_CASE(2, 1) SP_CASE(0, 2) SP_CASE(0, 1)
        SP_CASE(2, 0) SP_CASE(1, 0)
        {   // a wave without M-tiles still stages and keeps the barriers
            step_loop(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        }
    } else {
        SP_CASE(1, 4) SP_CASE(1, 5) SP_CASE(2, 4) SP_CASE(2, 3) SP_CASE(1, 3) SP_CASE(0, 6) SP_CASE(0, 5) SP_CASE(2, 2) SP_CASE(0, 4)
        SP_CASE(1, 2) SP_CASE(0, 3)
        {
            step_loop(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        }
    }
#undef SP_CASE

#ifdef SNN_EXP_TIMELINE
    SP_TL_STAMP(2);
#endif
    if (FAT && (CONV || args.lif_regs)) {                          // (block-uniform; the FAT conv has no other epilogue: its launcher plans it only where an instance exists)
        // ---- epilogue of the FAT shapes: the LIF in registers (sp_lif_regs), each (row-)wave for its own 16 positions / RoIs; no LDS, no barrier
        if constexpr (FAT) {
            const int T = args.T;
            uint32_t mine[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
            const bool counting = args.cnt_row != nullptr;
#define SP_R(n) case n: if (counting) sp_lif_regs<n, CONV ? 1 : 2, NT, MTS, true>(acc, args.p, lane, mine, cnt); \
                        else sp_lif_regs<n, CONV ? 1 : 2, NT, MTS, false>(acc, args.p, lane, mine, cnt); break;
            if constexpr (CONV && WN == 1) { switch (T) { SP_R(7) SP_R(8) SP_R(9) default: break; } }
            else if constexpr (CONV) { switch (T) { SP_R(10) SP_R(11) SP_R(12) SP_R(13) SP_R(14) SP_R(15) SP_R(16) default: break; } }
            else { switch (T) { SP_R(6) SP_R(7) SP_R(8) SP_R(9) SP_R(10) SP_R(11) SP_R(12) SP_R(13) SP_R(14) default: break; } }
#undef SP_R
            // this lane's (step, word): 64 columns per wave = two words per position (lane lr <-> step 1 + (lr >> 1), word lr & 1), 32 columns per
            // column-wave = one (lane lr <-> step 1 + lr)
            const int t_mine = 1 + (NT == 4 ? ((lane & 15) >> 1) : (lane & 15)), word = NT == 4 ? nb * 2 + (lane & 1) : nb * 2 + wn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lp = 16 * wm + 4 * lg + r, pos = m0 + lp;
                if (pos >= M || lp >= pb) continue;
                uint32_t* dst;
                if (!CONV) dst = args.spk + (size_t)word * M + pos;                     // word-major spike planes [T][word][RoI] (fc6 -> fc7)
                else if (args.out_split) dst = args.spk + ((size_t)(word >> 2) * M + pos) * 4 + (word & 3);
                else dst = args.spk + (size_t)pos * (Np >> 5) + word;
                if (t_mine < T) dst[(size_t)t_mine * args.spk_stride] = mine[r];
                if ((lane & 15) < (NT == 4 ? 2 : 1)) dst[0] = 0u;                      // step 0: no spike
                if (counting && (lane & 15) == 0 && cnt[r]) atomicAdd(args.cnt_row + pos, cnt[r]);      // (this wave's 32 / 64 columns of the position / RoI)
            }
#ifdef SNN_EXP_TIMELINE
            SP_TL_STAMP(7); SP_TL_STAMP(3);
            if (tid == 0) {
                unsigned long long tl_exit;
                uint32_t hw, xcc;
                asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                             : "=s"(tl_exit), "=s"(hw), "=s"(xcc) :: "memory");
                unsigned long long* o = args.tl + (size_t)blockIdx.x * 8;
                o[4] = tl_exit; o[5] = hw; o[6] = xcc;
            }
#endif
        }
        return;
    }
    // ---- epilogue: currents -> LDS tile image (two passes of 32 columns), LIF over the T steps, spike words out
    const int T = args.T, Tc = args.Tc;
    float* const tile = reinterpret_cast<float*>(smem);
    const int rows_l = Tc * pb;
    __syncthreads();                                           // ring reads done
    const int group_stride = pb * SP_PITCH;
    const bool counting = args.cnt_row != nullptr;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (WN == 1 || wn == h) {                              // (4 x 2 grid: the column-wave that holds this pass's 32 columns)
#pragma unroll
            for (int mt = 0; mt < MTS; ++mt) {
                const int plane = args.mt_plane[wm][mt];       // wave-uniform
                if (plane < 0) continue;
                const int lp0 = args.mt_j[wm][mt] * 16 + lg * 4;
#pragma unroll
                for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float val;
                        if constexpr (WN == 1) val = h == 0 ? acc[mt][nq][r] : acc[mt][2 + nq][r];
                        else val = acc[mt][nq][r];
                        if (lp0 + r < pb) tile[(plane * pb + lp0 + r) * SP_PITCH + nq * 16 + lr] = val;
                    }
            }
        }
        __syncthreads();
#ifdef SNN_EXP_TIMELINE
        if (h == 0) SP_TL_STAMP(7);
#endif
        const int word0 = (nb * 64 + h * 32) >> 5;
        const int par = lane >> 5, col = lane & 31;
        // position pairs per wave and iteration (see sp_lif_fixed): two up to T = 10, one beyond (registers: T - 1 period sums per neuron);
        // the FAT shape's waves have the registers for twice that - and half the waves to hide the recurrence's latencies with
        constexpr int NP_SHORT = FAT ? 4 : 2, NP_LONG = FAT ? 2 : 1;
        auto lif_pass = [&](auto np_c, auto count_c) __attribute__((always_inline)) {
        constexpr int NP = decltype(np_c)::value;
        constexpr bool COUNT = decltype(count_c)::value;
        for (int pp0 = wave; 2 * pp0 < pb; pp0 += NWAVES * NP) {
            if (m0 + 2 * pp0 >= M) break;
            uint32_t my0[NP], my1[NP], cnt_lo[NP], cnt_hi[NP];
            const float* src[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int pi = 2 * (pp0 + NWAVES * u) + par;
                const bool live = pi < pb && m0 + pi < M;
                my0[u] = 0; my1[u] = 0; cnt_lo[u] = 0; cnt_hi[u] = 0;
                src[u] = tile + (live ? pi : 2 * pp0) * SP_PITCH + col;    // (dead lanes / pairs recompute a live row: never stored)
            }
#define SP_T(n) case n: sp_lif_fixed<n, CONV ? 1 : 2, NP, COUNT>(src, group_stride, args.p, my0, my1, cnt_lo, cnt_hi); break;
            if (NP == 1 && !CONV && args.epi_general) {               // (block-uniform; linear layers only: the conv's launcher keeps to the fixed grid)
                if constexpr (NP == 1 && !CONV) sp_lif_general<COUNT>(src[0], group_stride, args.p, T, Tc, args.div, lane, my0[0], my1[0], cnt_lo[0], cnt_hi[0]);
            } else if constexpr (COUNT) {
                switch (T) { SP_T(5) SP_T(6) SP_T(7) SP_T(8) SP_T(9) SP_T(10) SP_T(11) SP_T(12) SP_T(13) SP_T(14) SP_T(15) SP_T(16) default: break; }
            } else if constexpr (NP == NP_SHORT) {
                switch (T) { SP_T(5) SP_T(6) SP_T(7) SP_T(8) SP_T(9) SP_T(10) default: break; }
            } else {
                switch (T) { SP_T(11) SP_T(12) SP_T(13) SP_T(14) SP_T(15) SP_T(16) default: break; }
            }
#undef SP_T
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int pp = pp0 + NWAVES * u;
                if (2 * pp >= pb || m0 + 2 * pp >= M) continue;
                const bool odd_ok = 2 * pp + 1 < pb && m0 + 2 * pp + 1 < M;
                if (COUNT && lane == 0) {                                  // (dead odd rows recompute the even one: not counted)
                    if (cnt_lo[u]) atomicAdd(args.cnt_row + m0 + 2 * pp, cnt_lo[u]);
                    if (odd_ok && cnt_hi[u]) atomicAdd(args.cnt_row + m0 + 2 * pp + 1, cnt_hi[u]);
                }
                if (lane < T) {
                    if (!CONV) {                                           // linear layer: word-major spike planes [T][word][RoI] (fc6 -> fc7)
                        uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + (size_t)word0 * M + (m0 + 2 * pp);
                        dst[0] = my0[u];
                        if (odd_ok) dst[1] = my1[u];
                    } else if (args.out_split) {
                        uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + ((size_t)(word0 >> 2) * M + m0 + 2 * pp) * 4 + (word0 & 3);
                        dst[0] = my0[u];
                        if (odd_ok) dst[4] = my1[u];
                    } else {
                        uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + (size_t)(m0 + 2 * pp) * (Np >> 5) + word0;
                        dst[0] = my0[u];
                        if (odd_ok) dst[Np >> 5] = my1[u];
                    }
                }
            }
        }
        };
        if (args.epi_general) {                                    // (linear layers outside the straight-line grid: one run-time recurrence per lane)
            if (counting) lif_pass(std::integral_constant<int, 1>{}, std::true_type{});
            else lif_pass(std::integral_constant<int, 1>{}, std::false_type{});
        } else if (counting) {                                     // (the counters take the registers of one recurrence)
            lif_pass(std::integral_constant<int, NP_LONG>{}, std::true_type{});
        } else {
            if (T <= 10) lif_pass(std::integral_constant<int, NP_SHORT>{}, std::false_type{});
            else lif_pass(std::integral_constant<int, NP_LONG>{}, std::false_type{});
        }
#ifdef SNN_EXP_TIMELINE
        if (h == 0) SP_TL_STAMP(3);
#endif
    }
#ifdef SNN_EXP_TIMELINE
    if (tid == 0) {
        unsigned long long tl_exit;
        uint32_t hw, xcc;
        asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(tl_exit), "=s"(hw), "=s"(xcc) :: "memory");
        unsigned long long* o = args.tl + (size_t)blockIdx.x * 8;       // (behind the compressed planes: tools/sparse_timeline.py allocates more)
        o[4] = tl_exit; o[5] = hw; o[6] = xcc;                          // (o[7] != 0 marks the record)
    }
#endif
}
