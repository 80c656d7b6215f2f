// fp32-matrix-core family (precision = "f32"): fused 3x3 spike convolution + LIF (K2) and the time-batched spike
// GEMM (K3).  Included by snn_kernels.hip.
#pragma once

// ------------------------------------------------------------------------------------------------
// K2: fused 3x3 spike convolution (implicit GEMM on the fp32 matrix cores) + LIF over the T loop.
//
// Work-group = 512 threads = 8 waves; tile = 64 positions (an 8x8 patch of one image of one level)
// x 256 output channels; wave w owns all 64 positions x channels [32w, 32w+32)  (2 x 1 MFMA tiles).
// Per lane: 32 accumulators + 32 membrane voltages + 32 synaptic currents stay in registers for the
// whole T loop; nothing but spike bits is written to HBM.
//   A operand: the encoder spikes of the (8+2)x(8+2) halo of the patch are expanded ONCE per time
//              step from bit-planes to an fp32 LDS image [halo position][channel] (104 KB at C=256);
//              every (tap, channel-chunk) operand is then a plain ds_read_b128 - the MFMA loop holds
//              no VALU instruction (on gfx950 each one costs ~6.5 cycles of fp32 matrix-pipe time).
//   B operand: every wave streams its own 4-KiB weight fragment per 32-deep chunk straight from the
//              packed global image (L2-resident: 2.4 MB) into registers, one chunk ahead of the
//              MFMAs (ping-pong register sets) - no LDS staging, no per-chunk barrier.
// Spatial tiles are independent for the whole T loop (the only coupling is the conv halo on the
// *encoder* spikes), so there is no inter-work-group synchronisation.
// ------------------------------------------------------------------------------------------------
struct ConvLevelDev {
    int pos_base;        // first row (position) of this level in the plane buffers
    int N, H, W;
    int tiles_x, tiles_per_img;
    int tile_begin;      // first blockIdx.x of this level
    int pad;
};
struct ConvArgs {
    const uint32_t* enc;
    uint32_t* spk;
    const float* wpk;
    unsigned long long* counts;
    float* dbg_cur;                              // nullable: input currents [T][P][Nw*32] (parity tests)
    unsigned long long enc_stride, spk_stride;   // words per time plane
    int Cw, Nw, T, n_levels, max_n;
    int Tc;                                      // input currents are formed for steps 0 .. Tc-1; later steps integrate +0 (dead time
                                                 // steps: the current of step t first moves the membrane at t+1, Gemm3Args.Tc)
    NeuronP p;
    ConvLevelDev lv[SNN_MAX_LEVELS];
};

#define CONV_PH 8
#define CONV_PW 8
#define CONV_HALO ((CONV_PH + 2) * (CONV_PW + 2))
#define CONV_BNT 8                                // n-tiles (waves) per block
#define CONV_APAD 4                               // floats of padding per halo position (bank spread)
#define CONV_MAX_CW 12                            // 100 x (384+4) x 4 B = 155 KB of LDS
#define CONV_HW ((CONV_HALO * CONV_MAX_CW + 511) / 512)   // halo words per thread

template <bool DBG>
__global__ __launch_bounds__(512) void k_conv3x3_lif(const ConvArgs args) {
    constexpr int MT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ldsA = reinterpret_cast<float*>(smem);                  // [CONV_HALO][CST]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // ---- which tile ----
    int l = 0;
    const int b = blockIdx.x;
    while (l + 1 < args.n_levels && b >= args.lv[l + 1].tile_begin) ++l;
    const ConvLevelDev L = args.lv[l];
    const int local = b - L.tile_begin;
    const int n = local / L.tiles_per_img;
    const int rem = local % L.tiles_per_img;
    const int y0 = (rem / L.tiles_x) * CONV_PH, x0 = (rem % L.tiles_x) * CONV_PW;
    const int H = L.H, W = L.W;
    const size_t img_base = (size_t)L.pos_base + (size_t)n * H * W;
    const int Cw = args.Cw, Nw = args.Nw;
    const int CST = Cw * 32 + CONV_APAD;             // floats per halo position
    const int ntg = blockIdx.y * CONV_BNT + wave;    // this wave's 32-channel output tile
    const bool active = ntg < Nw;                    // wave-uniform

    // this lane's A rows: LDS offset (in 16-byte units: CST is a multiple of 4 floats) of position
    // (py, px) of the patch for the two M-tiles
    const f32x4* lds16 = reinterpret_cast<const f32x4*>(smem);
    uint32_t a_q[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        a_q[mt] = (uint32_t)(((mt * 4 + (li >> 3)) * (CONV_PW + 2) + (li & 7)) * (CST / 4) + 4 * lh);

    // which (mt, r) accumulator rows of this lane are inside the image -> spike mask
    uint32_t valid_bits = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lh);
            const int yy = y0 + mt * 4 + (row >> 3), xx = x0 + (row & 7);
            valid_bits |= (uint32_t)(yy < H && xx < W) << (mt * 16 + r);
        }

    f32x16 acc[MT];
    float v[MT][16], cur_i[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { v[mt][r] = args.p.v_leak; cur_i[mt][r] = 0.0f; }   // state fallback

    const int KC = 9 * Cw;
    // weight fragment of (chunk kc, tile ntg): 256 float4, lane reads [qq*64 + lane]
    // (wave-uniform base pointer + lane index: global_load with an SGPR base, no per-chunk VALU address math)
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(args.wpk) + (size_t)(active ? ntg : 0) * 256;
    const size_t wchunk = (size_t)Nw * 256;          // float4 per reduction chunk (all n-tiles)

    // ---- halo spike words of one time step: thread owns words idx = tid + 512*j ----
    uint32_t hw[CONV_HW];
    auto fetch_halo = [&](int t) {
        const uint32_t* enc_t = args.enc + (size_t)t * args.enc_stride;
#pragma unroll
        for (int j = 0; j < CONV_HW; ++j) {
            const int idx = tid + 512 * j;
            hw[j] = 0;
            if (idx < CONV_HALO * Cw) {
                const int cc = idx % Cw, hp = idx / Cw;
                const int y = y0 - 1 + hp / (CONV_PW + 2), x = x0 - 1 + hp % (CONV_PW + 2);
                if (y >= 0 && y < H && x >= 0 && x < W) hw[j] = enc_t[(img_base + (size_t)y * W + x) * Cw + cc];
            }
        }
    };
    auto expand_halo = [&]() {
#pragma unroll
        for (int j = 0; j < CONV_HW; ++j) {
            const int idx = tid + 512 * j;
            if (idx < CONV_HALO * Cw) expand_word(hw[j], ldsA + (idx / Cw) * CST + (idx % Cw) * 32);
        }
    };

    // ---- software pipeline state: chunk whose operands are fetched NEXT ----
    const f32x4* wnext = wsrc;       // weights: wraps around at KC (next step re-reads the same image)
    int kcB = 0;
    auto load_b = [&](f32x4 (&dst)[4]) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) dst[qq] = wnext[qq * 64 + lane];
        if (++kcB == KC) { kcB = 0; wnext = wsrc; } else wnext += wchunk;
    };
    int tapA = 0, ccA = 0;           // spike operands: (tap, channel word) of the chunk being fetched
    uint32_t a_cur[MT];              // lane's LDS index (16-B units) of the chunk in flight: one v_add per M-tile
    auto load_a_lo = [&](f32x4 (&dst)[MT][2]) {      // first half (qq 0,1) of the next chunk
        const uint32_t cq = (uint32_t)(((tapA / 3) * (CONV_PW + 2) + (tapA % 3)) * (CST / 4) + ccA * 8);   // scalar
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            a_cur[mt] = a_q[mt] + cq;
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) dst[mt][q2] = lds16[a_cur[mt] + q2];
        }
        if (++ccA == Cw) { ccA = 0; ++tapA; }
    };
    auto load_a_hi = [&](f32x4 (&dst)[MT][2]) {      // second half (qq 2,3) of the same chunk
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) dst[mt][q2] = lds16[a_cur[mt] + 2 + q2];
    };

    f32x4 b0[4], b1[4];
    f32x4 alo[MT][2], ahi[MT][2];
    fetch_halo(0);
    if (active) load_b(b0);
    unsigned long long n_spikes = 0;

    for (int t = 0; t < args.T; ++t) {
        const bool live_t = t < args.Tc;                   // block-uniform: a step whose input current is formed
        if (live_t) {
            __syncthreads();                               // everyone done reading the previous image
            expand_halo();
            if (t + 1 < args.Tc) fetch_halo(t + 1);        // latency hidden behind this step's MFMAs
            __syncthreads();                               // image of step t complete
        }
        if (active) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
          if (live_t) {
            tapA = 0; ccA = 0;
            load_a_lo(alo);
            // A operands run half a chunk ahead of the MFMAs, B operands one chunk ahead
            auto chunk = [&](int kc, const f32x4 (&bcur)[4], f32x4 (&bnext)[4]) {
                load_b(bnext);                 // chunk kc+1 (or chunk 0 of the next time step)
                load_a_hi(ahi);
                __builtin_amdgcn_sched_barrier(0);
                mma_half<MT, 0>(acc, alo, bcur);
                __builtin_amdgcn_sched_barrier(0);
                if (kc + 1 < KC) load_a_lo(alo);
                __builtin_amdgcn_sched_barrier(0);
                mma_half<MT, 1>(acc, ahi, bcur);
                __builtin_amdgcn_sched_barrier(0);
            };
            int kc = 0;
            for (; kc + 1 < KC; kc += 2) {     // two chunks per trip: ping-pong weight registers, no copies
                chunk(kc, b0, b1);
                chunk(kc + 1, b1, b0);
            }
            if (kc < KC) {                     // odd chunk count (C_in = 32 * odd)
                chunk(kc, b0, b1);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) b0[qq] = b1[qq];
            }
          }
            if (DBG) {                         // test-hook instantiation: dump the step's input currents
                // (one base pointer per lane and 32-bit element offsets formed per store from two values the compiler may not hoist out of the time
                // loop: 32 loop-invariant 64-bit addresses kept live made this instance spill 4 registers - 20 bytes of scratch per lane - at its
                // 256-register limit; tests/test_code_object.py wants no scratch anywhere in the library)
                float* const d = args.dbg_cur + (size_t)t * (args.spk_stride * 32) + (img_base + (size_t)y0 * W + x0) * (size_t)(Nw * 32) + ntg * 32 + li;
                uint32_t row_pitch = (uint32_t)W * (uint32_t)(Nw * 32), col_pitch = (uint32_t)(Nw * 32);
                asm volatile("" : "+v"(row_pitch), "+v"(col_pitch));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = acc_row(r, lh);
                        if ((valid_bits >> (mt * 16 + r)) & 1u) d[(uint32_t)(mt * 4 + (row >> 3)) * row_pitch + (uint32_t)(row & 7) * col_pitch] = acc[mt][r];
                    }
            }
            // ---- LIF epilogue in registers; spikes leave as ballots ----
            uint32_t myword = 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    bool z = lif_step(acc[mt][r], v[mt][r], cur_i[mt][r], args.p);
                    z = z && ((valid_bits >> (mt * 16 + r)) & 1u);
                    const unsigned long long m = __ballot(z);
                    n_spikes += __popcll(m);
                    const int L0 = mt * 32 + r * 2;
                    myword = (lane == L0) ? (uint32_t)m : myword;
                    myword = (lane == L0 + 1) ? (uint32_t)(m >> 32) : myword;
                }
            }
            {   // lane -> (mt, r, half): one 32-channel word of one position
                const int mt = lane >> 5, r = (lane >> 1) & 15, hh = lane & 1;
                const int row = acc_row(r, hh);
                const int yy = y0 + mt * 4 + (row >> 3), xx = x0 + (row & 7);
                if (yy < H && xx < W)
                    args.spk[(size_t)t * args.spk_stride + (img_base + (size_t)yy * W + xx) * Nw + ntg] = myword;
            }
        }
    }
    if (args.counts != nullptr && lane == 0 && n_spikes != 0)
        atomicAdd(&args.counts[l * args.max_n + n], n_spikes);
}

// ------------------------------------------------------------------------------------------------
// K3: time-batched spike GEMM  cur[M][ldo] = A_bits[M][K] x W[K][N]   (rows m = t*R + r)
// Work-group = 8 waves = 128 rows x 256 columns; wave w owns the 128 rows x columns of n-tile w
// (4 x 1 MFMA tiles).  Per 32-deep chunk the 512 threads expand the 128 spike words of the tile to
// an fp32 LDS image [row][32 k] (double-buffered, one barrier per chunk; 16 bit->float conversions
// per thread per 64 MFMAs per wave), A operands are ds_read_b128, B fragments stream from global
// one chunk ahead exactly as in K2.
// blockIdx -> (row tile, panel) puts all work-groups of one XCD (blockIdx % 8) on the same weight
// panel whenever the panel count divides 8, so a panel is fetched from HBM once per XCD and then
// served from that XCD's L2.
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    const uint32_t* A;
    const float* wpk;
    float* out;
    int M, Kw, Nw, ldo, n_blocks, pad;
};

#define GEMM_AST 36                                 // floats per LDS row: 32 + 4 (conflict-free b128 reads)

template <int MT>
__global__ __launch_bounds__(512) void k_spike_gemm(const GemmArgs args) {
    static_assert(MT == 4, "512 threads expand 128 rows x 4 bytes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ldsA = reinterpret_cast<float*>(smem);                  // 2 x [128][GEMM_AST]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nb = blockIdx.x % args.n_blocks;
    const int mb = blockIdx.x / args.n_blocks;
    const int m0 = mb * (MT * 32);
    const int Kw = args.Kw, Nw = args.Nw, M = args.M;
    const int ntg = nb * 8 + wave;
    const bool active = ntg < Nw;                     // wave-uniform

    // expansion role: thread -> (row, byte) of the tile's spike words
    const int xrow = tid >> 2, xbyte = tid & 3;
    const uint32_t* xsrc = args.A + (size_t)min(m0 + xrow, M - 1) * Kw;
    float* xdst = ldsA + xrow * GEMM_AST + xbyte * 8;
    auto expand = [&](uint32_t w, int buf) {
        const uint32_t byte = (w >> (8 * xbyte)) & 0xffu;
        f32x4 lo, hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) { lo[r] = (float)((byte >> r) & 1u); hi[r] = (float)((byte >> (4 + r)) & 1u); }
        float* d = xdst + buf * (128 * GEMM_AST);
        *reinterpret_cast<f32x4*>(d) = lo;
        *reinterpret_cast<f32x4*>(d + 4) = hi;
    };

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

    const f32x4* wsrc = reinterpret_cast<const f32x4*>(args.wpk) + (size_t)(active ? ntg : 0) * 256;   // uniform
    const size_t wchunk = (size_t)Nw * 256;
    const float* abase = ldsA + li * GEMM_AST + 16 * lh;

    f32x4 b0[4], b1[4];
    auto load_b = [&](f32x4 (&dst)[4], int kc) {
        const f32x4* wn = wsrc + (size_t)min(kc, Kw - 1) * wchunk;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) dst[qq] = wn[qq * 64 + lane];
    };
    auto chunk = [&](int kc, const f32x4 (&bcur)[4], f32x4 (&bnext)[4], uint32_t& wnext) {
        // stage chunk kc+1 (spike image + weights) while chunk kc is multiplied
        if (kc + 1 < Kw) expand(wnext, (kc + 1) & 1);
        wnext = xsrc[min(kc + 2, Kw - 1)];
        if (active) {
            load_b(bnext, kc + 1);
            f32x4 a[MT][4];
            const float* ab = abase + (kc & 1) * (128 * GEMM_AST);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) a[mt][qq] = *reinterpret_cast<const f32x4*>(ab + mt * 32 * GEMM_AST + 4 * qq);
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk<MT>(acc, a, bcur);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    uint32_t wnext = xsrc[0];
    expand(wnext, 0);
    wnext = xsrc[min(1, Kw - 1)];
    if (active) load_b(b0, 0);
    __syncthreads();
    int kc = 0;
    for (; kc + 1 < Kw; kc += 2) {
        chunk(kc, b0, b1, wnext);
        chunk(kc + 1, b1, b0, wnext);
    }
    if (kc < Kw) chunk(kc, b0, b1, wnext);
    if (!active) return;
    // ---- store currents (lanes 0-31 / 32-63 write two 128-B row segments per instruction) ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mt * 32 + acc_row(r, lh);
            if (m < M) args.out[(size_t)m * args.ldo + ntg * 32 + li] = acc[mt][r];
        }
}
