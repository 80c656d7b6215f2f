// Block-scaled fp6 digit planes for the spike GEMMs (included by snn_kernels.hip).
//
// v_mfma_scale_f32_16x16x128_f8f6f4 multiplies fp4 x fp6 at ~2.8x the sustained rate of the bf16 MFMA
// (tools/mx_probe.hip: 7.0 PF against 2.47 PF register-resident), its products of small integers are exact and
// it accumulates in fp32.  A = spikes {0,1} as fp4; B = the fp32 weights as MX_P planes of signed base-32 digits
// (fp6 e2m3 holds every integer in [-16,16] in units of 1/8), one E8M0 scale per (32 consecutive k, column):
//     w  ~=  sum_p  d_p * 2^(Eb - 3 - 5p),    Eb = exponent of the largest magnitude in the block
// MX_P = 6 planes reach 2^(Eb-28): a weight is represented EXACTLY when it is within 2^5 of the largest one of its
// block, otherwise it is rounded at 2^(Eb-28) (|error| <= 2^-29 of the block maximum - 32x below the fp32 rounding
// of one addition to a sum of that size).  Executed MFMA work = 6/4 of the bf16-equivalent instead of 3.
#pragma once

#define MX_P 6
#ifndef MX_PD
#define MX_PD 1
#endif
#ifndef MX_MW_DEFAULT
#define MX_MW_DEFAULT 4
#endif

// fp6 e2m3 code of an integer digit d in [-16, 16] (value d/8)
__device__ __forceinline__ uint32_t mx_fp6_code(int d) {
    const uint32_t s = d < 0 ? 32u : 0u;
    const uint32_t u = (uint32_t)(d < 0 ? -d : d);
    return s | (u < 16u ? u : 16u);          // u < 8: subnormal m = u; 8..15: e = 1, m = u - 8 (same bits); 16: e = 2, m = 0
}

// packed layout (uint32 words):  X [P][Kc][Np][4 kgroups][4]   first 16 B of a lane's 24-B fragment (32 digits x 6 bit)
//                                Y [P][Kc][Np][4 kgroups][2]   last 8 B
//                                S [Kc][Np]                    byte lg = biased exponent Eb of block (kc, n, lg)
// Kc = 128-deep chunks; conv: k = tap*Cp + ci with Cp = C_in rounded up to 128.
__host__ __device__ inline size_t mx_x_words(int Kc, int Np) { return (size_t)MX_P * Kc * Np * 16; }
__host__ __device__ inline size_t mx_y_words(int Kc, int Np) { return (size_t)MX_P * Kc * Np * 8; }
__host__ __device__ inline size_t mx_s_words(int Kc, int Np) { return (size_t)Kc * Np; }
// + 16 spare zero words
__host__ __device__ inline size_t mx_words(int Kc, int Np) { return mx_x_words(Kc, Np) + mx_y_words(Kc, Np) + mx_s_words(Kc, Np) + 16; }

__global__ void k_pack_mx(const float* __restrict__ src, uint32_t* __restrict__ dst, int mode, int K, int N, int Kc, int Np,
                          int Cin, int Cp) {
    const size_t blocks = (size_t)Kc * Np * 4;
    uint32_t* const X = dst;
    uint32_t* const Y = dst + mx_x_words(Kc, Np);
    uint8_t* const S = reinterpret_cast<uint8_t*>(dst + mx_x_words(Kc, Np) + mx_y_words(Kc, Np));
    if (blockIdx.x == 0 && threadIdx.x < 16) dst[mx_words(Kc, Np) - 16 + threadIdx.x] = 0u;
    for (size_t blk = (size_t)blockIdx.x * blockDim.x + threadIdx.x; blk < blocks; blk += (size_t)gridDim.x * blockDim.x) {
        const int lg = (int)(blk & 3);
        const int n = (int)((blk >> 2) % Np);
        const int kc = (int)((blk >> 2) / Np);
        uint32_t bits[32];
        uint32_t eb = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int k = kc * 128 + lg * 32 + j;
            float w = 0.0f;
            if (n < N) {
                if (mode == PACK_CONV3X3) {
                    const int tap = k / Cp, ci = k % Cp;
                    if (tap < 9 && ci < Cin) w = src[((size_t)n * Cin + ci) * 9 + tap];
                } else if (k < K) {
                    w = src[(size_t)n * K + k];
                }
            }
            bits[j] = __float_as_uint(w);
            const uint32_t e = (bits[j] >> 23) & 0xffu;
            eb = max(eb, max(e, 1u));                           // denormals count as exponent field 1
        }
        eb = max(eb, 32u);                                      // plane scales Eb - 5p stay valid E8M0 bytes
        uint32_t frag[MX_P][6];
#pragma unroll
        for (int p = 0; p < MX_P; ++p)
#pragma unroll
            for (int q = 0; q < 6; ++q) frag[p][q] = 0u;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint32_t e = max((bits[j] >> 23) & 0xffu, 1u);
            const long long mant = (long long)((bits[j] & 0x7fffffu) | (((bits[j] >> 23) & 0xffu) ? 0x800000u : 0u));
            // Q = w / 2^L with L = Eb - 3 - 5 (MX_P - 1) (unbiased): mant * 2^(e - Eb - 20 + 5 (MX_P - 1))
            const int sh = (int)e - (int)eb - 20 + 5 * (MX_P - 1);
            long long q;
            if (sh >= 0) q = mant << sh;
            else if (sh <= -40) q = 0;
            else {                                              // round to nearest, ties to even
                const long long half = 1ll << (-sh - 1), mask = (1ll << -sh) - 1;
                const long long fl = mant >> -sh, rem = mant & mask;
                q = fl + ((rem > half || (rem == half && (fl & 1))) ? 1 : 0);
            }
            if (bits[j] >> 31) q = -q;
#pragma unroll
            for (int p = MX_P - 1; p >= 0; --p) {
                int d;
                if (p > 0) { d = (int)(((q + 16) & 31) - 16); q = (q - d) >> 5; }
                else d = (int)q;                                // |d| <= 16 by construction
                const uint32_t code = mx_fp6_code(d);
                const int bit = 6 * j;
                frag[p][bit >> 5] |= code << (bit & 31);
                if ((bit & 31) > 26) frag[p][(bit >> 5) + 1] |= code >> (32 - (bit & 31));
            }
        }
#pragma unroll
        for (int p = 0; p < MX_P; ++p) {
            uint32_t* x = X + (((size_t)p * Kc + kc) * Np + n) * 16 + lg * 4;
            uint32_t* y = Y + (((size_t)p * Kc + kc) * Np + n) * 8 + lg * 2;
            x[0] = frag[p][0]; x[1] = frag[p][1]; x[2] = frag[p][2]; x[3] = frag[p][3];
            y[0] = frag[p][4]; y[1] = frag[p][5];
        }
        S[((size_t)kc * Np + n) * 4 + lg] = (uint8_t)eb;
    }
}

// ------------------------------------------------------------------------------------------------
// k_gemm_mx<MODE>: the spike GEMMs on the block-scaled fp4 x fp6 matrix path.
//
// Work-group = 8 waves x 1 = 512 rows x 64 columns, wave = 64 x 64 = 4 x 4 tiles of v_mfma_scale_f32_16x16x128_f8f6f4.
// A 128-deep chunk (4 spike words per row) is multiplied in two micro-steps of 3 digit planes each (least significant
// planes first): 12 groups (plane, N-tile) of 4 MFMAs per micro-step - the cadence of k_gemm_bf16x3.
//   A: the row's 4 spike words are copied global -> LDS by LDS-DMA (16 B per row, no registers) as raw words (conv: the
//      encoder planes carry a one-position zero halo, so a 3x3 tap is a plain offset with no border logic), and a
//      lane's fragment (32 fp4 = one spike word) is 4 reads of a byte -> 8-nibble table; built once per
//      chunk, used by all 6 planes.
//   B: per micro-step 3 planes x (64 rows x 64 B + 64 rows x 32 B) = 18 KB by LDS-DMA into a 3-slot ring, two
//      micro-steps ahead; a lane's 24-B fragment = ds_read_b128 + ds_read_b64.  Scales: one dword (4 bytes = the 4
//      k-groups) per column and chunk, staged with the spike words.
// Modes: G3_FC, G3_CONV, G3_CONV_LIF_TILE, G3_FC_LIF_TILE as in k_gemm_bf16x3 (the T-in-tile epilogue is the 8 x 1 one).
// Requires K (fc) / C_in (conv) to be a multiple of 128.
// ------------------------------------------------------------------------------------------------
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));
struct MxB { v4i_t lo; v2i_t hi; };             // a lane's 24-B fp6 fragment
#define MX_BM 512
#define MX_BN 64
#define MX_LUT_BYTES 1024
#define MX_AW_BYTES (MX_BM * 16)                    // 4 spike words per row
#define MX_SC_BYTES 256                             // 64 columns x 4 scale bytes
#define MX_SLOT (3 * (MX_BN * 64 + MX_BN * 32))     // 18432: X and Y parts of three planes
#define MX_RING_OFF (MX_LUT_BYTES + 2 * MX_AW_BYTES + 2 * MX_SC_BYTES)
#define MX_PARK_OFF (MX_RING_OFF + 3 * MX_SLOT)     // conv: 8 bytes per thread of rarely used addressing state
#define MX_LDS (MX_PARK_OFF + 512 * 8)              // 77312 B

struct MxArgs {
    Gemm3Args g;                 // shared fields: A, out, M, Np, ldo, n_blocks, Cw, P_total, n_levels, T, pb, spk, spk_stride, p, lv, enc_stride
    const uint32_t* wq;          // digit planes (snn_mx.h layout)
    int Kc;                      // 128-deep chunks
};

// fp4 (A) x fp6 e2m3 (B) for digit plane `plane`: the B scale is the block's exponent byte Eb (low byte of `scb`), the
// plane's 2^(-5 plane) rides on the A scale (spikes are 1.0): byte plane & 3 of MX_ASC_LO / MX_ASC_HI; the op_sel fields are
// immediates.  (The builtin takes 8-dword operands; only 4 (fp4) / 6 (fp6) are read.  They are widened HERE with undefined
// upper lanes: fragments carried through the loop as 8-dword values cost 8 registers each and zero fills.)
#define MX_ASC_LO 0x70757A7Fu                       // E8M0 bytes 127, 122, 117, 112  = 2^0, 2^-5, 2^-10, 2^-15
#define MX_ASC_HI 0x0000666Bu                       //            107, 102            = 2^-20, 2^-25
__device__ __forceinline__ f32x4 mx_mfma(const v4i_t a4, const MxB b6, const f32x4 c, const int plane, const uint32_t asc_lo,
                                         const uint32_t asc_hi, const uint32_t scb) {
    const v8i_t a = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1);
    const v4i_t h4 = __builtin_shufflevector(b6.hi, b6.hi, 0, 1, -1, -1);
    const v8i_t b = __builtin_shufflevector(b6.lo, h4, 0, 1, 2, 3, 4, 5, -1, -1);
    switch (plane) {
    case 0: return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 2, 0, asc_lo, 0, scb);
    case 1: return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 2, 1, asc_lo, 0, scb);
    case 2: return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 2, 2, asc_lo, 0, scb);
    case 3: return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 2, 3, asc_lo, 0, scb);
    case 4: return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 2, 0, asc_hi, 0, scb);
    default: return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 2, 1, asc_hi, 0, scb);
    }
}

__device__ __forceinline__ const uint32_t* sgpr_u32(const uint32_t* p) { return static_cast<const uint32_t*>(sgpr_ptr(p)); }

// MW = 16-row M-tiles per wave: 4 -> 8 waves of 64 rows, 128 registers, two work-groups (4 waves) per SIMD;
//                              8 -> 4 waves of 128 rows, 256 registers, two work-groups (2 waves) per SIMD: every B fragment
//                                   read from LDS feeds 8 MFMAs instead of 4.
template <int MODE, int MW>
__global__ __launch_bounds__(2048 / MW, 16 / MW) void k_gemm_mx(const MxArgs ma) {
    constexpr bool CONV = MODE == G3_CONV || MODE == G3_CONV_LIF_TILE;
    constexpr bool TILE = MODE == G3_CONV_LIF_TILE || MODE == G3_FC_LIF_TILE;
    constexpr int NWV = 32 / MW, WROWS = 16 * MW, RPT = MW / 4;  // waves, rows per wave, rows staged per thread
    constexpr int PD = MW == 8 ? 2 : MX_PD, RING = PD + 1;      // fragment prefetch distance in groups; RING divides 12
    constexpr int NPC = MW == 8 ? 5 : 3;                        // B pieces a wave copies per micro-step
    const Gemm3Args& args = ma.g;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned char* const lut = smem;                                        // byte -> 8 fp4 nibbles (0 / 1.0 = 0x2)
    unsigned char* const awb = smem + MX_LUT_BYTES;                         // [2][512 rows][16 B]
    unsigned char* const scb = awb + 2 * MX_AW_BYTES;                       // [2][64 columns] dword
    unsigned char* const ring = smem + MX_RING_OFF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);              // = row block of WROWS
    const int nb = blockIdx.x % args.n_blocks, mb = blockIdx.x / args.n_blocks;
    const int m0 = TILE ? mb * args.pb : mb * MX_BM;
    const int Kc = ma.Kc, Np = args.Np, M = args.M;
    const int lr = lane & 15, lg = lane >> 4;

    if (smem_base != 0) __builtin_trap();                                   // rd_a addresses the table absolutely
    for (int e = tid; e < 256; e += 64 * NWV) {
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) v |= ((e >> b) & 1u) << (4 * b + 1);
        reinterpret_cast<uint32_t*>(lut)[e] = v;
    }

    // ---- A staging: a thread owns RPT rows (wave * WROWS + i * 64 + lane).  One 16-B load per row and chunk: scalar base +
    // 32-bit lane offset ----
    const int Kw = CONV ? args.Cw : Kc * 4;                                 // spike words per row (fc: K / 32)
    uint32_t f_voff[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int xrow = wave * WROWS + i * 64 + lane;
        const int xt = TILE ? xrow / args.pb : 0;
        const int xm = TILE ? (xt < args.Tc ? m0 + xrow % args.pb : M) : m0 + xrow;
        uint32_t a_off = 0;
        int a_pitch = 0;
        if (CONV) {
            // the encoder planes carry a one-position zero halo around every image: every tap of every position is a plain
            // read (rows past M read the first halo row: zeros)
            if (xm < M) {
                const int t = TILE ? xt + args.t0 : xm / args.P_total, p = TILE ? xm : xm % args.P_total;
                int l = 0;
                while (l + 1 < args.n_levels && p >= args.lv[l + 1].pos_base) ++l;
                const int H = args.lv[l].H, W = args.lv[l].W;
                const int local = p - args.lv[l].pos_base;
                const int n = local / (H * W), rem = local % (H * W);
                const int y = rem / W, x = rem % W;
                const size_t prow = (size_t)args.lv[l].tile_begin + ((size_t)n * (H + 2) + y + 1) * (W + 2) + x + 1;
                a_off = (uint32_t)(((size_t)t * args.enc_stride + prow * args.Cw) * 4);
                a_pitch = (W + 2) * args.Cw * 4;
            } else {
                a_off = (uint32_t)((args.lv[0].W + 3) * args.Cw * 4);      // (y, x) = (0, 0) of level 0, image 0: its taps are in range
                a_pitch = (args.lv[0].W + 2) * args.Cw * 4;
            }
            // a_off and a_pitch are needed once per tap row / per pass only; parked in LDS so that they do not hold
            // registers through the MFMA loop
            reinterpret_cast<uint2*>(smem + MX_PARK_OFF)[xrow] = make_uint2(a_off, (uint32_t)a_pitch);
        } else if (TILE) {
            a_off = xm < M ? (uint32_t)(((size_t)(xt + args.t0) * M + xm) * Kw * 4) : 0u;
        } else {
            a_off = (uint32_t)((size_t)min(xm, M - 1) * Kw * 4);
        }
        f_voff[i] = CONV ? a_off - (uint32_t)a_pitch : a_off;              // fc: constant
    }
    auto park_get = [&](int i) {
        uint32_t l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return reinterpret_cast<const uint2*>(smem + MX_PARK_OFF)[wave * WROWS + i * 64 + l];
    };
    // fetch stream over the chunk sequence (tap dy, tap dx, 128-channel group): running scalar source pointer
    const int cpt = __builtin_amdgcn_readfirstlane(CONV ? args.Cw / 4 : 0);  // chunks per tap
    const uint32_t* f_ptr = sgpr_u32(args.A + (CONV ? -args.Cw : 0));
    int f_kc = 0, f_cc = 0, f_dx = 0;
    // The rows' 16 bytes go straight into LDS by LDS-DMA (lane L -> base + 16 L: exactly the [row][4 words] layout).
    const uint32_t* const Sg = ma.wq + mx_x_words(Kc, Np) + mx_y_words(Kc, Np) + nb * MX_BN;
    auto fetch_next = [&](int par) {                                        // spike words of the next chunk
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + MX_LUT_BYTES + par * MX_AW_BYTES + (wave * WROWS + i * 64) * 16);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(f_voff[i]), "s"(f_ptr), "s"(dst) : "memory");
        }
        // (the stream position is wave-uniform: say so, or the counters live in vector registers)
        f_ptr = sgpr_u32(f_ptr + 4);
        if (CONV) {
            f_cc = __builtin_amdgcn_readfirstlane(f_cc + 1);
            if (f_cc == cpt) {
                f_cc = 0;
                f_dx = __builtin_amdgcn_readfirstlane(f_dx + 1);
                if (f_dx == 3) {                                            // next tap row
                    f_dx = 0;
                    f_ptr = sgpr_u32(f_ptr - 3 * args.Cw);
#pragma unroll
                    for (int i = 0; i < RPT; ++i) f_voff[i] += park_get(i).y;
                }
            }
        }
        f_kc = __builtin_amdgcn_readfirstlane(f_kc + 1);
        if (f_kc == Kc) {                                                   // past the end: wrap (staged, never multiplied)
            f_kc = 0; f_cc = 0; f_dx = 0;
            f_ptr = sgpr_u32(args.A + (CONV ? -args.Cw : 0));
            if (CONV) {
#pragma unroll
                for (int i = 0; i < RPT; ++i) { const uint2 pk = park_get(i); f_voff[i] = pk.x - pk.y; }
            }
        }
    };

    // The chunk's column scales (one dword per column) are shared by all waves, so their buffer may only be overwritten
    // once every wave has provably read it: load_scales(par) runs right after the barrier that ends chunk kc-1, its reads
    // are drained (lgkmcnt(0)) before the barrier that ends micro-step (kc, h=0), and only then - in micro-step (kc, h=1) -
    // wave 0 copies the scales of chunk kc+2 over them.  (Issued in (kc, h=0) the copy usually lands after the reads, but
    // nothing orders it: the younger waves of a SIMD can lag far enough to read the new bytes.)
    int sc_kc = 0;
    auto fetch_scales = [&](int par) {
        if (wave == 0) {
            const uint32_t sdst = __builtin_amdgcn_readfirstlane(smem_base + MX_LUT_BYTES + 2 * MX_AW_BYTES + par * MX_SC_BYTES);
            uint32_t l;                             // lane id, re-derived instead of a register held all loop
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            const uint32_t sc_voff = min(l, (uint32_t)(Np - nb * MX_BN - 1)) * 4u;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1" :: "v"(sc_voff), "s"(sgpr_ptr(Sg + (size_t)sc_kc * Np)), "s"(sdst) : "memory");
        }
        sc_kc = __builtin_amdgcn_readfirstlane(sc_kc + 1);
        if (sc_kc == Kc) sc_kc = 0;
    };

    // ---- B staging: per micro-step 3 planes x (4 X pieces + 2 Y pieces) of 1 KiB = 18 pieces; every wave issues
    // exactly NPC (8 waves: piece = wave, wave + 8, wave + 16; 4 waves: X pieces wave, wave + 4, wave + 8 and Y pieces wave,
    // 4 + (wave & 1); surplus slots repeat Y pieces - same bytes, harmless), with all piece parameters fixed before the
    // loop: branch-free staging, running scalar source pointers.  Every X piece of a wave is part wave & 3 of its plane
    // and every Y piece part wave & 1, so there are two lane offsets ----
    const unsigned long long xplane = (unsigned long long)Kc * Np * 64, yplane = (unsigned long long)Kc * Np * 32;   // bytes per plane
    const unsigned char* const Xg = reinterpret_cast<const unsigned char*>(ma.wq);
    const unsigned char* const Yg = Xg + (size_t)MX_P * xplane;
    unsigned long long pc_base[NPC], pc_plane3[NPC], pc_ptr[NPC];           // scalar: source of (chunk 0, half 0), 3 planes, running
    uint32_t pc_cstride[NPC], pc_dst[NPC], pc_voff[NPC];                    // chunk stride, LDS offset in the slot (scalar); lane offset
    {
        const int ncols = Np - nb * MX_BN;                                  // columns this block really has (>= 32)
        const int rx = (wave & 3) * 16 + (lane >> 2);
        // Y rows are 32 B: rows r and r + 8 share banks for an 8-byte read, so rows with bit 3 set keep their two 16-B
        // halves swapped (the copy permutes the SOURCE address; LDS-DMA always lands lane L at base + 16 L)
        const int ry = (wave & 1) * 32 + (lane >> 1);
        const uint32_t vx = (uint32_t)(min(rx, ncols - 1) * 64 + (((lane & 3) ^ G3_SWZ(rx)) << 4));
        const uint32_t vy = (uint32_t)(min(ry, ncols - 1) * 32 + (((lane & 1) ^ ((ry >> 3) & 1)) << 4));
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            int piece;
            if (NWV == 8) { piece = wave + 8 * j; if (piece >= 18) piece -= 6; }        // repeat Y pieces 0..5
            else piece = j < 3 ? wave + 4 * j : (j == 3 ? 12 + wave : 16 + (wave & 1));
            const bool isx = NWV == 8 ? piece < 12 : j < 3;
            const int q = isx ? piece : piece - 12;
            const int pj = isx ? q >> 2 : q >> 1, part = isx ? q & 3 : q & 1;   // plane slot 0..2, 1-KiB part
            pc_voff[j] = isx ? vx : vy;
            pc_dst[j] = isx ? pj * (MX_BN * 64) + part * 1024 : 3 * (MX_BN * 64) + pj * (MX_BN * 32) + part * 1024;
            const unsigned long long plane = isx ? xplane : yplane;
            pc_base[j] = (unsigned long long)(isx ? Xg : Yg) + (5 - pj) * plane + (unsigned long long)nb * MX_BN * (isx ? 64 : 32);
            pc_plane3[j] = 3 * plane;
            pc_cstride[j] = (uint32_t)Np * (isx ? 64u : 32u);
            pc_ptr[j] = pc_base[j];
        }
    }
    int s_kc = 0, s_h = 0;                                                  // staging stream position (chunk, half)
    auto stage_next = [&](uint32_t slot_off) {
        const uint32_t dbase = smem_base + MX_RING_OFF + slot_off;
#pragma unroll
        for (int j = 0; j < NPC; ++j)
            glds16(sgpr_ptr(reinterpret_cast<const void*>(pc_ptr[j])), pc_voff[j], __builtin_amdgcn_readfirstlane(dbase + pc_dst[j]));
        if (s_h == 0) {                                                     // next: the three more significant planes of this chunk
            s_h = 1;
#pragma unroll
            for (int j = 0; j < NPC; ++j) pc_ptr[j] -= pc_plane3[j];
        } else {
            s_h = 0;
            const bool wrap = ++s_kc == Kc;
            if (wrap) s_kc = 0;
#pragma unroll
            for (int j = 0; j < NPC; ++j) pc_ptr[j] = wrap ? pc_base[j] : pc_ptr[j] + pc_plane3[j] + pc_cstride[j];
        }
    };

    f32x4 acc[MW][4];
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment readers
    // lane_q = 4 lr + lg: the scale byte of (column nt*16 + lr, k-group lg) sits at scb + nt*64 + lane_q, the spike word of
    // (row mt*16 + lr, k-group lg) at awb + mt*256 + 4 lane_q
    const uint32_t lane_q = (uint32_t)(lr * 4 + lg);
    auto rd_w = [&](int par, uint32_t (&w)[MW]) {
        const unsigned char* base = awb + par * MX_AW_BYTES + wave * (WROWS * 16) + lane_q * 4;
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) w[mt] = *reinterpret_cast<const uint32_t*>(base + mt * 256);
    };
    // 32 spike bits -> 32 fp4: four table reads at (byte b of w) * 4; the table sits at LDS address 0 (checked above), so
    // there is no base to add.  (Plain C on purpose: a vector instruction written as inline asm next to MFMAs is invisible to
    // hipcc's hazard recognizer - an asm result that landed on a register an in-flight MFMA still reads as its
    // accumulator input corrupted it.)
    typedef __attribute__((address_space(3))) const int* lds_int_p;
    auto rd_a = [&](uint32_t w) {
        v4i_t a = {0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 4; ++b) a[b] = *(lds_int_p)(uintptr_t)(((w >> (8 * b)) & 0xffu) << 2);
        return a;
    };
    const unsigned char* const bx_rd = ring + lr * 64 + ((lg ^ G3_SWZ(lr)) << 4);     // + slot, plane slot * 4096, nt * 1024
    const unsigned char* const by_rd = ring + 3 * (MX_BN * 64) + lr * 32 + (((lg >> 1) ^ (lr >> 3)) << 4) + (lg & 1) * 8;   // + slot, plane slot * 2048, nt * 512
    auto rd_b = [&](uint32_t slot_off, int g) {                                      // group g = (N-tile g / 3, plane slot g % 3)
        const int nt = g / 3, pj = g % 3;
        MxB b;
        b.lo = *reinterpret_cast<const v4i_t*>(bx_rd + slot_off + pj * (MX_BN * 64) + nt * 1024);
        b.hi = *reinterpret_cast<const v2i_t*>(by_rd + slot_off + pj * (MX_BN * 32) + nt * 512);
        return b;
    };

    const int n_ms = 2 * Kc;                                                // micro-steps
    // ---- prologue: spike words + scales of chunks 0 and 1, planes of micro-steps 0 and 1 ----
    fetch_next(0);
    fetch_next(1);
    fetch_scales(0);
    fetch_scales(1);
    stage_next(0);
    stage_next(MX_SLOT);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v4i_t af[MW];
    MxB bq[RING];
    uint32_t wq[MW], scq[4];                                                // scq[nt]: exponent byte Eb of the lane's block (column, k-group)
    uint32_t asc_lo = MX_ASC_LO, asc_hi = MX_ASC_HI;
    asm volatile("" : "+v"(asc_lo), "+v"(asc_hi));                          // two registers for the whole loop, never re-made
    auto load_scales = [&](int par) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) scq[nt] = (scb + par * MX_SC_BYTES + nt * 64)[lane_q];
    };
    rd_w(0, wq);
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) af[mt] = rd_a(wq[mt]);
    load_scales(0);
#pragma unroll
    for (int g = 0; g < PD; ++g) bq[g] = rd_b(0, g);

#ifdef SNN_EXP_MX_BAR2
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
    uint32_t o_cur = 0, o_nxt = MX_SLOT, o_wr = 2 * MX_SLOT;
    int kc = 0;                                                              // chunk of the current micro-step
    for (int ms0 = 0; ms0 < n_ms; ms0 += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                        // micro-step ms0 + h: planes 5-3h .. 3-3h
            const int par = kc & 1;
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                const int gp = g + PD;
#ifndef SNN_EXP_MX_NOREADB
                bq[gp % RING] = gp < 12 ? rd_b(o_cur, gp) : rd_b(o_nxt, gp - 12);
#endif
#ifndef SNN_EXP_MX_NOSTAGE
                // spike words of chunk kc+2 into the (wave-private) rows of chunk kc, whose fragments were built a chunk ago
                if (h == 0 && g == 2) fetch_next(par);
                if (h == 1 && g == 2) fetch_scales(par);
                if (g == 3) stage_next(o_wr);                                // planes of micro-step +2
#endif
#ifndef SNN_EXP_MX_RDW_G
#define SNN_EXP_MX_RDW_G 8
#endif
                if (h == 1 && g == SNN_EXP_MX_RDW_G) rd_w(par ^ 1, wq);
                const int nt = g / 3, pl = 5 - 3 * h - g % 3;
#pragma unroll
                for (int mt = 0; mt < MW; ++mt) {
                    acc[mt][nt] = mx_mfma(af[mt], bq[g % RING], acc[mt][nt], pl, asc_lo, asc_hi, scq[nt]);
                    // the chunk's last use of af[mt]: rebuild it for the next chunk right away
#ifndef SNN_EXP_MX_NOA
                    if (h == 1 && g == 11) af[mt] = rd_a(wq[mt]);
#endif
                }
                // Pin the group: the MFMAs are pure register operations whose only users are the next MFMAs of their
                // accumulation chains, and hipcc otherwise SINKS a whole micro-step of them past the barrier into the next
                // one (all 12 fragments live across it, accumulators spilled).  An empty volatile asm that "modifies" the
                // accumulators keeps them in program order with the staging asm and the barrier fences.
                if (MW == 8)
                    asm volatile("" : "+v"(acc[0][nt]), "+v"(acc[1][nt]), "+v"(acc[2][nt]), "+v"(acc[3][nt]), "+v"(acc[4 % MW][nt]),
                                 "+v"(acc[5 % MW][nt]), "+v"(acc[6 % MW][nt]), "+v"(acc[7 % MW][nt]) :: "memory");
                else
                    asm volatile("" : "+v"(acc[0][nt]), "+v"(acc[1][nt]), "+v"(acc[2][nt]), "+v"(acc[3][nt]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0) lgkmcnt(0)
#ifndef SNN_EXP_MX_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            if (h == 1) {
                load_scales(par ^ 1);                                        // scales of chunk kc+1
                if (++kc == Kc) kc = 0;
            }
            { const uint32_t o = o_cur; o_cur = o_nxt; o_nxt = o_wr; o_wr = o; }
        }
    }

    if (TILE) {
        // ---- LIF over the T time steps held in this tile (the 8 x 1 epilogue of k_gemm_bf16x3): two passes of 32 columns ----
        constexpr int CG = 32, PITCH = CG + 4;
        float* const tile = reinterpret_cast<float*>(smem);
        uint32_t* const pos_cnt = reinterpret_cast<uint32_t*>(smem + G3_TILE_BYTES(1));     // behind the tile image
        const bool counting = args.cnt_img != nullptr || args.cnt_row != nullptr;
        const int pb = args.pb, T = args.T, t0 = args.t0, t1 = args.t0 + args.Tc;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < MW; ++mt)
#pragma unroll
                for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        tile[(wave * WROWS + mt * 16 + lg * 4 + r) * PITCH + nq * 16 + lr] = h == 0 ? acc[mt][nq][r] : acc[mt][2 + nq][r];
            __syncthreads();
            const int word0 = (nb * MX_BN + h * CG) >> 5;
            if (word0 * 32 >= Np) continue;
            const int par = lane >> 5, col = lane & 31;
            for (int pp = wave; 2 * pp < pb; pp += NWV) {
                const int pi = 2 * pp + par;
                const bool live = pi < pb && m0 + pi < M;
                if (m0 + 2 * pp >= M) break;
                float vv = args.p.v_leak, ii = 0.0f;
                uint32_t my0 = 0, my1 = 0;
                const float* src = tile + (live ? pi : 2 * pp) * PITCH + col;
                uint32_t cnt0 = 0, cnt1 = 0;               // wave-uniform: spikes of the even / odd position in this pass
                for (int t = 0; t < T; ++t) {
                    float cur = 0.0f;                      // steps outside the tile's window: no input current (Gemm3Args.Tc)
                    if (t >= t0 && t < t1) cur = src[(size_t)(t - t0) * pb * PITCH];
                    const bool z = lif_step(cur, vv, ii, args.p);
                    const unsigned long long b = __ballot(z);
                    my0 = lane == t ? (uint32_t)b : my0;
                    my1 = lane == t ? (uint32_t)(b >> 32) : my1;
                    cnt0 += __popc((uint32_t)b);
                    cnt1 += __popc((uint32_t)(b >> 32));
                }
                const bool odd_ok = 2 * pp + 1 < pb && m0 + 2 * pp + 1 < M;
                if (lane < T) {
                    uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + (size_t)(m0 + 2 * pp) * (Np >> 5) + word0;
                    dst[0] = my0;
                    if (odd_ok) dst[Np >> 5] = my1;
                }
                if (counting && lane == 0) {
                    pos_cnt[2 * pp] = (h == 0 ? 0u : pos_cnt[2 * pp]) + cnt0;
                    if (odd_ok) pos_cnt[2 * pp + 1] = (h == 0 ? 0u : pos_cnt[2 * pp + 1]) + cnt1;
                }
            }
        }
        if (counting) {
            __syncthreads();
            tile_counts_flush<CONV>(args, pos_cnt, m0, pb, tid);
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nb * MX_BN + nt * 16 + lr;
            if (col >= Np) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wave * WROWS + mt * 16 + lg * 4 + r;
                if (m < M) args.out[(size_t)m * args.ldo + col] = acc[mt][nt][r];
            }
        }
}
