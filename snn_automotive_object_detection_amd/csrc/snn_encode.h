// constant-current LIF encoders -> spike bit-planes: NCHW pyramid levels (K1a), row-major RoI features (K1b, K1b'),
// and the RoIAlign-fused form (K1c).  Included by snn_kernels.hip.
#pragma once

// ------------------------------------------------------------------------------------------------
// K1a: constant-current LIF encoder, NCHW fp32 -> bit-planes [T][N*HW][Cw]
// Block = 64 positions x 4 channel words; thread = one position x 32 channels, a wave = 64 consecutive positions of one channel
// word (every load instruction reads one 256-byte run of a channel plane; round 2's 32 x 8 shape read two 128-byte segments of
// two planes per instruction and topped out at 3.3 TB/s once the threshold form had removed the VALU bound).  The T words of a
// thread go through LDS so that the plane stores of the block are runs of consecutive positions (word-major planes: 256 bytes).
// VALU-bound (32 x T encoder steps per thread): packed fp32 arithmetic, see enc_step2_word.
// ------------------------------------------------------------------------------------------------
#define ENC_PB 64                                   // positions per block (a wave = 64 consecutive positions of one channel: 256-byte runs)
#define ENC_WB 4                                    // channel words per block
#define ENC_LDS_BYTES(T) ((size_t)(T) * ENC_PB * (ENC_WB + 1) * 4)
// Wpad > 0: the planes carry a one-position zero halo around every image (row (n, y, x) -> (n*(H+2) + y+1)*(W+2) + x+1,
// W = Wpad), written here as well (by the threads of the border positions).  The conv kernels of the bf16x3 and mxfp6 families
// read their 3x3 taps from such planes without any border logic.
// cmp != nullptr (round 5; word-major period planes of the bf16x3 RPN head whose conv runs the structured-sparse launch, csrc/snn_sparse.h): the
// planes e_n, n > nd, leave COMPRESSED - the four dwords per (row, 64 k) k_compress_planes would make of them (primary occupancy, index
// halves, secondary occupancy), computed from the block's words while they are in LDS - and their raw words are not written at all:
// one launch and a 32-MB write + 32-MB read less per head (k_compress_planes stays for the stage-level entry points and the linear layers).
template <int EM>
__device__ __forceinline__ void encode_block(const float* __restrict__ feat, int C, int HW, int Cw, int T, const NeuronP& p,
                                             const EncTh& eth, uint32_t* __restrict__ planes, size_t plane_stride, int n, int bx, int by,
                                             int Wpad = 0, size_t wm_rows = 0, uint32_t* __restrict__ cmp = nullptr, int nd = 0) {
    constexpr bool ZR = EM != ENC_GENERIC;
    __shared__ uint16_t nib_code[256];
    if (cmp) nib_code[threadIdx.x] = sp_byte_code(threadIdx.x);          // (256 threads; the block's barrier below comes before the first use)
    // dynamic: ENC_LDS_BYTES(T) - sized by the launch's T, not by SNN_MAX_STEPS (40 KB, four blocks per CU: the loads of 16 waves did not
    // cover the HBM latency; T = 8: 9 KB)
    extern __shared__ uint32_t wbuf[];                                     // [t][position][4 words + 1 pad]
    const int pl = threadIdx.x & (ENC_PB - 1), cgl = threadIdx.x / ENC_PB;
    const int pos = bx * ENC_PB + pl;
    const int cg = by * ENC_WB + cgl;
    float x[32], v[32];
    if (cg * 32 + 32 <= C) {
        // all 32 channels of the word exist (wave-uniform: a wave is one channel word): 32 plain loads off ONE address - the per-channel
        // predicate cost 14 instructions per load, as many as the encoder steps themselves (round 5: the launch was VALU-bound at 62 % of
        // the HBM rate).  Lanes past the level's last position read its last position; their words are never stored
        const float* src = feat + ((size_t)n * C + cg * 32) * HW + min(pos, HW - 1);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            x[j] = src[(size_t)j * HW];
            v[j] = 0.0f;                          // rpn.py:93  v = zeros
        }
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int c = cg * 32 + j;
            x[j] = (pos < HW && c < C) ? feat[((size_t)n * C + c) * HW + pos] : 0.0f;
            v[j] = 0.0f;
        }
    }
    uint32_t prev = 0;
    for (int t = 0; t < T; ++t) {
        uint32_t word = 0;
        if (EM == ENC_QUANT) {                      // period planes by thresholds: cumulative word, then the newly fired bits
            const float th = eth.th[t];
#pragma unroll
            for (int j = 31; j >= 0; --j) enc_quant_word(x[j], th, word);
            const uint32_t cum = word;
            word = cum & ~prev;
            prev = cum;
        } else {
#pragma unroll
            for (int j = 31; j >= 0; --j) enc_step_word<ZR>(x[j], v[j], p, word);   // bit 31 first ... bit 0 last
        }
        wbuf[(t * ENC_PB + pl) * (ENC_WB + 1) + cgl] = word;
    }
    __syncthreads();
    // store: thread -> (position tid >> 3, word tid & 7): consecutive threads write consecutive plane words.
    // wm_rows > 0: word-major planes [T][word][wm_rows rows] (`planes` then points at the level's first ROW of word plane 0):
    // thread -> (word tid >> 5, position tid & 31), 128-byte runs of 32 positions per (t, word)
    const int sp = wm_rows ? (threadIdx.x & (ENC_PB - 1)) : (threadIdx.x / ENC_WB), sw = wm_rows ? (threadIdx.x / ENC_PB) : (threadIdx.x % ENC_WB);
    const int spos = bx * ENC_PB + sp, scg = by * ENC_WB + sw;
    if (spos < HW && scg < Cw) {
        size_t row = (size_t)n * HW + spos;
        if (Wpad) {
            const int H = HW / Wpad, y = spos / Wpad, x = spos % Wpad;
            row = ((size_t)n * (H + 2) + y + 1) * (Wpad + 2) + x + 1;
        }
        uint32_t* out = wm_rows ? planes + (size_t)scg * wm_rows + row : planes + row * Cw + scg;
        const int t_raw = cmp ? min(T, nd) : T;                // (compressed planes: no raw words)
        for (int t = 0; t < t_raw; ++t) out[(size_t)t * plane_stride] = wbuf[(t * ENC_PB + sp) * (ENC_WB + 1) + sw];
        // compressed planes: the two threads of a word pair (sw = 2 pair + par) take alternate planes; array j of (plane ts, pair w2) starts
        // at cmp + ((ts Cw / 2 + w2) 4 + j) wm_rows (k_compress_planes' layout)
        const int w2 = scg >> 1, par = sw & 1;
        uint32_t* cout = cmp ? cmp + (size_t)w2 * SP_A_ARR * wm_rows + row : nullptr;
        const size_t cmp_plane = (size_t)(Cw / 2) * SP_A_ARR * wm_rows;
        if (cmp && 2 * w2 + 1 < Cw) {
            for (int t = nd + par; t < T; t += 2) {
                const uint32_t* wp = wbuf + (t * ENC_PB + sp) * (ENC_WB + 1) + (sw & ~1);
                uint32_t c4[4];
                sp_compress_pair(wp[0], wp[1], nib_code, c4);
                uint32_t* o = cout + (size_t)(t - nd) * cmp_plane;
#pragma unroll
                for (int j = 0; j < SP_A_ARR; ++j) o[(size_t)j * wm_rows] = c4[j];
            }
        }
        if (Wpad) {
            // the zero halo, written by the threads that own the image's border positions (round 4: a separate k_zero_halo launch
            // did this before - 5-6 us of a 1.1-ms head at T_rpn = 4): a border position zeroes the padded rows / columns next to it,
            // corner positions the corners too.  dr = row offset to the neighbour (in padded rows), rs = plane words per padded row
            const int H = HW / Wpad, y = spos / Wpad, x = spos % Wpad, W2 = Wpad + 2;
            const size_t rs = wm_rows ? 1 : (size_t)Cw;
            auto zero = [&](const long long dr) __attribute__((always_inline)) {
                uint32_t* z = out + dr * (long long)rs;
                for (int t = 0; t < t_raw; ++t) z[(size_t)t * plane_stride] = 0u;
                if (cmp && 2 * w2 + 1 < Cw) {                  // (an empty word pair compresses to occupancy 0 and the index pattern (0, 3) per nibble)
                    for (int t = nd + par; t < T; t += 2) {
                        uint32_t* o = cout + dr + (size_t)(t - nd) * cmp_plane;
                        o[0] = 0u; o[wm_rows] = 0xCCCCCCCCu; o[2 * wm_rows] = 0xCCCCCCCCu; o[3 * wm_rows] = 0u;
                    }
                }
            };
            const bool top = y == 0, bot = y == H - 1, lef = x == 0, rig = x == Wpad - 1;
            if (top) zero(-W2);
            if (bot) zero(W2);
            if (lef) zero(-1);
            if (rig) zero(1);
            if (top && lef) zero(-W2 - 1);
            if (top && rig) zero(-W2 + 1);
            if (bot && lef) zero(W2 - 1);
            if (bot && rig) zero(W2 + 1);
        }
    }
}

template <int EM>
__global__ __launch_bounds__(256) void k_encode_nchw(const float* __restrict__ feat, int C, int HW, int Cw,
                                                     int T, NeuronP p, const EncTh eth, uint32_t* __restrict__ planes,
                                                     size_t plane_stride) {
    encode_block<EM>(feat, C, HW, Cw, T, p, eth, planes, plane_stride, blockIdx.z, blockIdx.x, blockIdx.y);
}

// all pyramid levels of the RPN head in one launch (the small levels are launch-latency bound on their own)
struct EncLevels {
    const float* feat[SNN_MAX_LEVELS];
    int HW[SNN_MAX_LEVELS], bpi[SNN_MAX_LEVELS];   // positions / blocks per image
    int blk_base[SNN_MAX_LEVELS + 1];               // first block of the level
    int pos_base[SNN_MAX_LEVELS];                   // first plane row of the level
    int Wpad[SNN_MAX_LEVELS];                       // 0, or the level's width when the planes carry a zero halo
    int n_levels;
};
template <int EM>
__global__ __launch_bounds__(256) void k_encode_levels(const EncLevels lv, int C, int Cw, int T, NeuronP p, const EncTh eth,
                                                       uint32_t* __restrict__ planes, size_t plane_stride, size_t wm_rows,
                                                       uint32_t* __restrict__ cmp, int nd) {
    int l = 0;
    while (l + 1 < lv.n_levels && (int)blockIdx.x >= lv.blk_base[l + 1]) ++l;
    const int local = blockIdx.x - lv.blk_base[l];
    encode_block<EM>(lv.feat[l], C, lv.HW[l], Cw, T, p, eth, planes + (size_t)lv.pos_base[l] * (wm_rows ? 1 : Cw), plane_stride,
                     local / lv.bpi[l], local % lv.bpi[l], blockIdx.y, lv.Wpad[l], wm_rows, cmp ? cmp + lv.pos_base[l] : nullptr, nd);
}

// K1b: encoder on row-major x[R][D] -> bit-planes [T][R][Dw]; a wave covers 64 consecutive reduction indices per
// slot, so one ballot per step IS two plane words.  Each thread runs ENC_U independent elements (64 apart) to keep
// several loads and scan chains in flight.
#define ENC_U 4
template <int EM>
__global__ __launch_bounds__(256) void k_encode_rows(const float* __restrict__ x, int R, int D, int Dw, int T,
                                                     NeuronP p, const EncTh eth, uint32_t* __restrict__ planes,
                                                     size_t plane_stride) {
    constexpr bool ZR = EM != ENC_GENERIC;
    const size_t Dp = (size_t)Dw * 32;
    const size_t total = (size_t)R * Dp;
    const int lane = threadIdx.x & 63;
    const size_t wave_base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * ENC_U);
    float xv[ENC_U], v[ENC_U];
    size_t e[ENC_U];
#pragma unroll
    for (int u = 0; u < ENC_U; ++u) {
        e[u] = wave_base + (size_t)u * 64 + lane;
        xv[u] = 0.0f;
        v[u] = 0.0f;                              // faster_rcnn.py:484
        if (e[u] < total) {
            const size_t r = e[u] / Dp, k = e[u] % Dp;
            if (k < (size_t)D) xv[u] = x[r * D + k];
        }
    }
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int u = 0; u < ENC_U; ++u) {
            const bool z = EM == ENC_QUANT ? (xv[u] >= eth.th[t] && !(t > 0 && xv[u] >= eth.th[t - 1])) : enc_step_t<ZR>(xv[u], v[u], p);
            const unsigned long long m = __ballot(z);
            if ((lane & 31) == 0 && e[u] < total)
                planes[(size_t)t * plane_stride + (e[u] >> 5)] = (lane < 32) ? (uint32_t)m : (uint32_t)(m >> 32);
        }
    }
}

// K1b': the same encoder when D is a multiple of 32 (the detector: 12544): the [R][D] elements are then one contiguous
// run of plane words, and a LANE owns a whole word.  A work-group copies 256 words (32 KB) with 16-byte coalesced loads
// through LDS (row pitch 36 dwords: the 8 ds_read_b128 of a lane are conflict-free), then every lane scans its 32
// neurons with the 6-instruction step whose carry chain builds the plane word (enc_step_word) and the wave stores 64
// consecutive words per time step.  Against the ballot form: 6 instead of 11 vector instructions per neuron-step and 8
// times the bytes in flight per thread.
#define ENC_W_PITCH 36
template <int EM>
__global__ __launch_bounds__(256) void k_encode_rows_w(const float* __restrict__ x, size_t n_words, int T, NeuronP p, const EncTh eth,
                                                       uint32_t* __restrict__ planes, size_t plane_stride) {
    constexpr bool ZR = EM != ENC_GENERIC;
    __shared__ __attribute__((aligned(16))) float tile[256 * ENC_W_PITCH];
    const int tid = threadIdx.x;
    const size_t w0 = (size_t)blockIdx.x * 256;                 // first plane word of the work-group
    const size_t nw = n_words - w0 < 256 ? n_words - w0 : 256;
    const f32x4* src = reinterpret_cast<const f32x4*>(x + w0 * 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = tid + 256 * j;                            // 16-byte piece q of the work-group: word q / 8, elements 4 (q % 8) ..
        f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
        if ((size_t)(q >> 3) < nw) v4 = src[q];
        *reinterpret_cast<f32x4*>(tile + (q >> 3) * ENC_W_PITCH + (q & 7) * 4) = v4;
    }
    __syncthreads();
    if ((size_t)tid >= nw) return;
    float xv[32], v[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(tile + tid * ENC_W_PITCH + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) { xv[4 * q + r] = t4[r]; v[4 * q + r] = 0.0f; }      // v = 0: faster_rcnn.py:484
    }
    uint32_t* dst = planes + w0 + tid;
    uint32_t prev = 0;
    for (int t = 0; t < T; ++t) {
        uint32_t word = 0;
        if (EM == ENC_QUANT) {                      // period planes by thresholds (snn_common.h)
            const float th = eth.th[t];
#pragma unroll
            for (int b = 31; b >= 0; --b) enc_quant_word(xv[b], th, word);
            const uint32_t cum = word;
            word = cum & ~prev;
            prev = cum;
        } else {
#pragma unroll
            for (int b = 31; b >= 0; --b) enc_step_word<ZR>(xv[b], v[b], p, word);          // bit 31 first
        }
        dst[(size_t)t * plane_stride] = word;
    }
}

// K1b'': the same encoder writing WORD-MAJOR planes [T][Dw][R] (what the linear-layer kernels stream best): a work-group takes
// 32 rows x 8 words (32 x 1 KB of x, 16-byte coalesced loads through LDS), thread = (row tid & 31, word tid >> 5), so the 32
// lanes of a half-wave store 32 consecutive rows of one word plane: 128-byte runs.  D % 32 == 0, x 16-byte aligned.
#define ENC_WM_PITCH 260
template <int EM>
__global__ __launch_bounds__(256) void k_encode_rows_wm(const float* __restrict__ x, int R, int D, int T, NeuronP p, const EncTh eth,
                                                        uint32_t* __restrict__ planes, size_t plane_stride) {
    constexpr bool ZR = EM != ENC_GENERIC;
    __shared__ __attribute__((aligned(16))) float tile[32 * ENC_WM_PITCH];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * 32, w0 = blockIdx.x * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = tid + 256 * j;                            // 16-byte piece: row q / 64, floats 4 (q % 64) .. of the 256-float run
        const int row = q >> 6, col = w0 * 32 + (q & 63) * 4;
        f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < R && col < D) v4 = *reinterpret_cast<const f32x4*>(x + (size_t)(r0 + row) * D + col);
        *reinterpret_cast<f32x4*>(tile + row * ENC_WM_PITCH + (q & 63) * 4) = v4;
    }
    __syncthreads();
    const int row = tid & 31, wd = tid >> 5;
    if (r0 + row >= R || (w0 + wd) * 32 >= D) return;
    float xv[32], v[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(tile + row * ENC_WM_PITCH + wd * 32 + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) { xv[4 * q + r] = t4[r]; v[4 * q + r] = 0.0f; }      // v = 0: faster_rcnn.py:484
    }
    uint32_t* dst = planes + (size_t)(w0 + wd) * R + r0 + row;
    uint32_t prev = 0;
    for (int t = 0; t < T; ++t) {
        uint32_t word = 0;
        if (EM == ENC_QUANT) {                      // period planes by thresholds (snn_common.h)
            const float th = eth.th[t];
#pragma unroll
            for (int b = 31; b >= 0; --b) enc_quant_word(xv[b], th, word);
            const uint32_t cum = word;
            word = cum & ~prev;
            prev = cum;
        } else {
#pragma unroll
            for (int b = 31; b >= 0; --b) enc_step_word<ZR>(xv[b], v[b], p, word);          // bit 31 first
        }
        dst[(size_t)t * plane_stride] = word;
    }
}

// K1d (round 5): the detector's encoder for the structured-sparse fc6 in ONE launch - period planes by thresholds, written straight in
// fc6's reduction order k' = bin * C + channel (what k_permute_planes made of the reference-order planes) and, for the planes e_3 ..,
// COMPRESSED (what k_compress_planes made of those): three launches and two HBM round trips of the planes become one.
// x [R][C * S] fp32 in the reference's flatten order k = c * S + bin (faster_rcnn.py:473).  Block = ENCP_RB RoIs x 64 channels (two channel
// blocks cb = 2 cp, 2 cp + 1: a compressed step is a pair of words (bin, cb), (bin, cb + 1)).  Encode: a wave takes (RoI, channel block)
// tasks, lane = bin (S of 64 lanes), the lane's 32 channels are 32 loads at stride S floats - every load instruction reads one S-float run of
// the RoI's row - and the T words go to LDS [t][cb][RoI][bin] (odd pitch S: conflict-free both ways).  Store: thread = (RoI, item): the dense planes' words as they are, a sparse
// plane's pair through sp_compress_pair - runs of ENCP_RB consecutive RoIs of one word plane / array.
#define ENCP_LDS_WORDS 19200                        // planes per pass through LDS: 19200 / (2 S RB) - 12 at RB = 16, 24 at RB = 8 (75 KB); longer windows take more passes
template <int S, int RB, int NW>                    // NW waves per block (4 or 8: 2 RB tasks over NW waves, two at a time)
__global__ __launch_bounds__(64 * NW) void k_encode_rows_perm(const float* __restrict__ x, int R, int C, int T, int nd, const EncTh eth,
                                                          uint32_t* __restrict__ planes, uint32_t* __restrict__ cmp) {
    static_assert(S <= 64, "one lane per bin");
    extern __shared__ uint32_t pw[];                          // [min(T, TMAX)][2][RB][S]
    constexpr int TMAX = ENCP_LDS_WORDS / (2 * S * RB);
    static_assert((2 * RB) % (2 * NW) == 0, "tasks two at a time per wave");
    __shared__ uint16_t code[256];
    if (threadIdx.x < 256) code[threadIdx.x] = sp_byte_code(threadIdx.x);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * RB, cp = blockIdx.y, cbn = C / 32, D = C * S, Dw = D / 32;
    const size_t plane_words = (size_t)Dw * R, cmp_plane = (size_t)(Dw / 2) * SP_A_ARR * R;
    for (int t0 = 0; t0 < T; t0 += TMAX) {
        const int tn = min(TMAX, T - t0);
        if (t0) __syncthreads();
        // ---- encode: tasks (RoI, cb) over the waves; the next task's 32 loads are in flight while this one's words are formed
        auto load_task = [&](float (&xv)[32], const int task) __attribute__((always_inline)) {
            const int row = r0 + (task >> 1), cb = task & 1;
            const float* src = x + (size_t)min(row, R - 1) * D + (size_t)((2 * cp + cb) * 32) * S + min(lane, S - 1);
#ifdef SNN_EXP_ENCP_NOLOAD                             // (timing experiments - wrong results: which phase bounds the launch?)
#pragma unroll
            for (int j = 0; j < 32; ++j) xv[j] = (float)(j + lane) * 0.01f + (float)((size_t)src & 4);
#else
#pragma unroll
            for (int j = 0; j < 32; ++j) xv[j] = src[j * S];
#endif
        };
        auto encode_task = [&](const float (&xv)[32], const int task) __attribute__((always_inline)) {
            const int rl = task >> 1, cb = task & 1;
            if (lane < S) {
                uint32_t prev = 0;
                for (int t = 0; t < t0 + tn; ++t) {          // (cumulative words from step 0: a later pass re-derives what it needs)
                    uint32_t word = 0;
                    const float th = eth.th[t];
#ifdef SNN_EXP_ENCP_NOENC
                    word = __float_as_uint(xv[t & 31] + xv[(t + 7) & 31]) & (th > 0.0f ? 0x11111111u : 0u);
#else
#pragma unroll
                    for (int j = 31; j >= 0; --j) enc_quant_word(xv[j], th, word);
#endif
                    const uint32_t cum = word;
                    word = cum & ~prev;
                    prev = cum;
                    if (t >= t0) pw[(((t - t0) * 2 + cb) * RB + rl) * S + lane] = word;
                }
            }
        };
        float xa[32], xb[32];
        load_task(xa, wave);
#pragma unroll 1
        for (int task = wave; task < 2 * RB; task += 2 * NW) {     // (2 RB / NW tasks per wave: even)
            load_task(xb, task + NW);
            encode_task(xa, task);
            if (task + 2 * NW < 2 * RB) load_task(xa, task + 2 * NW);
            encode_task(xb, task + NW);
        }
        __syncthreads();
        // ---- store: thread = (RoI tid % RB, item tid / RB)
        const int rl = tid & (RB - 1), row = r0 + rl;                    // (RB = 8 or 16)
#ifdef SNN_EXP_ENCP_NOSTORE
        if (row < R && pw[tid] == 0x12345678u) {
#else
        if (row < R) {
#endif
            for (int t = t0; t < t0 + tn; ++t) {
                const uint32_t* pt = pw + (size_t)(t - t0) * 2 * S * RB;
                if (t < nd || !cmp) {                         // raw words (bin, cb) -> word plane bin * cbn + 2 cp + cb
                    for (int it = tid / RB; it < 2 * S; it += 64 * NW / RB) {
                        const int cb = it / S, bin = it % S;
                        planes[(size_t)t * plane_words + (size_t)(bin * cbn + 2 * cp + cb) * R + row] = pt[(cb * RB + rl) * S + bin];
                    }
                } else {                                      // compressed step (bin, cp): pair index (bin * cbn + 2 cp) / 2
                    for (int bin = tid / RB; bin < S; bin += 64 * NW / RB) {
                        uint32_t c4[4];
                        sp_compress_pair(pt[rl * S + bin], pt[(RB + rl) * S + bin], code, c4);
                        uint32_t* o = cmp + (size_t)(t - nd) * cmp_plane + (size_t)(bin * (cbn / 2) + cp) * SP_A_ARR * R + row;
#pragma unroll
                        for (int j = 0; j < SP_A_ARR; ++j) o[(size_t)j * R] = c4[j];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1c: MultiScaleRoIAlign (7x7, sampling_ratio 2, aligned=False) fused with the detector's constant-current
// encoder (roi_heads.py:1217 -> faster_rcnn.py:473,494): the [R,C,7,7] fp32 RoI features (100 MB at R=2000) are
// never materialised; each thread pools one (RoI, channel, bin) element - 4 samples x 4 bilinear taps - runs its
// T encoder steps in registers and the wave ballots straight into the bit-planes [T][R][Dw] (flatten order
// d = c*49 + ph*7 + pw).  Arithmetic follows torchvision 0.13.1's CPU roi_align kernel (restated in oracle/roi_align_oracle.py) op for op
// (explicit roundings, no fma): sample = (hy*hx)*v1 + (hy*lx)*v2 + (ly*hx)*v3 + (ly*lx)*v4, bin = mean of 4.
// ------------------------------------------------------------------------------------------------
struct RoiLevel { const float* feat; int H, W; float scale; };
struct RoiArgs {
    RoiLevel lv[4];
    const float* rois;        // [R][4] x1,y1,x2,y2 in image coordinates
    const int* roi_batch;     // [R]
    const int* roi_level;     // [R] index into lv
    float* pooled;            // nullable test hook: [R][C*49] fp32
    uint32_t* planes;
    unsigned long long plane_stride;
    int R, C, T, Dw;
    int quant;                // period planes by thresholds (eth valid): snn_common.h
    int E, n_rg, RW;          // k_roi_align_encode_tab: groups of 64 elements per work-group, RoI groups, RoIs per wave (4 waves)
    uint32_t* cmp;            // k_roi_align_encode_perm: compressed planes e_(nd+1) .. (csrc/snn_sparse.h), [T - nd][Dw / 2][4][R]
    int nd;                   // ... the first nd planes leave raw (fc6's dense planes)
    NeuronP p;
    EncTh eth;
};

__device__ __forceinline__ float roi_bilinear(const float* __restrict__ f, int H, int W, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
    y = fmaxf(y, 0.0f);
    x = fmaxf(x, 0.0f);
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = __fsub_rn(y, (float)y_low), lx = __fsub_rn(x, (float)x_low);
    const float hy = __fsub_rn(1.0f, ly), hx = __fsub_rn(1.0f, lx);
    const float v1 = f[y_low * W + x_low], v2 = f[y_low * W + x_high];
    const float v3 = f[y_high * W + x_low], v4 = f[y_high * W + x_high];
    float acc = __fmul_rn(__fmul_rn(hy, hx), v1);
    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(hy, lx), v2));
    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ly, hx), v3));
    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ly, lx), v4));
    return acc;
}

// pooled value of element d = c*49 + ph*7 + pw of RoI r
__device__ __forceinline__ float roi_pool_element(const RoiArgs& a, int r, int d) {
    const int D = a.C * 49;
    float val = 0.0f;
    {
        const int c = d / 49, ph = (d % 49) / 7, pw = d % 7;
        const RoiLevel L = a.lv[a.roi_level[r]];
        const float* roi = a.rois + (size_t)r * 4;
        const float x1 = __fmul_rn(roi[0], L.scale), y1 = __fmul_rn(roi[1], L.scale);
        const float rw = fmaxf(__fsub_rn(__fmul_rn(roi[2], L.scale), x1), 1.0f);
        const float rh = fmaxf(__fsub_rn(__fmul_rn(roi[3], L.scale), y1), 1.0f);
        const float bh = __fdiv_rn(rh, 7.0f), bw = __fdiv_rn(rw, 7.0f);
        const float* f = L.feat + ((size_t)a.roi_batch[r] * a.C + c) * (size_t)(L.H * L.W);
        // sample coordinate, in torchvision's own operation order (roi_align_common.h, pre_calc_for_bilinear_interpolate):
        //   yy = roi_start_h + ph * bin_size_h + (iy + .5f) * bin_size_h / roi_bin_grid_h     (left to right, fp32, no fma)
        const float by = __fadd_rn(y1, __fmul_rn((float)ph, bh)), bx = __fadd_rn(x1, __fmul_rn((float)pw, bw));
        float s[2][2];
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                const float yy = __fadd_rn(by, __fdiv_rn(__fmul_rn((float)iy + 0.5f, bh), 2.0f));
                const float xx = __fadd_rn(bx, __fdiv_rn(__fmul_rn((float)ix + 0.5f, bw), 2.0f));
                s[iy][ix] = roi_bilinear(f, L.H, L.W, yy, xx);
            }
        // mean over the 2x2 samples (roi_align_kernel.cpp: output_val += sample in (iy, ix) order, then /= count)
        val = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(s[0][0], s[0][1]), s[1][0]), s[1][1]), 4.0f);
        if (a.pooled) a.pooled[(size_t)r * D + d] = val;
    }
    return val;
}

__global__ __launch_bounds__(256) void k_roi_align_encode(const RoiArgs a) {
    const int r = blockIdx.y;
    const int D = a.C * 49;
    const int d = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const float val = d < D ? roi_pool_element(a, r, d) : 0.0f;
    // encoder over T steps; the wave's 64 consecutive d are two plane words
    float v = 0.0f;
    const size_t e = (size_t)r * a.Dw * 32 + d;
    const bool in = d < a.Dw * 32;
    for (int t = 0; t < a.T; ++t) {
        bool zz;
        if (a.quant) {
            zz = val >= a.eth.th[t] && !(t > 0 && val >= a.eth.th[t - 1]);
        } else {
            zz = enc_step(val, v, a.p);
            v = (zz && a.p.v_fire != 0.0f) ? a.p.v_fire : v;     // period planes (zero rest / reset potentials only): never fires again
        }
        const bool z = zz && d < D;
        const unsigned long long m = __ballot(z);
        if ((lane & 31) == 0 && in)
            a.planes[(size_t)t * a.plane_stride + (e >> 5)] = (lane < 32) ? (uint32_t)m : (uint32_t)(m >> 32);
    }
}

// K1c': the same with WORD-MAJOR planes [T][Dw][R]: a work-group takes 32 RoIs x 64 elements (two plane words); wave w pools and
// encodes RoIs 8w .. 8w+7 one after the other (lanes = the 64 elements, the ballot of a step is the RoI's word pair), the words
// meet in LDS and leave as 128-byte runs of 32 consecutive RoIs per (t, word).  RoI parameters are wave-uniform (scalar loads).
__global__ __launch_bounds__(256) void k_roi_align_encode_wm(const RoiArgs a) {
    __shared__ uint32_t wbuf[SNN_MAX_STEPS * 2 * 32];            // [t][word][RoI]
    const int D = a.C * 49;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.y * 32, d = blockIdx.x * 64 + lane;
    for (int i = 0; i < 8; ++i) {
        const int rl = wave * 8 + i, r = r0 + rl;                // wave-uniform
        if (r >= a.R) break;
        const float val = d < D ? roi_pool_element(a, r, d) : 0.0f;
        float v = 0.0f;
        for (int t = 0; t < a.T; ++t) {
            bool zz;
            if (a.quant) {
                zz = val >= a.eth.th[t] && !(t > 0 && val >= a.eth.th[t - 1]);
            } else {
                zz = enc_step(val, v, a.p);
                v = (zz && a.p.v_fire != 0.0f) ? a.p.v_fire : v; // period planes
            }
            const bool z = zz && d < D;
            const unsigned long long m = __ballot(z);
            if (lane == 0) { wbuf[(t * 2 + 0) * 32 + rl] = (uint32_t)m; wbuf[(t * 2 + 1) * 32 + rl] = (uint32_t)(m >> 32); }
        }
    }
    __syncthreads();
    const int w0 = blockIdx.x * 2;
    for (int idx = threadIdx.x; idx < a.T * 64; idx += 256) {
        const int rl = idx & 31, wd = (idx >> 5) & 1, t = idx >> 6;
        if (r0 + rl < a.R && w0 + wd < a.Dw)
            a.planes[(size_t)t * a.plane_stride + (size_t)(w0 + wd) * a.R + r0 + rl] = wbuf[(t * 2 + wd) * 32 + rl];
    }
}

// K1c'': K1c' with the sample geometry of a RoI computed ONCE per wave instead of once per element.  Everything about a
// bilinear sample except the feature values is separable and the same for all C channels: 14 sample rows (7 bins x 2) with
// {y_low * W, y_high * W, ly, hy} and 14 sample columns with {pair base, clamped, lx, hx} (l = -1 outside the map) - 28 x 16 bytes of LDS per wave,
// built by 28 lanes with exactly the operations of roi_pool_element / roi_bilinear (same roundings, same order), then looked
// up by bin.  K1c' spent ~200 vector instructions per element on coordinates (six divisions among them) and issued 16
// 4-byte gathers; here an element costs ~70 instructions and 8 eight-byte gathers: the two taps of a sample row are
// neighbours (x_high = x_low + 1) except when the column is clamped to the last one, where the pair is read one to the left
// and both taps take its second half.  A work-group covers 4 RW RoIs (RW per wave, one after the other) x E groups of 64 elements
// (E x 2 plane words per step; the table is reused E times); the words leave through LDS as runs of 4 RW consecutive RoIs.
// The kernel is bound by its gathers: ~28 L1 accesses per load instruction (7 bin rows x the row's lines, two channels per
// wave), and with the plain block order by L2 misses as well - 1.2 GB fetched for 200 MB of feature maps, since every XCD
// saw every channel plane.  The XCD-aware, channel-major block order below brings that to 0.34 GB (77 % L2 hits).
// Needs W >= 2 on every level and C*H*W < 2^29 elements per image (32-bit element offsets); the launcher checks.
struct __attribute__((aligned(16))) RoiTabEntry { int a, b; float l, h; };
__global__ __launch_bounds__(256) void k_roi_align_encode_tab(const RoiArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t wbuf_dyn[];          // [t][2E words][4 RW RoIs]
    __shared__ RoiTabEntry tab[4][32];                                           // per wave: 0..13 rows, 16..29 columns
    const int D = a.C * 49, E = a.E;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // work-group -> (element block, RoI group), XCD-aware and channel-major in time: work-group b runs on XCD b % 8, so XCD x takes
    // the element blocks x, x + 8, ... and for each of them ALL RoI groups, one block after the other - at any moment an XCD's
    // work-groups read a handful of channel planes (which its 4-MB L2 holds) instead of all of them (a.n_rg RoI groups)
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int eblk = (jj / a.n_rg) * 8 + xcd, rgrp = jj % a.n_rg;
    if (eblk * E * 2 >= a.Dw) return;
    const int RW = a.RW, RG = 4 * RW;                          // RoIs per wave / per work-group
    const int r0 = rgrp * RG;
    const int d0 = eblk * E * 64;
    RoiTabEntry* const tb = tab[wave];
    for (int i = 0; i < RW; ++i) {
        const int rl = wave * RW + i, r = r0 + rl;               // wave-uniform
        if (r >= a.R) break;
        const RoiLevel L = a.lv[a.roi_level[r]];
        const float* roi = a.rois + (size_t)r * 4;
        const float* const fimg = L.feat + (size_t)a.roi_batch[r] * a.C * (size_t)(L.H * L.W);
        const int HW = L.H * L.W;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                   // the previous RoI's lookups are done
        __builtin_amdgcn_wave_barrier();
        if (lane < 32 && (lane & 15) < 14) {
            // lane 0..13: sample row s = 2 ph + iy;  lane 16..29: sample column s = 2 pw + ix
            const bool is_x = lane >= 16;
            const int sidx = lane & 15, pb = sidx >> 1, ii = sidx & 1;
            const float lo = __fmul_rn(is_x ? roi[0] : roi[1], L.scale);
            const float ext = fmaxf(__fsub_rn(__fmul_rn(is_x ? roi[2] : roi[3], L.scale), lo), 1.0f);
            const float bin = __fdiv_rn(ext, 7.0f);
            const float b0 = __fadd_rn(lo, __fmul_rn((float)pb, bin));
            float y = __fadd_rn(b0, __fdiv_rn(__fmul_rn((float)ii + 0.5f, bin), 2.0f));
            const int n = is_x ? L.W : L.H;
            RoiTabEntry e;
            if (y < -1.0f || y > (float)n) { e.a = 0; e.b = 0; e.l = -1.0f; e.h = 0.0f; }     // outside the map: l < 0, offsets stay valid
            else {
                y = fmaxf(y, 0.0f);
                int y_low = (int)y, y_high;
                if (y_low >= n - 1) { y_high = y_low = n - 1; y = (float)y_low; } else y_high = y_low + 1;
                e.l = __fsub_rn(y, (float)y_low);
                e.h = __fsub_rn(1.0f, e.l);
                if (is_x) { e.b = y_high == y_low; e.a = e.b ? y_low - 1 : y_low; }      // pair base, clamped
                else { e.a = y_low * L.W; e.b = y_high * L.W; }
            }
            tb[lane] = e;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int eg = 0; eg < E; ++eg) {
            const int d = d0 + eg * 64 + lane;
            float val = 0.0f;
            if (d < D) {
                const int c = d / 49, bin = d - 49 * c, ph = bin / 7, pw = bin - 7 * ph;
                const float* const f = fimg + (unsigned)(c * HW);
                typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                RoiTabEntry ye[2], xe[2];
                ye[0] = tb[2 * ph]; ye[1] = tb[2 * ph + 1]; xe[0] = tb[16 + 2 * pw]; xe[1] = tb[16 + 2 * pw + 1];
                // all eight tap pairs are requested before any is used (samples outside the map read row / column 0 and are dropped)
                f32x2u pt[2][2], qt[2][2];
#pragma unroll
                for (int iy = 0; iy < 2; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix) {
                        pt[iy][ix] = *reinterpret_cast<const f32x2u*>(f + (ye[iy].a + xe[ix].a));
                        qt[iy][ix] = *reinterpret_cast<const f32x2u*>(f + (ye[iy].b + xe[ix].a));
                    }
                float s[2][2];
#pragma unroll
                for (int iy = 0; iy < 2; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix) {
                        const bool ok = ye[iy].l >= 0.0f && xe[ix].l >= 0.0f;
                        const f32x2u p2 = pt[iy][ix], q2 = qt[iy][ix];
                        const float v1 = xe[ix].b ? p2.y : p2.x, v2 = p2.y, v3 = xe[ix].b ? q2.y : q2.x, v4 = q2.y;
                        float acc = __fmul_rn(__fmul_rn(ye[iy].h, xe[ix].h), v1);
                        acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ye[iy].h, xe[ix].l), v2));
                        acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ye[iy].l, xe[ix].h), v3));
                        acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ye[iy].l, xe[ix].l), v4));
                        s[iy][ix] = ok ? acc : 0.0f;
                    }
                val = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(s[0][0], s[0][1]), s[1][0]), s[1][1]), 4.0f);
                if (a.pooled) a.pooled[(size_t)r * D + d] = val;
            }
            float v = 0.0f;
            unsigned long long prev = 0ull;
            for (int t = 0; t < a.T; ++t) {
                unsigned long long m;
                if (a.quant) {
                    const unsigned long long cum = __ballot(d < D && val >= a.eth.th[t]);     // first spike at or before t
                    m = cum & ~prev;
                    prev = cum;
                } else {
                    bool zz = enc_step(val, v, a.p);
                    v = (zz && a.p.v_fire != 0.0f) ? a.p.v_fire : v;                           // period planes
                    m = __ballot(zz && d < D);
                }
                if (lane == 0) {
                    wbuf_dyn[((t * E + eg) * 2 + 0) * RG + rl] = (uint32_t)m;
                    wbuf_dyn[((t * E + eg) * 2 + 1) * RG + rl] = (uint32_t)(m >> 32);
                }
            }
        }
    }
    __syncthreads();
    const int w0 = eblk * E * 2;
    for (int idx = threadIdx.x; idx < a.T * 2 * E * RG; idx += 256) {
        const int rl = idx % RG, wd = (idx / RG) % (2 * E), t = idx / (2 * E * RG);
        if (r0 + rl < a.R && w0 + wd < a.Dw)
            a.planes[(size_t)t * a.plane_stride + (size_t)(w0 + wd) * a.R + r0 + rl] = wbuf_dyn[(t * 2 * E + wd) * RG + rl];
    }
}


// K1e (round 6): K1c'' writing what the structured-sparse fc6 reads - period planes in fc6's reduction order k' = bin * C + channel, the planes
// e_3 .. COMPRESSED - so that the default product path (RoIHeadsSNN.fuse_roi_align) needs neither k_permute_planes nor k_compress_planes nor
// their two 31-MB plane copies (row f1's fold was only on the stand-alone head; roi_heads.py:1217 -> faster_rcnn.py:494).
// A word of a permuted plane = 32 CHANNELS at one bin, so the lanes of a wave are (bin column pw = lane >> 3 < 7, channel lane & 7) of ONE row of
// bins ph: the (8 x 8 transposed) ballot of a step holds, in byte pw, the eight channels of an octet at bin (ph, pw), and four octets make the word.  Work-group =
// (4 RW RoIs, 64 channels = one compressed step per bin, one row of seven bins): wave w takes RW RoIs, for each of the 8 octets one task per RoI
// (octet-major: the work-groups of an XCD sweep the channels in step, as in K1c''); the bytes meet in LDS as [t][pw][RoI][octet], i.e. the word
// pair of (t, bin, RoI) is 8 consecutive bytes; store = thread per (t, pw, RoI): the dense planes' two words as they are, a sparse plane's
// pair through sp_compress_pair - runs of 4 RW consecutive RoIs, exactly what k_encode_rows_perm writes.  The sample geometry comes from the
// same per-RoI table (built with the operations of roi_bilinear, op for op), the eight tap pairs of an element from the same offsets: pooled
// values and planes are bit-identical to K1c'' + k_permute_planes + k_compress_planes (tests/test_gpu_roialign.py compares the workspace bytes).
// A load instruction touches 8 channel planes x one or two lines of ONE feature row (K1c'': ~28 lines - 7 bin rows of 1.3 channels).
// Threshold form of the encoder only (a.quant: the launcher falls back to the three launches otherwise); C % 64 == 0.
template <int RW>
__global__ __launch_bounds__(256) void k_roi_align_encode_perm(const RoiArgs a) {
    constexpr int RG = 4 * RW;
    extern __shared__ __attribute__((aligned(16))) unsigned char pbuf[];          // word pairs [t][7][RG] (8 bytes each), then the raw ballots [t][RG][8 octets]
    unsigned long long* const mbuf = reinterpret_cast<unsigned long long*>(pbuf + (size_t)a.T * 7 * RG * 8);
    __shared__ RoiTabEntry tab[4][RW][16];                                        // per wave and RoI: 0, 1 = the two sample rows of ph; 2 .. 15 = sample columns
    __shared__ uint16_t code[256];
    code[threadIdx.x] = sp_byte_code(threadIdx.x);
    const int T = a.T;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // work-group -> (channel pair-block cp, RoI group, bin row): XCD b % 8 keeps to ONE cp where the cp count divides 8, and takes the seven bin rows of
    // a RoI group one after the other (neighbouring bin rows read neighbouring feature rows of the same windows)
    const int n_cp = a.C / 64, n_items = a.n_rg * 7;
    int cp, item;
    if (8 % n_cp == 0) {
        const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, per = 8 / n_cp;
        cp = xcd % n_cp;
        item = jj * per + xcd / n_cp;
    } else { cp = blockIdx.x % n_cp; item = blockIdx.x / n_cp; }
    if (item >= n_items) return;
    const int rgrp = item / 7, ph = item - 7 * rgrp;
    const int r0 = rgrp * RG;
    // lane = channel-of-the-octet * 8 + bin column (7 of 8 used): CONSECUTIVE lanes read neighbouring bins of one channel plane - the first form, with the
    // channel fastest, put every lane of a quad on another channel plane and cost 0.42 ms against 0.25 for the table kernel (profiles/r6_roi_fold_ab.txt)
    const int pw = min(lane & 7, 6), ch_l = lane >> 3;
    const bool act = (lane & 7) < 7;
    // ---- sample tables of this wave's RoIs (K1c'': same operations, same order)
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int r = r0 + wave * RW + i;                        // wave-uniform
        if (r >= a.R) break;
        const RoiLevel L = a.lv[a.roi_level[r]];
        const float* roi = a.rois + (size_t)r * 4;
        if (lane < 16) {
            // lane 0, 1: sample row s = 2 ph + lane;  lane 2 .. 15: sample column s = lane - 2 = 2 pw + ix
            const bool is_x = lane >= 2;
            const int sidx = is_x ? lane - 2 : 2 * ph + lane, pb = sidx >> 1, ii = sidx & 1;
            const float lo = __fmul_rn(is_x ? roi[0] : roi[1], L.scale);
            const float ext = fmaxf(__fsub_rn(__fmul_rn(is_x ? roi[2] : roi[3], L.scale), lo), 1.0f);
            const float bin = __fdiv_rn(ext, 7.0f);
            const float b0 = __fadd_rn(lo, __fmul_rn((float)pb, bin));
            float y = __fadd_rn(b0, __fdiv_rn(__fmul_rn((float)ii + 0.5f, bin), 2.0f));
            const int n = is_x ? L.W : L.H;
            RoiTabEntry e;
            if (y < -1.0f || y > (float)n) { e.a = 0; e.b = 0; e.l = -1.0f; e.h = 0.0f; }     // outside the map: l < 0, offsets stay valid
            else {
                y = fmaxf(y, 0.0f);
                int y_low = (int)y, y_high;
                if (y_low >= n - 1) { y_high = y_low = n - 1; y = (float)y_low; } else y_high = y_low + 1;
                e.l = __fsub_rn(y, (float)y_low);
                e.h = __fsub_rn(1.0f, e.l);
                if (is_x) { e.b = y_high == y_low; e.a = e.b ? y_low - 1 : y_low; }      // pair base, clamped
                else { e.a = y_low * L.W; e.b = y_high * L.W; }
            }
            tab[wave][i][lane] = e;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll 1
    for (int o = 0; o < 8; ++o) {
        const int c = cp * 64 + o * 8 + ch_l;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int rl = wave * RW + i, r = r0 + rl;           // wave-uniform
            if (r >= a.R) break;
            const RoiLevel L = a.lv[a.roi_level[r]];
            const int HW = L.H * L.W;
            const float* const f = L.feat + (size_t)a.roi_batch[r] * a.C * (size_t)HW + (unsigned)(c * HW);
            const RoiTabEntry* const tb = tab[wave][i];
            RoiTabEntry ye[2], xe[2];
            ye[0] = tb[0]; ye[1] = tb[1]; xe[0] = tb[2 + 2 * pw]; xe[1] = tb[3 + 2 * pw];
            // all eight tap pairs are requested before any is used (samples outside the map read row / column 0 and are dropped)
            f32x2u pt[2][2], qt[2][2];
#pragma unroll
            for (int iy = 0; iy < 2; ++iy)
#pragma unroll
                for (int ix = 0; ix < 2; ++ix) {
                    pt[iy][ix] = *reinterpret_cast<const f32x2u*>(f + (ye[iy].a + xe[ix].a));
                    qt[iy][ix] = *reinterpret_cast<const f32x2u*>(f + (ye[iy].b + xe[ix].a));
                }
            float sm[2][2];
#pragma unroll
            for (int iy = 0; iy < 2; ++iy)
#pragma unroll
                for (int ix = 0; ix < 2; ++ix) {
                    const bool ok = ye[iy].l >= 0.0f && xe[ix].l >= 0.0f;
                    const f32x2u p2 = pt[iy][ix], q2 = qt[iy][ix];
                    const float v1 = xe[ix].b ? p2.y : p2.x, v2 = p2.y, v3 = xe[ix].b ? q2.y : q2.x, v4 = q2.y;
                    float acc = __fmul_rn(__fmul_rn(ye[iy].h, xe[ix].h), v1);
                    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ye[iy].h, xe[ix].l), v2));
                    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ye[iy].l, xe[ix].h), v3));
                    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ye[iy].l, xe[ix].l), v4));
                    sm[iy][ix] = ok ? acc : 0.0f;
                }
            const float val = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(sm[0][0], sm[0][1]), sm[1][0]), sm[1][1]), 4.0f);
            unsigned long long prev = 0ull;
            unsigned long long* const dst = mbuf + (size_t)rl * 8 + o;
            for (int t = 0; t < T; ++t) {
                const unsigned long long cum = __ballot(act && val >= a.eth.th[t]);           // first spike at or before t
                const unsigned long long m = cum & ~prev;                                      // bit (channel * 8 + bin column): byte c = the bins of channel c
                prev = cum;
                if (lane == 0) dst[(size_t)t * (RG * 8)] = m;
            }
        }
    }
    __syncthreads();
    // ---- bit transposition by ballot: for a (t, RoI) the 64 bytes of mbuf are the 64 channels' bin bits (byte = octet * 8 + channel of the octet); lane = channel
    // reads its byte, and the ballot of bit pw over the lanes IS the word pair (channel blocks 2 cp, 2 cp + 1) of bin column pw.  (The first form
    // transposed every 8 x 8 ballot on the scalar unit inside the task loop - 18 scalar instructions per step and task: 0.34 ms for the launch.)
    for (int it = wave; it < T * RG; it += 4) {
        const unsigned int byte = reinterpret_cast<const unsigned char*>(mbuf)[(size_t)it * 64 + lane];
        const int t = it / RG, rl = it - t * RG;
        unsigned long long wp = 0ull;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const unsigned long long b = __ballot((byte >> q) & 1u);
            wp = lane == q ? b : wp;
        }
        if (lane < 7) *reinterpret_cast<unsigned long long*>(pbuf + ((size_t)(t * 7 + lane) * RG + rl) * 8) = wp;
    }
    __syncthreads();
    // ---- store: item = (t, pw, RoI): the word pair (channel blocks 2 cp, 2 cp + 1) of bin ph * 7 + pw
    const int cbn = a.C / 32;
    const size_t R = (size_t)a.R, cmp_plane = (size_t)(a.Dw / 2) * SP_A_ARR * R;
    for (int idx = threadIdx.x; idx < T * 7 * RG; idx += 256) {
        const int rl = idx % RG, pwi = (idx / RG) % 7, t = idx / (7 * RG);
        const size_t row = (size_t)(r0 + rl);
        if (row >= R) continue;
        const uint2 w2 = *reinterpret_cast<const uint2*>(pbuf + (size_t)idx * 8);
        const int bin = ph * 7 + pwi;
        if (t < a.nd || !a.cmp) {
            uint32_t* o = a.planes + (size_t)t * a.plane_stride + (size_t)(bin * cbn + 2 * cp) * R + row;
            o[0] = w2.x; o[R] = w2.y;
        } else {
            uint32_t c4[4];
            sp_compress_pair(w2.x, w2.y, code, c4);
            uint32_t* o = a.cmp + (size_t)(t - a.nd) * cmp_plane + (size_t)(bin * (cbn / 2) + cp) * SP_A_ARR * R + row;
#pragma unroll
            for (int j = 0; j < SP_A_ARR; ++j) o[(size_t)j * R] = c4[j];
        }
    }
}
