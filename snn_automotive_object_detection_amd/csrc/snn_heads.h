// LIF scan (K4), the leaky-integrator heads in their three forms (K5, K5b, K5c) and the spike counters of the
// spike-rate outputs.  Included by snn_kernels.hip (after snn_bf16x3.h: table and swizzle helpers).
#pragma once

// ------------------------------------------------------------------------------------------------
// K4: LIF scan over T of currents cur[T][R][ldc] -> spike planes [T][R][Nw]  (+ per-row counts)
// ------------------------------------------------------------------------------------------------
// cur holds the steps t0 .. t0+Tc-1 only (cur[t - t0][R][ldc]); the other steps integrate +0 (dead time steps, Gemm3Args.Tc)
__global__ __launch_bounds__(256) void k_lif_scan(const float* __restrict__ cur, int T, int t0, int Tc, int R, int N, int Nw,
                                                  int ldc, NeuronP p, uint32_t* __restrict__ spk,
                                                  size_t spk_stride, uint32_t* __restrict__ row_counts) {
    const size_t Np = (size_t)Nw * 32;
    const size_t total = (size_t)R * Np;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = e < total;
    const size_t r = in ? e / Np : 0;
    const int n = in ? (int)(e % Np) : 0;
    const bool live = in && n < N;
    const float* c = cur + r * ldc + n;
    const size_t tstride = (size_t)R * ldc;
    float v = p.v_leak, i = 0.0f;
    const int lane = threadIdx.x & 63;
    uint32_t cnt = 0;
    for (int t = 0; t < T; ++t) {
        const float x = (live && t >= t0 && t < t0 + Tc) ? c[(size_t)(t - t0) * tstride] : 0.0f;
        const bool z = lif_step(x, v, i, p) && live;
        const unsigned long long m = __ballot(z);
        if ((lane & 31) == 0 && in) {
            const uint32_t w = (lane < 32) ? (uint32_t)m : (uint32_t)(m >> 32);
            spk[(size_t)t * spk_stride + (e >> 5)] = w;
            cnt += __popc(w);
        }
    }
    if (row_counts != nullptr && (lane & 31) == 0 && in && cnt != 0) atomicAdd(&row_counts[r], cnt);
}

// ------------------------------------------------------------------------------------------------
// K5: both leaky-integrator heads, time-collapsed.  The LI cell and the bias-free 1x1 conv / linear
// in front of it are linear and only the last membrane is used (rpn.py:118-119,
// faster_rcnn.py:513-514), so  mem_T = W . (sum_t kappa_last[t] * spk_t)  and the spike-rate
// variant's sum over t of the membranes is  W . (sum_t kappa_sum[t] * spk_t).
// ------------------------------------------------------------------------------------------------
struct Kappa { float last[SNN_MAX_STEPS]; float sum[SNN_MAX_STEPS]; };

// Block = 256 threads = RB rows; the reduction runs in slabs of HEADS_KS channels:
//   phase 1  thread = (row, 32-channel word): S[row][k] = sum_t kappa[t] * bit_t(row, k)  -> LDS (fp32)
//   phase 2  thread = (row, group of 4 outputs): acc4 += S[row][k] * Wt[k][4jg..4jg+3]   (Wt slab in LDS)
// NOp (outputs rounded up to 16) * RB / 4 <= 256.
#define HEADS_KS 128
template <int RB>
__global__ __launch_bounds__(256) void k_li_heads(const uint32_t* __restrict__ spk, size_t spk_stride, int T,
                                                  int M, int Kw, const float* __restrict__ wT, int NOp, int ldw, int col0, int NA,
                                                  int NB, const Kappa kap, float* __restrict__ out_a,
                                                  float* __restrict__ out_b, float* __restrict__ sum_a,
                                                  float* __restrict__ sum_b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SST = HEADS_KS + 4;                          // padded row of S
    float* S_last = reinterpret_cast<float*>(smem);            // [RB][SST]
    const bool want_sum = (sum_a != nullptr);
    float* S_sum = S_last + RB * SST;                          // [RB][SST], present only if sums are requested
    float* Wl = S_last + (want_sum ? 2 : 1) * RB * SST;        // [HEADS_KS][NOp]
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * RB;
    const int JG = NOp / 4;                                    // output groups per row
    const int prow = tid / JG, pjg = tid % JG;                 // phase-2 role
    const bool pact = prow < RB;
    f32x4 acc_l = {0.f, 0.f, 0.f, 0.f}, acc_s = {0.f, 0.f, 0.f, 0.f};
    const int n_slabs = (Kw * 32 + HEADS_KS - 1) / HEADS_KS;
    for (int sl = 0; sl < n_slabs; ++sl) {
        const int w0 = sl * (HEADS_KS / 32);                   // first channel word of the slab
        const int nw = min(HEADS_KS / 32, Kw - w0);
        // weights of the slab -> LDS (coalesced float4 copy; rows beyond Kp are never touched)
        {
            // (this launch's NOp columns start at column col0 of the packed matrix, whose rows hold ldw columns: both multiples of 16)
            f32x4* dst = reinterpret_cast<f32x4*>(Wl);
            const int q4 = NOp / 4;
            for (int i = tid; i < nw * 32 * q4; i += 256)
                dst[i] = *reinterpret_cast<const f32x4*>(wT + (size_t)(w0 * 32 + i / q4) * ldw + col0 + 4 * (i % q4));
        }
        // phase 1: item = (row, channel word, byte of the word) - 8 channels each, so that small row blocks (detector
        // heads: RB = 16) still give every thread an item
        for (int item = tid; item < RB * (HEADS_KS / 32) * 4; item += 256) {
            const int q = item & 3, wi = (item >> 2) % (HEADS_KS / 32), row = (item >> 2) / (HEADS_KS / 32);
            const int m = m0 + row;
            float sl_[8], ss_[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) { sl_[b] = 0.f; ss_[b] = 0.f; }
            if (m < M && wi < nw) {
                const uint32_t* wp = spk + (size_t)m * Kw + w0 + wi;
#pragma unroll 4
                for (int t = 0; t < T; ++t) {
                    const uint32_t w = wp[(size_t)t * spk_stride] >> (8 * q);
                    const float kl = kap.last[t], ks = kap.sum[t];
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const float bit = (float)((w >> b) & 1u);
                        sl_[b] = fmaf(bit, kl, sl_[b]);            // exact: bit is 0 or 1
                        if (want_sum) ss_[b] = fmaf(bit, ks, ss_[b]);
                    }
                }
            }
            f32x4* dl = reinterpret_cast<f32x4*>(S_last + row * SST + wi * 32 + q * 8);
            f32x4* ds = reinterpret_cast<f32x4*>(S_sum + row * SST + wi * 32 + q * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                dl[j] = f32x4{sl_[4 * j], sl_[4 * j + 1], sl_[4 * j + 2], sl_[4 * j + 3]};
                if (want_sum) ds[j] = f32x4{ss_[4 * j], ss_[4 * j + 1], ss_[4 * j + 2], ss_[4 * j + 3]};
            }
        }
        __syncthreads();
        // phase 2
        if (pact) {
            const float* sl_row = S_last + prow * SST;
            const float* ss_row = S_sum + prow * SST;
            const float* wcol = Wl + 4 * pjg;
            const int kn = nw * 32;
#pragma unroll 4
            for (int k = 0; k < kn; ++k) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wcol + (size_t)k * NOp);
                const float a = sl_row[k];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_l[r] = fmaf(a, w[r], acc_l[r]);
                if (want_sum) {
                    const float c = ss_row[k];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc_s[r] = fmaf(c, w[r], acc_s[r]);
                }
            }
        }
        __syncthreads();
    }
    const int m = m0 + prow;
    if (pact && m < M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = col0 + 4 * pjg + r;
            if (j < NA) { out_a[(size_t)m * NA + j] = acc_l[r]; if (want_sum) sum_a[(size_t)m * NA + j] = acc_s[r]; }
            else if (j < NA + NB) {
                out_b[(size_t)m * NB + (j - NA)] = acc_l[r];
                if (want_sum) sum_b[(size_t)m * NB + (j - NA)] = acc_s[r];
            }
        }
    }
}

// K5b: the same heads on the bf16 matrix cores.  mem_T = sum_t kappa[t] * (spk_t . W): each  spk_t . W  is an exact
// bf16x3 spike GEMM (spikes {0,1}, W = hi + mid + lo, fp32 accumulate) with all NOp <= 64 outputs in 1-4 MFMA column
// tiles, and the kappa-weighted sum over t is an fma chain in the epilogue.  Wave = 16 rows x all T steps; the A
// fragments come from the byte -> 8 bf16 table (as in k_gemm_bf16x3), the weights are split into their three bf16
// planes while they are staged into LDS (no second packed copy): resident when all of W fits, else streamed per
// 32-deep chunk (double-buffered).  Time steps go in groups of 8 (8 x NT accumulators).
// RPN heads (196k rows, K=256, 15 outputs): 137 -> ~40 us;  detector heads (2000 rows, K=1024, 45 outputs): 88 -> ~15 us.
#define LIH_TG 8
struct LiHeadsArgs {
    const uint32_t* spk; unsigned long long spk_stride;
    const float* wT;              // [Kp][ldw] fp32 (snn_pack_heads_weight)
    float *out_a, *out_b, *sum_a, *sum_b;
    int T, M, Kw, NOp, NA, NB, n_groups, resident;
    int ldw, col0;                // this launch: columns col0 .. col0 + NOp - 1 of the packed matrix, whose rows hold ldw columns
    int half_split;               // k_li_heads_mfma, Kw = 8: planes in blocks of four words [T][2][M][4] (Gemm3Args.out_split)
    Kappa kap;
};

template <int NT>
__global__ __launch_bounds__(256) void k_li_heads_mfma(const LiHeadsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lut = smem;                           // 4 KB
    unsigned char* const bbase = smem + G3_LUT_BYTES;          // [chunk slot][3][NOp][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4, lg8 = 8 * lg;
    const int NOp = a.NOp, Kc = a.Kw;
    const uint32_t slot_bytes = 3u * NOp * 64u;
    {
        uint4 q;
        q.x = bf16_pair(tid, 0); q.y = bf16_pair(tid, 1); q.z = bf16_pair(tid, 2); q.w = bf16_pair(tid, 3);
        *reinterpret_cast<uint4*>(lut + tid * 16) = q;
    }
    auto stage = [&](int kc, int slot) {                       // split W[32kc .. 32kc+31][NOp] into 3 bf16 planes
        unsigned char* dst = bbase + (size_t)slot * slot_bytes;
        for (int item = tid; item < 16 * NOp; item += 256) {
            const int n = item % NOp, kp = item / NOp;
            const float* src = a.wT + (size_t)(32 * kc + 2 * kp) * a.ldw + a.col0 + n;
            uint32_t pl[3] = {0u, 0u, 0u};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float w = src[(size_t)h * a.ldw];
                const uint16_t hi = f2bf_rn(w);
                const float r1 = __fsub_rn(w, bf2f(hi));
                const uint16_t mid = f2bf_rn(r1);
                const uint16_t lo = f2bf_rn(__fsub_rn(r1, bf2f(mid)));
                pl[0] |= (uint32_t)hi << (16 * h); pl[1] |= (uint32_t)mid << (16 * h); pl[2] |= (uint32_t)lo << (16 * h);
            }
            const int off = n * 64 + ((((kp >> 2) ^ G3_SWZ(n)) << 4) | ((kp & 3) << 2));
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<uint32_t*>(dst + q * NOp * 64 + off) = pl[q];
        }
    };
    if (a.resident)
        for (int kc = 0; kc < Kc; ++kc) stage(kc, kc);
    __syncthreads();
    const unsigned char* const b_rd = bbase + lr * 64 + ((lg ^ G3_SWZ(lr)) << 4);
    for (int g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
        const int m0 = (g * 4 + wave) * 16;
        const int mrow = min(m0 + lr, a.M - 1);                 // rows past M: recomputed, never stored
        const uint32_t* wsrc = a.spk + (size_t)mrow * (a.half_split ? 4 : a.Kw);
        f32x4 o_last[NT], o_sum[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { o_last[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; o_sum[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // One group of up to LIH_TG time steps.  TN (the steps whose accumulators exist) is a COMPILE-TIME constant: 8, or 4 / 2 / 1 for a
        // shorter last group; steps past the group's real count tn < TN multiply all-zero spike words (exact: their sums stay 0).
        // (Until round 4 the loops tested t < tn at run time; the compiler specialised the unrolled chunks on tn itself, and its tn == 2
        // variant of the Kc == 8 path returned wrong sums in accumulator register 3 - rows 3 mod 4 of every tile, T = 2, 10, 18, 26 at
        // C = 256; nothing here depends on run-time control flow around the MFMAs any more.  tests/test_gpu_stages.py sweeps T = 1 .. 26.)
        auto group = [&](auto tn_c, const int tg0, const int tn) __attribute__((always_inline)) {
            constexpr int TN = decltype(tn_c)::value;
            f32x4 acc[TN][NT];
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto chunk = [&](int kc, const uint32_t (&w_cur)[TN]) {
                const unsigned char* bs = b_rd + (size_t)(a.resident ? kc : (kc & 1)) * slot_bytes;
                bf16x8 b[3][NT];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        b[pl][nt] = *reinterpret_cast<const bf16x8*>(bs + (pl * NOp + nt * 16) * 64);
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(lut + (__builtin_amdgcn_ubfe(w_cur[t], lg8, 8) << 4));
#pragma unroll
                    for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b[pl][nt], acc[t][nt], 0, 0, 0);
                }
            };
            if (a.resident && Kc == 8) {
                // C = 256: the 8 spike words of a (t, row) are one 32-byte line - all T x 8 words are requested up
                // front (one memory latency per row tile instead of one per chunk)
                uint4 wl[TN][2];
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    const uint4* q = reinterpret_cast<const uint4*>(wsrc + (size_t)(tg0 + (t < tn ? t : 0)) * a.spk_stride);
                    wl[t][0] = q[0]; wl[t][1] = a.half_split ? q[(size_t)a.M] : q[1];     // second half: M x 16 bytes on
                    if (t >= tn) { wl[t][0] = uint4{0u, 0u, 0u, 0u}; wl[t][1] = uint4{0u, 0u, 0u, 0u}; }
                }
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    uint32_t w_cur[TN];
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const uint4 v = wl[t][kc >> 2];
                        w_cur[t] = (kc & 3) == 0 ? v.x : (kc & 3) == 1 ? v.y : (kc & 3) == 2 ? v.z : v.w;
                    }
                    chunk(kc, w_cur);
                }
            } else {
                uint32_t w_nxt[TN];
#pragma unroll
                for (int t = 0; t < TN; ++t) w_nxt[t] = t < tn ? wsrc[(size_t)(tg0 + t) * a.spk_stride] : 0u;
                if (!a.resident) { stage(0, 0); __syncthreads(); }
                for (int kc = 0; kc < Kc; ++kc) {
                    uint32_t w_cur[TN];
#pragma unroll
                    for (int t = 0; t < TN; ++t) w_cur[t] = w_nxt[t];
                    if (kc + 1 < Kc) {
#pragma unroll
                        for (int t = 0; t < TN; ++t) w_nxt[t] = t < tn ? wsrc[(size_t)(tg0 + t) * a.spk_stride + kc + 1] : 0u;
                        if (!a.resident) stage(kc + 1, (kc + 1) & 1);
                    }
                    chunk(kc, w_cur);
                    if (!a.resident) __syncthreads();           // chunk kc+1 staged, chunk kc consumed
                }
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int ti = min(tg0 + t, a.T - 1);           // (steps past tn: their accumulators are zero)
                const float kl = a.kap.last[ti], ks = a.kap.sum[ti];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o_last[nt][r] = fmaf(kl, acc[t][nt][r], o_last[nt][r]);
                        o_sum[nt][r] = fmaf(ks, acc[t][nt][r], o_sum[nt][r]);
                    }
            }
        };
        for (int tg0 = 0; tg0 < a.T; tg0 += LIH_TG) {
            const int tn = min(LIH_TG, a.T - tg0);              // block-uniform
            if (tn > 4) group(std::integral_constant<int, 8>{}, tg0, tn);
            else if (tn > 2) group(std::integral_constant<int, 4>{}, tg0, tn);
            else if (tn == 2) group(std::integral_constant<int, 2>{}, tg0, tn);
            else group(std::integral_constant<int, 1>{}, tg0, tn);
        }
        // lane holds rows lg*4 + r, output column nt*16 + lr
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int j = a.col0 + nt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + lg * 4 + r;
                if (m >= a.M) continue;
                if (j < a.NA) { a.out_a[(size_t)m * a.NA + j] = o_last[nt][r]; if (a.sum_a) a.sum_a[(size_t)m * a.NA + j] = o_sum[nt][r]; }
                else if (j < a.NA + a.NB) {
                    a.out_b[(size_t)m * a.NB + (j - a.NA)] = o_last[nt][r];
                    if (a.sum_b) a.sum_b[(size_t)m * a.NB + (j - a.NA)] = o_sum[nt][r];
                }
            }
        }
    }
}

// K5c: the same heads when W does not fit in LDS (detector: K = 1024, 45 outputs) and there are few rows (2000): one
// work-group per 16-row tile, its 4 waves split the reduction into quarters (so 125 work-groups of independent waves
// instead of 32 that march through K together).  A wave keeps the accumulators of 16 time steps (12 with 4 column tiles;
// longer T runs in groups, each a pass over the wave's chunks), builds the
// three bf16 planes of its weight fragments in registers straight from the fp32 W^T (read through L2, next chunk's
// values requested before this chunk's MFMAs), and needs no barrier until the four partial results meet in LDS and are
// added in wave order (deterministic).  2000 x 1024 x 45, T = 12: 68 us (fp32 VALU kernel) -> ~15 us.
#define LIH_KS_TM(nt) ((nt) <= 2 ? 16 : 12)     // time steps per group: accumulators beside NT column tiles
template <int NT>
__global__ __launch_bounds__(256) void k_li_heads_ksplit(const LiHeadsArgs a) {
    constexpr int TM = LIH_KS_TM(NT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lut = smem;                           // 4 KB
    f32x4* const red = reinterpret_cast<f32x4*>(smem + G3_LUT_BYTES);      // [4 waves][2][NT][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4, lg8 = 8 * lg;
    const int NOp = a.NOp, Kc = a.Kw, T = a.T;
    {
        uint4 q;
        q.x = bf16_pair(tid, 0); q.y = bf16_pair(tid, 1); q.z = bf16_pair(tid, 2); q.w = bf16_pair(tid, 3);
        *reinterpret_cast<uint4*>(lut + tid * 16) = q;
    }
    __syncthreads();
    const int m0 = blockIdx.x * 16;
    const int mrow = min(m0 + lr, a.M - 1);                    // rows past M: recomputed, never stored
    const uint32_t* wsrc = a.spk + (size_t)mrow * a.Kw;
    const int c0 = wave * Kc / 4, c1 = (wave + 1) * Kc / 4;    // this wave's chunks
    // element (k = 32 kc + 8 lg + j, n = nt*16 + lr) of W^T: the lane's B fragment is j = 0..7
    const float* const wlane = a.wT + (size_t)lg8 * a.ldw + a.col0 + lr;
    f32x4 o_last[NT], o_sum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { o_last[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; o_sum[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // time steps in groups of TM (one group up to T = 16 / 12; T = 24: two passes over this wave's chunks).  As in k_li_heads_mfma the
    // number of steps with accumulators, TN, is a compile-time constant (TM, or 8 / 4 / 2 / 1 for a shorter last group); steps past
    // the group's real count tn multiply all-zero spike words - no run-time control flow around the MFMAs.
    auto group = [&](auto tn_c, const int tg0, const int tn) __attribute__((always_inline)) {
        constexpr int TN = decltype(tn_c)::value;
        f32x4 acc[TN][NT];
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float wf[NT][8];
        uint32_t w_nxt[TN];
        auto request = [&](int kc) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 8; ++j) wf[nt][j] = wlane[(size_t)(32 * kc + j) * a.ldw + nt * 16];
#pragma unroll
            for (int t = 0; t < TN; ++t) w_nxt[t] = t < tn ? wsrc[(size_t)(tg0 + t) * a.spk_stride + kc] : 0u;
        };
        if (c0 < c1) request(c0);
        for (int kc = c0; kc < c1; ++kc) {
            bf16x8 b[3][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w = wf[nt][j];
                    const uint16_t hi = f2bf_rn(w);
                    const float r1 = __fsub_rn(w, bf2f(hi));
                    const uint16_t mid = f2bf_rn(r1);
                    const uint16_t lo = f2bf_rn(__fsub_rn(r1, bf2f(mid)));
                    b[0][nt][j] = (short)hi; b[1][nt][j] = (short)mid; b[2][nt][j] = (short)lo;
                }
            uint32_t w_cur[TN];
#pragma unroll
            for (int t = 0; t < TN; ++t) w_cur[t] = w_nxt[t];
            if (kc + 1 < c1) request(kc + 1);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(lut + (__builtin_amdgcn_ubfe(w_cur[t], lg8, 8) << 4));
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b[pl][nt], acc[t][nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int ti = min(tg0 + t, T - 1);                 // (steps past tn: their accumulators are zero)
            const float kl = a.kap.last[ti], ks = a.kap.sum[ti];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o_last[nt][r] = fmaf(kl, acc[t][nt][r], o_last[nt][r]);
                    o_sum[nt][r] = fmaf(ks, acc[t][nt][r], o_sum[nt][r]);
                }
        }
    };
    for (int tg0 = 0; tg0 < T; tg0 += TM) {
        const int tn = min(TM, T - tg0);                        // block-uniform
        if (tn > 8) group(std::integral_constant<int, TM>{}, tg0, tn);
        else if (tn > 4) group(std::integral_constant<int, 8>{}, tg0, tn);
        else if (tn > 2) group(std::integral_constant<int, 4>{}, tg0, tn);
        else if (tn == 2) group(std::integral_constant<int, 2>{}, tg0, tn);
        else group(std::integral_constant<int, 1>{}, tg0, tn);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        red[((wave * 2 + 0) * NT + nt) * 64 + lane] = o_last[nt];
        red[((wave * 2 + 1) * NT + nt) * 64 + lane] = o_sum[nt];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f32x4 ol = red[(0 * NT + nt) * 64 + lane], os = red[(1 * NT + nt) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 pl = red[((w * 2 + 0) * NT + nt) * 64 + lane], ps = red[((w * 2 + 1) * NT + nt) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) { ol[r] = __fadd_rn(ol[r], pl[r]); os[r] = __fadd_rn(os[r], ps[r]); }
        }
        // lane holds rows lg*4 + r, output column nt*16 + lr
        const int j = a.col0 + nt * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + lg * 4 + r;
            if (m >= a.M) continue;
            if (j < a.NA) { a.out_a[(size_t)m * a.NA + j] = ol[r]; if (a.sum_a) a.sum_a[(size_t)m * a.NA + j] = os[r]; }
            else if (j < a.NA + a.NB) {
                a.out_b[(size_t)m * a.NB + (j - a.NA)] = ol[r];
                if (a.sum_b) a.sum_b[(size_t)m * a.NB + (j - a.NA)] = os[r];
            }
        }
    }
}

// spikes per image of one level, counted from the bit-planes (spike-rate mode of the bf16x3 path): blockIdx.x = image,
// blockIdx.y = slice of the image's words; integer atomics, so the result does not depend on the order.
// counts must be zeroed by the caller.
__global__ __launch_bounds__(256) void k_count_spikes(const uint32_t* __restrict__ spk, unsigned long long spk_stride,
                                                      int T, int words_per_image,
                                                      unsigned long long* __restrict__ counts) {
    unsigned long long sum = 0;
    for (int t = 0; t < T; ++t) {
        const uint32_t* src = spk + (size_t)t * spk_stride + (size_t)blockIdx.x * words_per_image;
        for (int i = blockIdx.y * 256 + threadIdx.x; i < words_per_image; i += gridDim.y * 256) sum += __popc(src[i]);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&counts[blockIdx.x], part[0] + part[1] + part[2] + part[3]);
}

// ------------------------------------------------------------------------------------------------
// Finished spike-rate tensors (rpn.py:171-195, faster_rcnn.py:568-618): [.., 2] = (rate, "FLOPs") rows as float32, from the
// integer spike counts of the LIF epilogues and the time-summed LI membranes of the head kernels.  Rates are formed in double
// and rounded once; the LI "rates" are means of membrane sums (the reference's quirk), reduced in a
// fixed order (bitwise repeatable).  The FLOP constants are int64 in the reference and become float32 in its hstack.
// ------------------------------------------------------------------------------------------------
#define RATE_CH 32                                   // partial sums per (level, image, head)
struct RpnRatesArgs {
    const unsigned long long* counts;                // [n_levels][max_n]
    const float* sum_l;                              // [P][A]   sum over t of the objectness membranes
    const float* sum_b;                              // [P][4A]
    double* part;                                    // [n_levels][max_n][2][RATE_CH]
    float* rates;                                    // [n_levels][3][max_n][2]
    int n_levels, max_n, A, T, C;
    int pos_base[SNN_MAX_LEVELS], N[SNN_MAX_LEVELS], HW[SNN_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void k_rpn_rate_partials(const RpnRatesArgs a) {
    const int ch = blockIdx.x, which = blockIdx.y, l = blockIdx.z / a.max_n, n = blockIdx.z % a.max_n;
    double acc = 0.0;
    if (n < a.N[l]) {
        const int width = which == 0 ? a.A : 4 * a.A;
        const float* src = (which == 0 ? a.sum_l : a.sum_b) + ((size_t)a.pos_base[l] + (size_t)n * a.HW[l]) * width;
        const size_t len = (size_t)a.HW[l] * width, lo = len * ch / RATE_CH, hi = len * (ch + 1) / RATE_CH;
        for (size_t i = lo + threadIdx.x; i < hi; i += 256) acc += (double)src[i];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ double wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) a.part[((size_t)blockIdx.z * 2 + which) * RATE_CH + ch] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void k_rpn_rate_final(const RpnRatesArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_levels * 3 * a.max_n) return;
    const int n = idx % a.max_n, j = (idx / a.max_n) % 3, l = idx / (3 * a.max_n);
    float rate = 0.0f, flops = 0.0f;
    if (n < a.N[l]) {
        const long long HW = a.HW[l];
        if (j == 0) {                                // shared LIF: spikes / (T * C * H * W); FLOPs 9 * HW * C^2 (rpn.py:177-180)
            rate = (float)((double)a.counts[l * a.max_n + n] / ((double)a.T * a.C * (double)HW));
            flops = (float)(9ll * HW * a.C * a.C);
        } else {                                     // LI heads: mean over (A, H, W) of sum_t membrane / T
            const double* p = a.part + ((size_t)(l * a.max_n + n) * 2 + (j - 1)) * RATE_CH;
            double sum = 0.0;
            for (int c = 0; c < RATE_CH; ++c) sum += p[c];
            const long long width = j == 1 ? a.A : 4 * a.A;
            rate = (float)(sum / (double)a.T / (double)(HW * width));
            // the reference labels these two the other way round (rpn.py:181-188): obj gets A*4, bbox gets A - kept as is
            flops = (float)(j == 1 ? HW * a.C * a.A * 4 : HW * a.C * a.A);
        }
    }
    float* o = a.rates + ((size_t)(l * 3 + j) * a.max_n + n) * 2;
    o[0] = rate; o[1] = flops;
}

// detector: rates[4][R][2] from the per-RoI spike counts of lif6 / lif7 and the time-summed LI membranes
__global__ void k_det_rates(const uint32_t* __restrict__ c6, const uint32_t* __restrict__ c7, const float* __restrict__ sum_c,
                            const float* __restrict__ sum_b, int R, long long D, long long Hd, int K, int K4, int T, int one_bbox,
                            float* __restrict__ rates) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double sc = 0.0, sb = 0.0;
    for (int k = 0; k < K; ++k) sc += (double)sum_c[(size_t)r * K + k];
    for (int k = 0; k < K4; ++k) sb += (double)sum_b[(size_t)r * K4 + k];
    const float v[4][2] = {{(float)((double)c6[r] / ((double)T * Hd)), (float)(D * Hd)},
                           {(float)((double)c7[r] / ((double)T * Hd)), (float)(Hd * Hd)},
                           {(float)(sc / T / K), (float)(Hd * K)},
                           {(float)(sb / T / K4), (float)(one_bbox ? Hd * K : Hd * K * 4)}};      // faster_rcnn.py:575-593
    for (int j = 0; j < 4; ++j) {
        rates[((size_t)j * R + r) * 2] = v[j][0];
        rates[((size_t)j * R + r) * 2 + 1] = v[j][1];
    }
}
