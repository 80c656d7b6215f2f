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
__host__ __device__ inline size_t mx_words(int Kc, int Np) { return mx_x_words(Kc, Np) + mx_y_words(Kc, Np) + (size_t)Kc * Np; }

__global__ void k_pack_mx(const float* __restrict__ src, uint32_t* __restrict__ dst, int mode, int K, int N, int Kc, int Np,
                          int Cin, int Cp) {
    const size_t blocks = (size_t)Kc * Np * 4;
    uint32_t* const X = dst;
    uint32_t* const Y = dst + mx_x_words(Kc, Np);
    uint8_t* const S = reinterpret_cast<uint8_t*>(dst + mx_x_words(Kc, Np) + mx_y_words(Kc, Np));
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
