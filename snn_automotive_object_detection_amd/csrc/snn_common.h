// Shared device helpers for the gfx950 spiking-head kernels.
// Neuron arithmetic follows Norse 0.0.7 op-for-op (see include/snn_hip.h for the reference call
// sites); every fp32 operation is an explicitly rounded __f*_rn so that hipcc can never contract
// a multiply-add into an fma (the reference's element-wise torch kernels round twice).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct NeuronP {
    float ca;       // fl32(dt * tau_mem_inv)
    float cb;       // fl32(-dt * tau_syn_inv)
    float v_leak;
    float v_reset;
    float v_th;
    // encoders with zero rest / reset potentials only: the membrane value after a spike.  +0 = the reference's reset.
    // ENC_FIRED (-infinity) = "period planes": the neuron then never crosses the threshold again, so plane
    // t holds the neurons whose FIRST spike is at step t (see PERIOD PLANES below).  -inf is a LATCH for every input and every
    // dt * tau_mem_inv: the next update forms -inf + ca * (x + inf) = NaN (or stays -inf for x = -inf / ca <= 0), and neither
    // ever compares greater than the threshold.  (Until round 4 the sentinel was -1e30, which decays as (1 - ca)^k and let a
    // neuron fire again within T for ca >= ~0.9 or inputs >= ~1e31: the e_n planes were then not disjoint.)
    float v_fire;
};
#define ENC_FIRED (-__builtin_inff())

// PERIOD PLANES.  The constant-current encoder (lif_current_encoder with v = 0 at the start and reset to 0, which is what the
// reference builds: rpn.py:58,93,101 / faster_rcnn.py:444,484,494) is exactly periodic: after a spike the membrane is +0 again,
// the same state it started from, so the same fp32 operations repeat - the spike train of a neuron is
//     z_t = 1  iff  n divides t + 1,      n = (index of its first spike) + 1,
// and the encoder's spike planes are  z_t = OR over the divisors n of t + 1 of e_n  with DISJOINT planes e_n = (period == n).
// A bias-free layer is linear, so its input currents are  cur_t = W z_t = sum over n | (t + 1) of u_n,  u_n = W e_n:  the
// matrix-core kernels multiply the e_n planes (for the bench's features: densities 0.27, 0.09, 0.03, 0.02, ... against
// 0.27-0.40 for every z_t plane) and the LIF epilogue adds up the u_n of the divisors (Gemm3Args.periods).  Same MFMA count,
// but the kernels are power-limited and a quarter of the operand switching buys clock: conv+LIF -4 %, fc6+LIF -8 % on the
// bench's planes (tools/period_probe.py).  The sums are fp32 additions of partial chains - as accurate as one chain
// (shorter chains, 1-3 extra roundings); against the oracle the results are one more fp32 summation order.

// lif_current_encoder (norse/torch/functional/lif.py; reference rpn.py:101, faster_rcnn.py:494)
//   dv = dt*tau_mem_inv * ((v_leak - v) + x);  v = v + dv;  z = (v - v_th > 0);  v = v - z*(v - v_reset)
__device__ __forceinline__ bool enc_step(const float x, float& v, const NeuronP& p) {
    const float dv = __fmul_rn(p.ca, __fadd_rn(__fsub_rn(p.v_leak, v), x));
    v = __fadd_rn(v, dv);
    const bool z = __fsub_rn(v, p.v_th) > 0.0f;
    const float d = __fsub_rn(v, p.v_reset);
    v = z ? __fsub_rn(v, d) : v;
    return z;
}

// The encoders are VALU-bound, not HBM-bound (10 instructions per neuron-step against 4 bytes read once), and packed
// fp32 instructions run at half rate on gfx950, so what helps is fewer operations.  ZR = (v_leak == 0 && v_reset == 0),
// Norse's defaults and the reference's: three identities, each exact in IEEE arithmetic with gradual underflow,
//   (0 - v) + x == x - v,      v - (v - 0) == +0 (finite v),      (v - v_th > 0) == (v > v_th)
// bring a step to  sub, mul, add, cmp, select  (+ v_addc for  word = 2*word + spike).  Any other parameters take the
// op-for-op path.  Callers of enc_step_word feed the elements of a plane word from bit 31 down to bit 0.
template <bool ZR>
__device__ __forceinline__ void enc_step_word(const float x, float& v, const NeuronP& p, uint32_t& word) {
    if (ZR) {
        v = __fadd_rn(v, __fmul_rn(p.ca, __fsub_rn(x, v)));
        const float fire = p.v_fire;                    // +0 (the reset) or ENC_FIRED (period planes): one loop-invariant register
        asm("v_cmp_lt_f32 vcc, %2, %1\n\tv_cndmask_b32 %1, %1, %3, vcc\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
            : "+v"(word), "+v"(v) : "s"(p.v_th), "v"(fire) : "vcc");
    } else {
        v = __fadd_rn(v, __fmul_rn(p.ca, __fadd_rn(__fsub_rn(p.v_leak, v), x)));
        const float th = __fsub_rn(v, p.v_th), vr = __fsub_rn(v, __fsub_rn(v, p.v_reset));
        asm("v_cmp_lt_f32 vcc, 0, %2\n\tv_cndmask_b32 %1, %1, %3, vcc\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
            : "+v"(word), "+v"(v) : "v"(th), "v"(vr) : "vcc");
    }
}
// THRESHOLD FORM of the period-plane encoder.  "First spike at or before step t" is, for the zero-rest encoder, the same as
// x >= th[t] for a table of fp32 thresholds th[0] > th[1] > ... (the membrane before the first spike is a non-decreasing function of
// the input): the encoder is a quantiser.  The table is found on the host by bisection over the fp32 bit patterns with the device's
// exact operation sequence and VERIFIED on every float within 1024 ulps of each threshold (rounding can only matter that close: the
// accumulated error of t steps is < 1e-5 relative); launchers fall back to the recurrence if the verification ever fails
// (snn_kernels.hip: enc_thresholds).  Two instructions per neuron-step instead of six: the cumulative plane word
// C_t = (first spike at or before t), and e_t = C_t & ~C_{t-1}.
struct EncTh { float th[32]; };
__device__ __forceinline__ void enc_quant_word(const float x, const float th, uint32_t& word) {     // word = 2 * word + (th <= x)
    asm("v_cmp_le_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(word) : "v"(x), "s"(th) : "vcc");
}
enum { ENC_GENERIC = 0, ENC_ZR = 1, ENC_QUANT = 2 };

// same arithmetic, the spike returned as a predicate (for ballot-based plane words)
template <bool ZR>
__device__ __forceinline__ bool enc_step_t(const float x, float& v, const NeuronP& p) {
    if (ZR) {
        v = __fadd_rn(v, __fmul_rn(p.ca, __fsub_rn(x, v)));
        const bool z = v > p.v_th;
        v = z ? p.v_fire : v;
        return z;
    }
    return enc_step(x, v, p);
}

// lif_feed_forward_step (norse lif.py; reference rpn.py:106, faster_rcnn.py:499,501)
//   v_dec = v + ca*((v_leak - v) + i)   (OLD i);   i_dec = i + cb*i;   z = (v_dec - v_th > 0)
//   v = (1-z)*v_dec + z*v_reset;   i = i_dec + cur
__device__ __forceinline__ bool lif_step(const float cur, float& v, float& i, const NeuronP& p) {
    const float dv = __fmul_rn(p.ca, __fadd_rn(__fsub_rn(p.v_leak, v), i));
    const float v_dec = __fadd_rn(v, dv);
    const float i_dec = __fadd_rn(i, __fmul_rn(p.cb, i));
    const bool z = __fsub_rn(v_dec, p.v_th) > 0.0f;
    v = z ? p.v_reset : v_dec;
    i = __fadd_rn(i_dec, cur);
    return z;
}

// The decay / integrate part of the same step on four neurons at once (one 16x16 MFMA accumulator): plain vector
// expressions, which hipcc lowers to packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32: IEEE-rounded per
// element, and the kernels are built with -ffp-contract=off, so nothing is fused).  The caller thresholds
// d = v_dec - v_th per element (z = d > 0) and sets v = z ? v_reset : v_dec.
__device__ __forceinline__ void lif_decay4(const f32x4 cur, const f32x4 v, f32x4& i, const NeuronP& p, f32x4& v_dec, f32x4& d) {
    v_dec = v + p.ca * ((p.v_leak - v) + i);
    const f32x4 i_dec = i + p.cb * i;
    d = v_dec - p.v_th;
    i = i_dec + cur;
}

// row of a 32x32 MFMA accumulator held in register r by lane-half h (col = lane & 31)
__device__ __forceinline__ int acc_row(const int r, const int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// One 32-deep reduction chunk on the fp32 matrix cores.
//   a[mt][qq] : 4 consecutive fp32 spike values (0.0f / 1.0f) of row (lane&31) of M-tile mt, read by one
//               ds_read_b128 from the expanded LDS image: reduction indices k = 4*qq + r + 16*(lane>>5)
//   b[qq]     : weights W[k = 4*qq + r + 16*(lane>>5)][n = lane&31], one coalesced 1-KiB wave load each
//               from the fragment-major packed image
// The loop body is MFMA only.  On gfx950 the fp32 MFMA (64 FLOP/clk/SIMD = the VALU rate) loses ~6.5
// cycles of matrix-pipe time per VALU instruction issued beside it (tools/mfma_probe*.hip), so spike
// bits are never converted inside this loop.
// v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32 fma chain (one rounding per product).
template <int MT>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[MT], const f32x4 (&a)[MT][4], const f32x4 (&b)[4]) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][qq][r], b[qq][r], acc[mt], 0, 0, 0);
}

// half a chunk (qq = 2*HALF, 2*HALF+1): lets the A operands be fetched from LDS half a chunk ahead with
// only 8*MT live registers per half
template <int MT, int HALF>
__device__ __forceinline__ void mma_half(f32x16 (&acc)[MT], const f32x4 (&a)[MT][2], const f32x4 (&b)[4]) {
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][q2][r], b[2 * HALF + q2][r], acc[mt], 0, 0, 0);
}

// 32 spike bits -> 32 fp32 values (0.0f / 1.0f) written as 8 x 16 B to dst (LDS)
__device__ __forceinline__ void expand_word(uint32_t w, float* dst) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)((w >> (4 * j + r)) & 1u);
        *reinterpret_cast<f32x4*>(dst + 4 * j) = v;
    }
}

// ---- compressed period planes of the structured-sparse launches (snn_sparse.h; written by k_compress_planes and, for the RPN head, by the
// encoder launch itself: snn_encode.h) ----
#define SP_A_ARR 4                                  // dword arrays per step: dense rows use two (the chunks' spike words), sparse rows all four:
                                                    // occupancy bytes of the four 16-k blocks, index halves 0-1, index halves 2-3 of the primary
                                                    // plane, occupancy bytes of the secondary plane (whose indices are the constant (2, 3))
// nibble -> (two occupancy bits, two 2-bit positions, leftover bits).  bits p0 < p1: slots (1, p0), (1, p1); one bit p < 3: (1, p), (0, 3);
// p == 3: (0, 0), (1, 3); none: (0, 0), (0, 3): index 0 < index 1 always (tools/sparse_probe.hip A4 ran exactly this encoding).
__host__ __device__ inline uint32_t sp_nibble_code(uint32_t x) {
    int p0 = -1, p1 = -1;
    uint32_t left = 0;
    for (int b = 0; b < 4; ++b)
        if (x & (1u << b)) {
            if (p0 < 0) p0 = b;
            else if (p1 < 0) p1 = b;
            else left |= 1u << b;
        }
    uint32_t occ, i0, i1;
    if (p1 >= 0) { occ = 3; i0 = p0; i1 = p1; }
    else if (p0 >= 0 && p0 < 3) { occ = 1; i0 = p0; i1 = 3; }
    else if (p0 == 3) { occ = 2; i0 = 0; i1 = 3; }
    else { occ = 0; i0 = 0; i1 = 3; }
    return occ | ((i0 | (i1 << 2)) << 2) | (left << 6);       // bits 1:0 occupancy, 5:2 indices, 9:6 leftover
}

// byte (two nibbles) -> bits 3:0 occupancy, 11:4 indices, 15:12 leftover bits 2 / 3 of the two nibbles: one table look-up per 8 k
__host__ __device__ inline uint16_t sp_byte_code(uint32_t b) {
    const uint32_t c0 = sp_nibble_code(b & 15u), c1 = sp_nibble_code((b >> 4) & 15u);
    return (uint16_t)((c0 & 3u) | ((c1 & 3u) << 2) | (((c0 >> 2) & 15u) << 4) | (((c1 >> 2) & 15u) << 8) | (((c0 >> 8) & 3u) << 12) | (((c1 >> 8) & 3u) << 14));
}
// the four dwords of a (row, 64 k) step from its two spike words: primary occupancy (2 bits per nibble), index halves (4 bits per nibble)
// of word 0 / word 1, secondary occupancy.  `code` = sp_byte_code of 0 .. 255 (in LDS)
__device__ __forceinline__ void sp_compress_pair(const uint32_t wd0, const uint32_t wd1, const uint16_t* code, uint32_t (&out)[4]) {
    const uint32_t wd[2] = {wd0, wd1};
    uint32_t occ = 0, idx[2] = {0, 0}, occ2 = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t c = code[(wd[h] >> (8 * b)) & 255u];
            occ |= (c & 15u) << (16 * h + 4 * b);
            idx[h] |= ((c >> 4) & 255u) << (8 * b);
            occ2 |= (c >> 12) << (16 * h + 4 * b);
        }
    out[0] = occ; out[1] = idx[0]; out[2] = idx[1]; out[3] = occ2;
}
