// probe 2: hand-scheduled chunk (inline asm): a-values produced 2 q-steps ahead into 4 rotating VGPRs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define STR2(x) #x
#define STR(x) STR2(x)
// one q-step: 2 MFMAs sharing a (NT=2) + produce a for q+2
#define QSTEP(q, tcur, tnext, bA, bB, el)                                     \
    "v_mfma_f32_32x32x2_f32 %[acc0], " tcur ", %[" bA "], %[acc0]\n\t"           \
    "v_mfma_f32_32x32x2_f32 %[acc1], " tcur ", %[" bB "], %[acc1]\n\t"           \
    "v_bfe_u32 " tnext ", %[w], " STR(q) " + 2, 1\n\t"                          \
    "v_cvt_f32_ubyte0 " tnext ", " tnext "\n\t"

template <int VARIANT>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ w, const uint32_t* __restrict__ aw,
                                             float* out, int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    uint32_t word = aw[threadIdx.x];
    const f32x4* wp = reinterpret_cast<const f32x4*>(w) + lane + (size_t)wave * 256;
    float b[2][16];
    for (int n = 0; n < 2; ++n) for (int qq = 0; qq < 4; ++qq) { f32x4 t = wp[qq * 64 + n * 1024]; for (int r = 0; r < 4; ++r) b[n][qq * 4 + r] = t[r]; }
    float t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) {
            // a for q and q+1 prepared before; each step prepares q+2
            asm volatile(
                "v_bfe_u32 %[t0], %[w], 0, 1\n\tv_cvt_f32_ubyte0 %[t0], %[t0]\n\t"
                "v_bfe_u32 %[t1], %[w], 1, 1\n\tv_cvt_f32_ubyte0 %[t1], %[t1]\n\t"
                "s_nop 1\n\t"
#define Q(q, tc, tn, i) "v_mfma_f32_32x32x2_f32 %[acc0], %[" tc "], %[b0_" #i "], %[acc0]\n\tv_mfma_f32_32x32x2_f32 %[acc1], %[" tc "], %[b1_" #i "], %[acc1]\n\tv_bfe_u32 %[" tn "], %[w], " #q "+2, 1\n\tv_cvt_f32_ubyte0 %[" tn "], %[" tn "]\n\t"
                Q(0, "t0", "t2", 0) Q(1, "t1", "t3", 1) Q(2, "t2", "t0", 2) Q(3, "t3", "t1", 3)
                Q(4, "t0", "t2", 4) Q(5, "t1", "t3", 5) Q(6, "t2", "t0", 6) Q(7, "t3", "t1", 7)
                Q(8, "t0", "t2", 8) Q(9, "t1", "t3", 9) Q(10, "t2", "t0", 10) Q(11, "t3", "t1", 11)
                Q(12, "t0", "t2", 12) Q(13, "t1", "t3", 13) Q(14, "t2", "t0", 14) Q(15, "t3", "t1", 15)
                : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3)
                : [w] "v"(word),
                  [b0_0] "v"(b[0][0]), [b0_1] "v"(b[0][1]), [b0_2] "v"(b[0][2]), [b0_3] "v"(b[0][3]), [b0_4] "v"(b[0][4]), [b0_5] "v"(b[0][5]), [b0_6] "v"(b[0][6]), [b0_7] "v"(b[0][7]),
                  [b0_8] "v"(b[0][8]), [b0_9] "v"(b[0][9]), [b0_10] "v"(b[0][10]), [b0_11] "v"(b[0][11]), [b0_12] "v"(b[0][12]), [b0_13] "v"(b[0][13]), [b0_14] "v"(b[0][14]), [b0_15] "v"(b[0][15]),
                  [b1_0] "v"(b[1][0]), [b1_1] "v"(b[1][1]), [b1_2] "v"(b[1][2]), [b1_3] "v"(b[1][3]), [b1_4] "v"(b[1][4]), [b1_5] "v"(b[1][5]), [b1_6] "v"(b[1][6]), [b1_7] "v"(b[1][7]),
                  [b1_8] "v"(b[1][8]), [b1_9] "v"(b[1][9]), [b1_10] "v"(b[1][10]), [b1_11] "v"(b[1][11]), [b1_12] "v"(b[1][12]), [b1_13] "v"(b[1][13]), [b1_14] "v"(b[1][14]), [b1_15] "v"(b[1][15]));
#undef Q
        } else if (VARIANT == 1) {
            // same MFMAs, NO VALU at all (a = stale t regs): upper bound of the asm structure
            asm volatile(
#define Q(q, tc, tn, i) "v_mfma_f32_32x32x2_f32 %[acc0], %[" tc "], %[b0_" #i "], %[acc0]\n\tv_mfma_f32_32x32x2_f32 %[acc1], %[" tc "], %[b1_" #i "], %[acc1]\n\t"
                Q(0, "t0", "t2", 0) Q(1, "t1", "t3", 1) Q(2, "t2", "t0", 2) Q(3, "t3", "t1", 3)
                Q(4, "t0", "t2", 4) Q(5, "t1", "t3", 5) Q(6, "t2", "t0", 6) Q(7, "t3", "t1", 7)
                Q(8, "t0", "t2", 8) Q(9, "t1", "t3", 9) Q(10, "t2", "t0", 10) Q(11, "t3", "t1", 11)
                Q(12, "t0", "t2", 12) Q(13, "t1", "t3", 13) Q(14, "t2", "t0", 14) Q(15, "t3", "t1", 15)
                : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3)
                : [w] "v"(word),
                  [b0_0] "v"(b[0][0]), [b0_1] "v"(b[0][1]), [b0_2] "v"(b[0][2]), [b0_3] "v"(b[0][3]), [b0_4] "v"(b[0][4]), [b0_5] "v"(b[0][5]), [b0_6] "v"(b[0][6]), [b0_7] "v"(b[0][7]),
                  [b0_8] "v"(b[0][8]), [b0_9] "v"(b[0][9]), [b0_10] "v"(b[0][10]), [b0_11] "v"(b[0][11]), [b0_12] "v"(b[0][12]), [b0_13] "v"(b[0][13]), [b0_14] "v"(b[0][14]), [b0_15] "v"(b[0][15]),
                  [b1_0] "v"(b[1][0]), [b1_1] "v"(b[1][1]), [b1_2] "v"(b[1][2]), [b1_3] "v"(b[1][3]), [b1_4] "v"(b[1][4]), [b1_5] "v"(b[1][5]), [b1_6] "v"(b[1][6]), [b1_7] "v"(b[1][7]),
                  [b1_8] "v"(b[1][8]), [b1_9] "v"(b[1][9]), [b1_10] "v"(b[1][10]), [b1_11] "v"(b[1][11]), [b1_12] "v"(b[1][12]), [b1_13] "v"(b[1][13]), [b1_14] "v"(b[1][14]), [b1_15] "v"(b[1][15]));
#undef Q
        } else if (VARIANT == 2) {
            // VALU present but INDEPENDENT of the MFMAs (writes t2/t3 only, MFMAs read t0/t1)
            asm volatile(
#define Q(q, tc, tn, i) "v_mfma_f32_32x32x2_f32 %[acc0], %[" tc "], %[b0_" #i "], %[acc0]\n\tv_mfma_f32_32x32x2_f32 %[acc1], %[" tc "], %[b1_" #i "], %[acc1]\n\tv_bfe_u32 %[" tn "], %[w], " #q ", 1\n\tv_cvt_f32_ubyte0 %[" tn "], %[" tn "]\n\t"
                Q(0, "t0", "t2", 0) Q(1, "t1", "t3", 1) Q(2, "t0", "t2", 2) Q(3, "t1", "t3", 3)
                Q(4, "t0", "t2", 4) Q(5, "t1", "t3", 5) Q(6, "t0", "t2", 6) Q(7, "t1", "t3", 7)
                Q(8, "t0", "t2", 8) Q(9, "t1", "t3", 9) Q(10, "t0", "t2", 10) Q(11, "t1", "t3", 11)
                Q(12, "t0", "t2", 12) Q(13, "t1", "t3", 13) Q(14, "t0", "t2", 14) Q(15, "t1", "t3", 15)
                : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3)
                : [w] "v"(word),
                  [b0_0] "v"(b[0][0]), [b0_1] "v"(b[0][1]), [b0_2] "v"(b[0][2]), [b0_3] "v"(b[0][3]), [b0_4] "v"(b[0][4]), [b0_5] "v"(b[0][5]), [b0_6] "v"(b[0][6]), [b0_7] "v"(b[0][7]),
                  [b0_8] "v"(b[0][8]), [b0_9] "v"(b[0][9]), [b0_10] "v"(b[0][10]), [b0_11] "v"(b[0][11]), [b0_12] "v"(b[0][12]), [b0_13] "v"(b[0][13]), [b0_14] "v"(b[0][14]), [b0_15] "v"(b[0][15]),
                  [b1_0] "v"(b[1][0]), [b1_1] "v"(b[1][1]), [b1_2] "v"(b[1][2]), [b1_3] "v"(b[1][3]), [b1_4] "v"(b[1][4]), [b1_5] "v"(b[1][5]), [b1_6] "v"(b[1][6]), [b1_7] "v"(b[1][7]),
                  [b1_8] "v"(b[1][8]), [b1_9] "v"(b[1][9]), [b1_10] "v"(b[1][10]), [b1_11] "v"(b[1][11]), [b1_12] "v"(b[1][12]), [b1_13] "v"(b[1][13]), [b1_14] "v"(b[1][14]), [b1_15] "v"(b[1][15]));
#undef Q
        }
        word = word * 1664525u + 1013904223u;
    }
    float s = t0 + t1 + t2 + t3;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
static void run(const char* name, int threads, const float* w, const uint32_t* aw, float* out) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<V>), dim3(grid), dim3(threads), 0, 0, w, aw, out, 50);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<V>), dim3(grid), dim3(threads), 0, 0, w, aw, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * (threads / 64) * iters * 32.0;
    printf("%-52s thr=%d  %.3f ms  %.1f TFLOP/s\n", name, threads, ms, mfma * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    float* w; uint32_t* aw; float* out;
    (void)hipMalloc(&w, 1 << 24); (void)hipMalloc(&aw, 4096); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemset(w, 0, 1 << 24); (void)hipMemset(aw, 0x5a, 4096);
    run<1>("asm: MFMA only", 256, w, aw, out);
    run<1>("asm: MFMA only, 2 waves/SIMD", 512, w, aw, out);
    run<2>("asm: MFMA + independent bfe+cvt", 256, w, aw, out);
    run<2>("asm: MFMA + independent bfe+cvt, 2 waves/SIMD", 512, w, aw, out);
    run<0>("asm: a two q-steps ahead, rotating regs", 256, w, aw, out);
    run<0>("asm: a two q-steps ahead, rotating regs, 2 w/SIMD", 512, w, aw, out);
    return 0;
}
