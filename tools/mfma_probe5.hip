// probe 5: what the bf16 matrix pipe SUSTAINS (>= 2 s of back-to-back launches, not a 26-ms burst) in probe 4's LDS-fed
// 16x16x32 loop, by operand content: the clock the chip holds under MFMA load depends on what the multipliers toggle.
// Prints executed TFLOP/s over the last second and the in-kernel clock (s_memtime cycles per s_memrealtime 100-MHz tick).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// a_rate_256: spike probability of the A operand in 1/256; b_mode 0: zero weights, 1: random sign + 7 mantissa bits, exponent ~[-2,2)
template <int SWAP>
__global__ __launch_bounds__(512) void probe(const uint32_t* __restrict__ seed, float* out, unsigned long long* clk, int iters,
                                             int a_rate_256, int b_mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* l32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = threadIdx.x; i < 81920 / 4; i += blockDim.x) {
        uint32_t x = seed[i & 1023] * 2654435761u + i * 40503u + blockIdx.x * 977u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        if (i < 8192) l32[i] = (((x & 255) < (uint32_t)a_rate_256) ? 0x3F80u : 0u) | ((((x >> 8) & 255) < (uint32_t)a_rate_256) ? 0x3F800000u : 0u);
        else l32[i] = b_mode ? ((x & 0x807F807Fu) | 0x3F003F00u) : 0u;
    }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    const int lr = lane & 15, lg = lane >> 4;
    const unsigned char* Ab = smem + (wm * 64 + lr) * 64 + (lg << 4);
    const unsigned char* Bb = smem + 32768 + (wn * 64 + lr) * 64 + (lg << 4);
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[4];
        for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(Ab + mt * 1024);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x8 b[3];
            for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(Bb + pl * 8192 + nt * 1024);
            for (int mt = 0; mt < 4; ++mt) for (int pl = 0; pl < 3; ++pl)
                acc[mt][nt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[pl], a[mt], acc[mt][nt], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[pl], acc[mt][nt], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int SWAP>
static void run(const char* name, int a_rate, int b_mode, const uint32_t* seed, float* out, unsigned long long* clk, double seconds) {
    const int iters = 2000, grid = 512;                  // ~2.6 ms per launch, two work-groups per CU
    (void)hipFuncSetAttribute((const void*)probe<SWAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flop = (double)grid * 8 * iters * 64.0 * 64.0 * 32.0 * 2.0 * 3.0;
    auto t0 = std::chrono::steady_clock::now();
    double burst = 0, last = 0;
    int round = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(probe<SWAP>, dim3(grid), dim3(512), 81920, 0, seed, out, clk, iters, a_rate, b_mode);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        last = 50 * flop / (ms * 1e-3) / 1e12;
        if (round++ == 0) burst = last;
    }
    static unsigned long long h[1024];
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0;
    for (int i = 0; i < 512; ++i) { cyc += (double)h[2 * i]; ticks += (double)h[2 * i + 1]; }
    printf("%-44s first 0.13 s %5.0f TF, after %.0f s %5.0f TF executed, in-kernel clock %.3f GHz\n", name, burst, seconds, last, cyc / ticks * 0.1);
    fflush(stdout);
}

int main() {
    uint32_t* seed; float* out; unsigned long long* clk;
    (void)hipMalloc(&seed, 4096); (void)hipMalloc(&out, 512 * 512 * 4); (void)hipMalloc(&clk, 1024 * 8);
    uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i * 747796405u + 2891336453u;
    (void)hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
    run<0>("A = 0, B = 0", 0, 0, seed, out, clk, 3.0);
    run<0>("A = 0, B random", 0, 1, seed, out, clk, 3.0);
    run<0>("A spikes 1/3 (the workload), B random", 85, 1, seed, out, clk, 3.0);
    run<0>("A spikes 1/2, B random", 128, 1, seed, out, clk, 3.0);
    run<0>("A all ones, B random", 256, 1, seed, out, clk, 3.0);
    run<0>("A spikes 1/3 (the workload), B random", 85, 1, seed, out, clk, 3.0);
    // (round 3) the same with the operand roles swapped: weights as the MFMA's first operand, spikes as its second (the accumulator
    // then holds the transposed tile) - does the multiplier array care which side is the sparse 0 / 1.0 one?
    run<1>("SWAPPED: A random, B spikes 1/3", 85, 1, seed, out, clk, 3.0);
    run<0>("A spikes 1/3 (the workload), B random", 85, 1, seed, out, clk, 3.0);
    run<1>("SWAPPED: A random, B spikes 1/3", 85, 1, seed, out, clk, 3.0);
    return 0;
}
