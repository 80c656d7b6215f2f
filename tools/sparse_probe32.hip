// Round-5 go / no-go probe: would the 32-row structured-sparse shape v_smfmac_f32_32x32x32_bf16 beat the
// v_smfmac_f32_16x16x64_bf16 loop of k_gemm_lif_sparse?  Same wave tile (64 rows x 64 columns x 64 k per step, three weight planes, every
// fragment re-read from LDS as in the production loop), so the LDS traffic is IDENTICAL by construction - a wave reads its 64 x 64 k of A
// and the 64 k x 64 columns x 3 planes of B once per step whatever the instruction shape; what differs is the instruction count (half),
// the accumulator layout and the clock the chip holds (MI355X_MICROARCH.md, DVFS give-back (7): the 16-row bf16 shape held 1.12-1.15 x the
// FLOP/s of the 32-row one).  Reports dense-equivalent TFLOP/s sustained over 2 s and the in-kernel clock, plus the register-only rate.
//
//  hipcc --offload-arch=gfx950 -O3 -o tools/_ab/sparse_probe32 tools/sparse_probe32.hip && tools/_ab/sparse_probe32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// SHAPE 0: 16x16x64 (4 x 4 tiles per wave and 64-k step); 1: 32x32x32 (2 x 2 tiles x 2 k halves)
template <int SHAPE, int LDS_FED>
__global__ __launch_bounds__(512) void loop_probe(const uint32_t* __restrict__ seed, float* out, unsigned long long* clk, int iters, int a_rate_256) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* l32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = threadIdx.x; i < 98304 / 4; i += blockDim.x) {
        uint32_t x = seed[i & 1023] * 2654435761u + i * 40503u + blockIdx.x * 977u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        if (i < 8192) l32[i] = (((x & 255) < (uint32_t)a_rate_256) ? 0x3F80u : 0u) | ((((x >> 8) & 255) < (uint32_t)a_rate_256) ? 0x3F800000u : 0u);
        else l32[i] = (x & 0x807F807Fu) | 0x3F003F00u;
    }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const unsigned char* Ab = smem + (wave * 64 + lr) * 64 + (lg << 4);
    const unsigned char* Bb = smem + 32768 + lr * 64 + (lg << 4);
    const int idx = 0x44444444;
    float s = 0.f;
    if (SHAPE == 0) {
        f32x4 acc[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
        bf16x8 a[4];
        bf16x16 b[3];
        for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(Ab + mt * 1024);
        for (int pl = 0; pl < 3; ++pl) for (int i = 0; i < 16; ++i) b[pl][i] = (__bf16)(0.5f + 0.125f * ((lane + i + pl) & 7));
        for (int it = 0; it < iters; ++it) {
            if (LDS_FED) for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(Ab + mt * 1024);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (LDS_FED)
                    for (int pl = 0; pl < 3; ++pl) {
                        const bf16x8 lo = *reinterpret_cast<const bf16x8*>(Bb + pl * 16384 + nt * 1024);
                        const bf16x8 hi = *reinterpret_cast<const bf16x8*>(Bb + pl * 16384 + nt * 1024 + 4096);
                        for (int i = 0; i < 8; ++i) { b[pl][i] = lo[i]; b[pl][8 + i] = hi[i]; }
                    }
                for (int mt = 0; mt < 4; ++mt) for (int pl = 0; pl < 3; ++pl)
                    acc[mt][nt] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a[mt], b[pl], acc[mt][nt], idx, 0, 0);
            }
        }
        for (int a2 = 0; a2 < 4; ++a2) for (int b2 = 0; b2 < 4; ++b2) for (int r = 0; r < 4; ++r) s += acc[a2][b2][r];
    } else {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        bf16x8 a[2][2];
        bf16x16 b[3];
        for (int mt = 0; mt < 2; ++mt) for (int kh = 0; kh < 2; ++kh) a[mt][kh] = *reinterpret_cast<const bf16x8*>(Ab + (2 * mt + kh) * 1024);
        for (int pl = 0; pl < 3; ++pl) for (int i = 0; i < 16; ++i) b[pl][i] = (__bf16)(0.5f + 0.125f * ((lane + i + pl) & 7));
        for (int it = 0; it < iters; ++it) {
            if (LDS_FED) for (int mt = 0; mt < 2; ++mt) for (int kh = 0; kh < 2; ++kh) a[mt][kh] = *reinterpret_cast<const bf16x8*>(Ab + (2 * mt + kh) * 1024);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    if (LDS_FED)
                        for (int pl = 0; pl < 3; ++pl) {         // the same bytes per step as SHAPE 0: 2 x 16 B per (plane, N-tile, k half)
                            const bf16x8 lo = *reinterpret_cast<const bf16x8*>(Bb + pl * 16384 + (2 * nt + kh) * 1024);
                            const bf16x8 hi = *reinterpret_cast<const bf16x8*>(Bb + pl * 16384 + (2 * nt + kh) * 1024 + 4096);
                            for (int i = 0; i < 8; ++i) { b[pl][i] = lo[i]; b[pl][8 + i] = hi[i]; }
                        }
                    for (int mt = 0; mt < 2; ++mt) for (int pl = 0; pl < 3; ++pl)
                        acc[mt][nt] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a[mt][kh], b[pl], acc[mt][nt], idx, 0, 0);
                }
        }
        for (int a2 = 0; a2 < 2; ++a2) for (int b2 = 0; b2 < 2; ++b2) for (int r = 0; r < 16; ++r) s += acc[a2][b2][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

static double clock_ghz(unsigned long long* clk_dev, int n) {
    std::vector<unsigned long long> h(2 * n);
    (void)hipMemcpy(h.data(), clk_dev, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0;
    for (int i = 0; i < n; ++i) { cyc += (double)h[2 * i]; ticks += (double)h[2 * i + 1]; }
    return cyc / ticks * 0.1;
}

int main() {
    uint32_t* seed; float* out; unsigned long long* clk;
    CHECK(hipMalloc(&seed, 4096)); CHECK(hipMalloc(&out, 512 * 512 * 4)); CHECK(hipMalloc(&clk, 4096 * 8));
    uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i * 747796405u + 2891336453u;
    CHECK(hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const void* kerns[4] = {(const void*)loop_probe<0, 0>, (const void*)loop_probe<1, 0>, (const void*)loop_probe<0, 1>, (const void*)loop_probe<1, 1>};
    const char* names[4] = {"registers only, 16x16x64", "registers only, 32x32x32", "LDS-fed,        16x16x64", "LDS-fed,        32x32x32"};
    for (int k = 0; k < 4; ++k) CHECK(hipFuncSetAttribute(kerns[k], hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    const int iters = 1000, grid = 512, a_rate = 20;
    const double flop = (double)grid * 8 * iters * 64.0 * 64.0 * 64.0 * 2.0 * 3.0;
    for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < 4; ++k) {
            auto t0 = std::chrono::steady_clock::now();
            double last = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
                CHECK(hipEventRecord(e0));
                for (int i = 0; i < 20; ++i) {
                    void* args[] = {(void*)&seed, (void*)&out, (void*)&clk, (void*)&iters, (void*)&a_rate};
                    CHECK(hipLaunchKernel(kerns[k], dim3(grid), dim3(512), args, 98304, 0));
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                last = 20 * flop / (ms * 1e-3) / 1e12;
            }
            printf("structured-sparse bf16, 8 waves x (64 x 64 x 64 k per step), 3 weight planes, %s: %.0f TFLOP/s dense-equivalent sustained, in-kernel clock %.3f GHz\n",
                   names[k], last, clock_ghz(clk, grid));
        }
    return 0;
}
