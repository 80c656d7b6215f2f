// probe 4: bf16 MFMA shape under an LDS-fed loop with this kernel's traffic (16 ds_read_b128 per 32-deep chunk of a
// 64x64 wave tile, 3 weight planes), random data: 32x32x16 vs 16x16x32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(const uint32_t* __restrict__ seed, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* l32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = threadIdx.x; i < 81920 / 4; i += blockDim.x) {
        uint32_t x = seed[i & 1023] * 2654435761u + i * 40503u;
        // A region (first 32 KB): bf16 0/1 spikes; B region: random bf16 weights in [-2,2)
        l32[i] = (i < 8192) ? ((x & 1 ? 0x3F80u : 0u) | (x & 2 ? 0x3F800000u : 0u)) : ((x & 0x807F807Fu) | 0x3F003F00u);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        const int li = lane & 31, lh = lane >> 5;
        const unsigned char* Ab = smem + (wm * 64 + li) * 64 + ((lh ^ ((li >> 2) & 3)) << 4);
        const unsigned char* Bb = smem + 32768 + (wn * 64 + li) * 64 + ((lh ^ ((li >> 2) & 3)) << 4);
        for (int it = 0; it < iters; ++it) {
            const int boff = (it & 1) * 16384 * 0;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 a[2], b[2][3];
                for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(Ab + mt * 2048 + (32 * st ^ 0) + boff);
                for (int nt = 0; nt < 2; ++nt) for (int pl = 0; pl < 3; ++pl)
                    b[nt][pl] = *reinterpret_cast<const bf16x8*>(Bb + pl * 8192 + nt * 2048 + 32 * st);
                for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) for (int pl = 0; pl < 3; ++pl)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt][pl], acc[mt][nt], 0, 0, 0);
            }
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    } else {
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
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[pl], acc[mt][nt], 0, 0, 0);
            }
        }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE>
static void run(const char* name, int wg_per_cu, const uint32_t* seed, float* out) {
    const int iters = 20000, grid = 256 * wg_per_cu;
    auto k = probe<SHAPE>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 81920, 0, seed, out, 200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 81920, 0, seed, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 8 * iters * 64.0 * 64.0 * 32.0 * 2.0 * 3.0;
    printf("%-40s wg/cu=%d  %.2f ms  %.0f TFLOP/s executed\n", name, wg_per_cu, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
    uint32_t* seed; float* out;
    (void)hipMalloc(&seed, 4096); (void)hipMalloc(&out, 512 * 512 * 4);
    uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i * 747796405u + 2891336453u;
    (void)hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
    run<32>("32x32x16, LDS-fed, 64x64 wave tile", 1, seed, out);
    run<32>("32x32x16, LDS-fed, 64x64 wave tile", 2, seed, out);
    run<16>("16x16x32, LDS-fed, 64x64 wave tile", 1, seed, out);
    run<16>("16x16x32, LDS-fed, 64x64 wave tile", 2, seed, out);
    return 0;
}
