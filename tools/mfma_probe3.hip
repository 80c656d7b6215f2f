// probe 3: A operands as fp32 from LDS (ds_read_b128 per 4 MFMAs per M-tile), B from global (ping-pong)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, int BGLOBAL>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ w, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds = reinterpret_cast<float*>(smem);          // [100 positions][260] floats
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 100 * 260; i += blockDim.x) lds[i] = (float)((i * 2654435761u >> 13) & 1u);
    __syncthreads();
    f32x16 acc[MT];
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const int li = lane & 31, lh = lane >> 5;
    int aoff[MT];
    for (int m = 0; m < MT; ++m) aoff[m] = ((m * 4 + (li >> 3)) * 10 + (li & 7)) * 260 + 16 * lh;
    const f32x4* wp = reinterpret_cast<const f32x4*>(w) + lane + (size_t)wave * 256;
    f32x4 b0[4], b1[4];
    for (int qq = 0; qq < 4; ++qq) { b0[qq] = wp[qq * 64]; b1[qq] = b0[qq]; }
    for (int it = 0; it < iters; ++it) {
        const int c0 = (it & 3) * 64, tap = (it >> 2) % 9;
        const int toff = ((tap / 3) * 10 + tap % 3) * 260;
        if (BGLOBAL) { const f32x4* p = wp + (size_t)((it * 2 + 1) & 31) * 2048; for (int qq = 0; qq < 4; ++qq) b1[qq] = p[qq * 64]; }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            f32x4 a[MT];
            for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(lds + aoff[m] + toff + c0 + 4 * qq);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][r], b0[qq][r], acc[m], 0, 0, 0);
        }
        if (BGLOBAL) { const f32x4* p = wp + (size_t)((it * 2 + 2) & 31) * 2048; for (int qq = 0; qq < 4; ++qq) b0[qq] = p[qq * 64]; }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            f32x4 a[MT];
            for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4*>(lds + aoff[m] + toff + c0 + 32 + 4 * qq);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][r], b1[qq][r], acc[m], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MT, int BG>
static void run(const char* name, int threads, const float* w, float* out) {
    const int iters = 3000, grid = 256;
    auto k = probe<MT, BG>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 104000);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 104000, 0, w, out, 50);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 104000, 0, w, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * (threads / 64) * iters * 32.0 * MT;
    printf("%-52s thr=%d  %.3f ms  %.1f TFLOP/s\n", name, threads, ms, mfma * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    float* w; float* out;
    (void)hipMalloc(&w, 1 << 24); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemset(w, 0, 1 << 24);
    run<2, 0>("A=LDS fp32 b128, B const, MT=2, 1 w/SIMD", 256, w, out);
    run<2, 0>("A=LDS fp32 b128, B const, MT=2, 2 w/SIMD", 512, w, out);
    run<2, 1>("A=LDS fp32 b128, B global, MT=2, 1 w/SIMD", 256, w, out);
    run<2, 1>("A=LDS fp32 b128, B global, MT=2, 2 w/SIMD", 512, w, out);
    run<4, 1>("A=LDS fp32 b128, B global, MT=4, 1 w/SIMD", 256, w, out);
    run<4, 1>("A=LDS fp32 b128, B global, MT=4, 2 w/SIMD", 512, w, out);
    run<1, 1>("A=LDS fp32 b128, B global, MT=1, 2 w/SIMD", 512, w, out);
    return 0;
}
