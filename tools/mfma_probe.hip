// Micro-probe: what limits v_mfma_f32_32x32x2_f32 issue in the spike-GEMM inner loop on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: constant a,b, NACC accumulators     MODE 1: a from bfe+cvt per MFMA (b constant)
// MODE 2: MODE 1 + b streamed from global (4 x dwordx4 per 32 MFMAs, one chunk ahead, ping-pong)
template <int MODE, int NACC>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ w, const uint32_t* __restrict__ aw,
                                             float* out, int iters, int stagger) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint32_t word = aw[threadIdx.x];
    const f32x4* wp = reinterpret_cast<const f32x4*>(w) + lane + (size_t)wave * 256;
    f32x4 b0[4], b1[4];
    for (int qq = 0; qq < 4; ++qq) { b0[qq] = wp[qq * 64]; b1[qq] = b0[qq]; }
    if (stagger && wave >= 4) __builtin_amdgcn_s_sleep(16);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) { const f32x4* p = wp + (size_t)((it * 2 + 1) & 63) * 2048; for (int qq = 0; qq < 4; ++qq) b1[qq] = p[qq * 64]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                float a = (MODE == 0) ? b0[0][0] : (float)((word >> ((q + i) & 31)) & 1u);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[q >> 2][q & 3], acc[i], 0, 0, 0);
            }
        }
        if (MODE == 2) { const f32x4* p = wp + (size_t)((it * 2 + 2) & 63) * 2048; for (int qq = 0; qq < 4; ++qq) b0[qq] = p[qq * 64]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                float a = (MODE == 0) ? b1[0][0] : (float)((word >> ((q + i + 7) & 31)) & 1u);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[q >> 2][q & 3], acc[i], 0, 0, 0);
            }
        }
        if (MODE != 0) word = word * 1664525u + 1013904223u;
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MODE 3: A operands expanded one chunk ahead into ping-pong register arrays (no VALU write ever
// targets a register an in-flight MFMA reads); SHARE=1: both accumulators use the same a (NT=2 tile),
// SHARE=0: one a-array per accumulator (MT=2 tile).  b streamed from global as in MODE 2.
template <int SHARE>
__global__ __launch_bounds__(512) void probe3(const float* __restrict__ w, const uint32_t* __restrict__ aw,
                                              float* out, int iters, int stagger) {
    constexpr int NA = SHARE ? 1 : 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint32_t word = aw[threadIdx.x];
    const f32x4* wp = reinterpret_cast<const f32x4*>(w) + lane + (size_t)wave * 256;
    f32x4 b0[2][4], b1[2][4];
    float af0[NA][16], af1[NA][16];
    for (int n = 0; n < 2; ++n) for (int qq = 0; qq < 4; ++qq) { b0[n][qq] = wp[qq * 64]; b1[n][qq] = b0[n][qq]; }
    for (int n = 0; n < NA; ++n) for (int q = 0; q < 16; ++q) { af0[n][q] = (float)((word >> q) & 1u); af1[n][q] = af0[n][q]; }
    if (stagger && wave >= 4) __builtin_amdgcn_s_sleep(16);
    for (int it = 0; it < iters; ++it) {
        { const f32x4* p = wp + (size_t)((it * 2 + 1) & 31) * 2048;
          for (int n = 0; n < (SHARE ? 2 : 1); ++n) for (int qq = 0; qq < 4; ++qq) b1[n][qq] = p[n * 1024 + qq * 64]; }
        uint32_t w1 = word * 1664525u + 1013904223u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[0][q], b0[0][q >> 2][q & 3], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[NA - 1][q], b0[SHARE ? 1 : 0][q >> 2][q & 3], acc[1], 0, 0, 0);
            for (int n = 0; n < NA; ++n) af1[n][q] = (float)((w1 >> ((q + n) & 31)) & 1u);
        }
        { const f32x4* p = wp + (size_t)((it * 2 + 2) & 31) * 2048;
          for (int n = 0; n < (SHARE ? 2 : 1); ++n) for (int qq = 0; qq < 4; ++qq) b0[n][qq] = p[n * 1024 + qq * 64]; }
        word = w1 * 1664525u + 1013904223u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[0][q], b1[0][q >> 2][q & 3], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[NA - 1][q], b1[SHARE ? 1 : 0][q >> 2][q & 3], acc[1], 0, 0, 0);
            for (int n = 0; n < NA; ++n) af0[n][q] = (float)((word >> ((q + n) & 31)) & 1u);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHARE>
static void run3(const char* name, int threads, int blocks_per_cu, int stagger, const float* w, const uint32_t* aw, float* out) {
    const int iters = 2000;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe3<SHARE>), dim3(grid), dim3(threads), 0, 0, w, aw, out, 50, stagger);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe3<SHARE>), dim3(grid), dim3(threads), 0, 0, w, aw, out, iters, stagger);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * (threads / 64) * iters * 32.0 * 2;
    printf("%-44s thr=%d wg/cu=%d stag=%d  %.3f ms  %.1f TFLOP/s\n", name, threads, blocks_per_cu, stagger, ms, mfma * 4096 / (ms * 1e-3) / 1e12);
}

template <int MODE, int NACC>
static void run(const char* name, int threads, int blocks_per_cu, int stagger, const float* w, const uint32_t* aw, float* out) {
    const int iters = 2000;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, NACC>), dim3(grid), dim3(threads), 0, 0, w, aw, out, 50, stagger);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NACC>), dim3(grid), dim3(threads), 0, 0, w, aw, out, iters, stagger);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * (threads / 64) * iters * 32.0 * NACC;
    printf("%-44s thr=%d wg/cu=%d stag=%d  %.3f ms  %.1f TFLOP/s\n", name, threads, blocks_per_cu, stagger, ms, mfma * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    float* w; uint32_t* aw; float* out;
    hipMalloc(&w, 64 * 2048 * 16 + 8 * 4096 * 4); hipMalloc(&aw, 4096); hipMalloc(&out, 256 * 4 * 512 * 4);
    hipMemset(w, 0, 64 * 2048 * 16 + 8 * 4096 * 4); hipMemset(aw, 0x5a, 4096);
    run<0, 4>("const operands, 4 acc", 256, 1, 0, w, aw, out);
    run<0, 2>("const operands, 2 acc", 256, 1, 0, w, aw, out);
    run<0, 1>("const operands, 1 acc", 256, 1, 0, w, aw, out);
    run<0, 2>("const operands, 2 acc, 2 waves/SIMD", 512, 1, 0, w, aw, out);
    run<1, 2>("a=bfe+cvt, 2 acc, 1 wave/SIMD", 256, 1, 0, w, aw, out);
    run<1, 2>("a=bfe+cvt, 2 acc, 2 waves/SIMD", 512, 1, 0, w, aw, out);
    run<1, 2>("a=bfe+cvt, 2 acc, 2 waves/SIMD stagger", 512, 1, 1, w, aw, out);
    run<1, 4>("a=bfe+cvt, 4 acc, 1 wave/SIMD", 256, 1, 0, w, aw, out);
    run<1, 4>("a=bfe+cvt, 4 acc, 2 waves/SIMD", 512, 1, 0, w, aw, out);
    run<2, 2>("a=bfe+cvt, b=global, 2 acc, 2 waves/SIMD", 512, 1, 0, w, aw, out);
    run<2, 2>("a=bfe+cvt, b=global, 2 acc, 2 w/SIMD stagger", 512, 1, 1, w, aw, out);
    run<2, 4>("a=bfe+cvt, b=global, 4 acc, 1 wave/SIMD", 256, 1, 0, w, aw, out);
    run<2, 4>("a=bfe+cvt, b=global, 4 acc, 2 waves/SIMD", 512, 1, 0, w, aw, out);
    run<2, 2>("a=bfe+cvt, b=global, 2 acc, 4 waves/SIMD", 512, 2, 0, w, aw, out);
    run3<1>("pingpong a (shared, NT=2), b=global, 1 w/SIMD", 256, 1, 0, w, aw, out);
    run3<1>("pingpong a (shared, NT=2), b=global, 2 w/SIMD", 512, 1, 0, w, aw, out);
    run3<1>("pingpong a (shared, NT=2), b=global, 2 w/SIMD st", 512, 1, 1, w, aw, out);
    run3<0>("pingpong a (per acc, MT=2), b=global, 1 w/SIMD", 256, 1, 0, w, aw, out);
    run3<0>("pingpong a (per acc, MT=2), b=global, 2 w/SIMD", 512, 1, 0, w, aw, out);
    return 0;
}
