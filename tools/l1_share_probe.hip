// Does the per-CU L1 (TCP) serve the second of two co-resident work-groups that stream the SAME bytes into LDS by LDS-DMA
// (global_load_lds_dwordx4) at about the same time?  Two work-groups per CU, each copies the same 1.77-MB panel chunk by chunk
// (24 KB per chunk, 8 waves x 3 x 1 KB) with a barrier per chunk and a pacing loop between chunks (the real kernel spends
// ~1.5 us per chunk).  `skew` delays every second work-group by that many chunks.  Read with
//   rocprofv3 --kernel-trace --pmc TCC_READ_sum TCC_REQ_sum -- tools/_ab/l1_share_probe
// no sharing: requests = work-groups x bytes / 128.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ const void* sgpr_ptr(const void* p) {
    const unsigned long long x = (unsigned long long)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}

template <int SKEW>
__global__ __launch_bounds__(512) void stream(const unsigned char* __restrict__ panel, int n_chunks, int pace, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // every second work-group of the launch (co-residents are b and b + 256 on this chip's dispatch order) starts SKEW chunks in
    const int start = ((blockIdx.x >> 8) & 1) ? SKEW : 0;
    float acc = 0.f;
    for (int c = 0; c < n_chunks; ++c) {
        const int cc = (c + start) % n_chunks;
        const unsigned char* src = panel + (size_t)cc * 24576 + wave * 3072;
        for (int j = 0; j < 3; ++j) {
            const uint32_t d = __builtin_amdgcn_readfirstlane(base + (c & 1) * 24576 + wave * 3072 + j * 1024);
            const void* p = sgpr_ptr(src + j * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((uint32_t)(lane * 16)), "s"(p), "s"(d) : "memory", "m0");
        }
        for (int i = 0; i < pace; ++i) acc = acc * 1.0001f + 0.5f;          // pacing (dependent VALU chain)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += reinterpret_cast<const float*>(smem)[((c & 1) * 24576 + threadIdx.x * 4) / 4];
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
    const int n_chunks = 72, grid = 512 * 4;                 // 4 rounds of 2 work-groups per CU
    unsigned char* panel; float* out;
    (void)hipMalloc(&panel, (size_t)n_chunks * 24576); (void)hipMalloc(&out, (size_t)grid * 512 * 4);
    (void)hipMemset(panel, 1, (size_t)n_chunks * 24576);
    (void)hipFuncSetAttribute((const void*)stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)stream<36>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (int pace : {0, 400, 1500}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(stream<0>, dim3(grid), dim3(512), 80 * 1024, 0, panel, n_chunks, pace, out);
            hipLaunchKernelGGL(stream<36>, dim3(grid), dim3(512), 80 * 1024, 0, panel, n_chunks, pace, out);
        }
    }
    (void)hipDeviceSynchronize();
    printf("no sharing would be %.4g requests of 128 B per launch (%d work-groups x %d bytes)\n", (double)grid * n_chunks * 24576 / 128.0, grid, n_chunks * 24576);
    return 0;
}
