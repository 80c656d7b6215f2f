// FETCH_SIZE calibration for the conv kernel's spike-word access pattern (MI355X_MICROARCH.md, HBM: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels read the SAME 64 MiB buffer once:
//   k_wide   : 16 B per lane, coalesced streaming read (the guide's reference case: FETCH_SIZE reports half the bytes)
//   k_gather : the conv's pattern - a wave's 64 lanes take one 4-byte word each of 64 consecutive 32-byte rows (global_load_lds_dword
//              into LDS, like stage_a), word after word: every byte of the buffer is requested exactly once
//   k_gather8: the same with rows of 8 words read by eight different waves of the work-group (one word each)
// Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  and compare the counter with 64 MiB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define BYTES (64u << 20)
__global__ __launch_bounds__(256) void k_wide(const uint4* __restrict__ p, uint32_t* out, size_t n16) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(512) void k_gather(const uint32_t* __restrict__ p, uint32_t* out, size_t rows, int split) {
    __shared__ uint32_t lds[512 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds;
    // work-group = 512 consecutive rows; wave w stages rows 64 w .. 64 w + 63 (split == 0: all 8 words; split == 1: wave w takes word w of ALL 512 rows)
    for (size_t r0 = (size_t)blockIdx.x * 512; r0 < rows; r0 += (size_t)gridDim.x * 512) {
        for (int j = 0; j < 8; ++j) {
            const size_t row = split ? r0 + 64 * j + lane : r0 + 64 * wave + lane;
            const int word = split ? wave : j;
            const uint32_t voff = (uint32_t)((row * 8 + word) * 4 - (size_t)0);
            const uint32_t d = __builtin_amdgcn_readfirstlane(lds_base + (wave * 8 + j) * 256);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(p), "s"(d) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (lds[threadIdx.x] == 0x12345678u && out) out[1] = 1;
}
int main() {
    uint32_t *buf, *out;
    (void)hipMalloc(&buf, BYTES); (void)hipMalloc(&out, 64);
    (void)hipMemset(buf, 1, BYTES);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_wide, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, out, (size_t)BYTES / 16);
        hipLaunchKernelGGL(k_gather, dim3(2048), dim3(512), 0, 0, buf, out, (size_t)BYTES / 32, 0);
        hipLaunchKernelGGL(k_gather, dim3(2048), dim3(512), 0, 0, buf, out, (size_t)BYTES / 32, 1);
        (void)hipDeviceSynchronize();
    }
    printf("done: each kernel read %u bytes once\n", BYTES);
    return 0;
}
