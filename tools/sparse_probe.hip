// Round-4 go / no-go probes for exploiting the SPARSITY of the encoder's period planes e_n, n >= 3 (densities 0.039, 0.019, 0.011,
// 0.007, 0.005 on the bench's pyramid; 5 of the conv tile's 7 row groups, each paying a full dense MFMA pass today):
//
//  (A) structured-sparse matrix cores: v_smfmac_f32_16x16x64_bf16 takes an A operand with at most 2 non-zeros per 4 consecutive k
//      (8 bf16 values + 2-bit positions per lane for a 16-deep k span) and multiplies 64 k per instruction.
//        A1  operand layout, found empirically (one-hot A value x coded B): which (row, k) a lane's value slot + index names,
//            which (k, column) a lane's B element is;
//        A2  instruction rate: register-resident loop, dense 16x16x32 against sparse 16x16x64 (dense-equivalent FLOP);
//        A3  sustained rate + in-kernel clock in the production kernel's LDS-fed shape with the workload's operand statistics.
//  (B) event-driven accumulation on the vector ALUs: u[pos][col] += W[k][col] for every set bit k of a position's spike word, W chunk
//      (32 k x 64 columns fp32) resident in LDS, lane = column, one wave per position group - cost per (position, chunk) against the
//      13 CU-cycles the MFMA loop spends on the same 5 sparse row groups.
//
//  hipcc --offload-arch=gfx950 -O3 -o sparse_probe tools/sparse_probe.hip && ./sparse_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <chrono>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ __bf16 bf(float f) { return (__bf16)f; }

// ---- A1: layout discovery.  Experiment e = (lane l, slot j, index q): A = one-hot (value 1.0 in slot j of lane l, all indices = q);
// B element i of lane l' = code.  mode 0: code = i + 1; mode 1: code = l' + 1.  out[e][lane][r] = accumulator.
__global__ void layout_probe(float* out, int mode) {
    const int lane = threadIdx.x;
    const int e = blockIdx.x, l = e >> 5, j = (e >> 2) & 7, q = e & 3;
    bf16x8 a;
    for (int s = 0; s < 8; ++s) a[s] = bf((lane == l && s == j) ? 1.0f : 0.0f);
    bf16x16 b;
    for (int i = 0; i < 16; ++i) b[i] = bf(mode == 0 ? (float)(i + 1) : (float)(lane + 1));
    int idx = 0;
    for (int s = 0; s < 8; ++s) idx |= q << (2 * s);
    idx |= idx << 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, acc, idx, 0, 0);
    for (int r = 0; r < 4; ++r) out[((size_t)e * 64 + lane) * 4 + r] = acc[r];
}

// ---- A4: functional check of the nibble encoding the kernels would use.  A row's 64 spike bits (<= 2 set per nibble) -> per 16-k block
// 8 value slots (1.0 / 0) + 8 two-bit positions: nibble bits p0 < p1 -> slots (1.0, p0), (1.0, p1); one bit p: p < 3 -> (1.0, p), (0, 3),
// p == 3 -> (0, 0), (1.0, 3); none -> (0, 0), (0, 3) - always index0 < index1.  B = small integers: the result must equal the host's
// dot products exactly.  idx_hi: garbage in bits 31:16 of the index register (are they ignored at ABID = 0?).
__global__ void func_probe(const unsigned long long* __restrict__ rows, const float* __restrict__ bmat, float* out, int idx_hi) {
    const int lane = threadIdx.x, lr = lane & 15, lg = lane >> 4;
    const unsigned long long bits = rows[blockIdx.x * 16 + lr];
    const uint32_t h = (uint32_t)(bits >> (16 * lg)) & 0xffffu;
    bf16x8 a;
    int idx = 0;
    for (int nb = 0; nb < 4; ++nb) {
        const uint32_t x = (h >> (4 * nb)) & 15u;
        int p0 = -1, p1 = -1;
        for (int b = 0; b < 4; ++b) if (x & (1u << b)) { if (p0 < 0) p0 = b; else if (p1 < 0) p1 = b; }
        float v0, v1; int i0, i1;
        if (p1 >= 0) { v0 = 1.f; i0 = p0; v1 = 1.f; i1 = p1; }
        else if (p0 >= 0 && p0 < 3) { v0 = 1.f; i0 = p0; v1 = 0.f; i1 = 3; }
        else if (p0 == 3) { v0 = 0.f; i0 = 0; v1 = 1.f; i1 = 3; }
        else { v0 = 0.f; i0 = 0; v1 = 0.f; i1 = 3; }
        a[2 * nb] = bf(v0); a[2 * nb + 1] = bf(v1);
        idx |= (i0 | (i1 << 2)) << (4 * nb);
    }
    if (idx_hi) idx |= 0x5a5a0000;
    bf16x16 b;                                                   // element i of lane (g, col): k = 8 g + i (i < 8), 32 + 8 g + i - 8
    for (int i = 0; i < 16; ++i) {
        const int k = i < 8 ? 8 * lg + i : 32 + 8 * lg + (i - 8);
        b[i] = bf(bmat[k * 16 + lr]);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, acc, idx, 0, 0);
    for (int r = 0; r < 4; ++r) out[(blockIdx.x * 16 + 4 * lg + r) * 16 + lr] = acc[r];       // row 4 lg + r, column lr
}

// ---- A2: instruction rate, everything in registers
template <int SPARSE>
__global__ __launch_bounds__(256) void rate_probe(float* out, int iters) {
    f32x4 acc[8];
    for (int a = 0; a < 8; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a8;
    bf16x16 b16;
    for (int i = 0; i < 8; ++i) a8[i] = bf((float)((threadIdx.x + i) & 1));
    for (int i = 0; i < 16; ++i) b16[i] = bf(0.5f + (float)((threadIdx.x * 3 + i) & 7) * 0.125f);
    s16x8 a8s, b8s;
    memcpy(&a8s, &a8, 16); memcpy(&b8s, &b16, 16);
    const int idx = 0x44444444 ^ 0;                              // slot pairs (0, 1): positions 0 and 1 ... (01 00 pattern = 0x4 per pair)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (SPARSE) acc[a] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a8, b16, acc[a], idx, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8s, b8s, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- A3: the production loop's shape (8 waves, 64 x 64 per wave, three weight planes, fragments from static LDS contents), dense
// against sparse, with the workload's statistics: A values 1.0 with probability a_rate / 256 (dense) - the sparse form holds the same
// spikes compressed (2 slots per 4 k: density d -> slot occupancy 2d), B random bf16.  Reports executed dense-equivalent TFLOP/s and clock.
template <int SPARSE>
__global__ __launch_bounds__(512) void lds_probe(const uint32_t* __restrict__ seed, float* out, unsigned long long* clk, int iters, int a_rate_256) {
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
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    const int lr = lane & 15, lg = lane >> 4;
    const unsigned char* Ab = smem + (wave * 64 + lr) * 64 + (lg << 4);
    const unsigned char* Bb = smem + 32768 + lr * 64 + (lg << 4);
    const int idx = 0x44444444;
    for (int it = 0; it < iters; ++it) {
        if (SPARSE) {
            // one "chunk" = 64 k: A = 8 compressed values + indices per M-tile, B = 16 values per plane and N-tile (two 16-byte reads)
            bf16x8 a[4];
            for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(Ab + mt * 1024);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                bf16x16 b[3];
                for (int pl = 0; pl < 3; ++pl) {
                    const bf16x8 lo = *reinterpret_cast<const bf16x8*>(Bb + pl * 16384 + nt * 1024);
                    const bf16x8 hi = *reinterpret_cast<const bf16x8*>(Bb + pl * 16384 + nt * 1024 + 4096);
                    for (int i = 0; i < 8; ++i) { b[pl][i] = lo[i]; b[pl][8 + i] = hi[i]; }
                }
                for (int mt = 0; mt < 4; ++mt) for (int pl = 0; pl < 3; ++pl)
                    acc[mt][nt] = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a[mt], b[pl], acc[mt][nt], idx, 0, 0);
            }
        } else {
            for (int half = 0; half < 2; ++half) {              // two 32-deep chunks = the same 64 k
                s16x8 a[4];
                for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const s16x8*>(Ab + mt * 1024 + half * 16384);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    s16x8 b[3];
                    for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const s16x8*>(Bb + pl * 16384 + nt * 1024 + half * 4096);
                    for (int mt = 0; mt < 4; ++mt) for (int pl = 0; pl < 3; ++pl)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[pl], acc[mt][nt], 0, 0, 0);
                }
            }
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

// ---- B: event-driven accumulation.  Work-group = 8 waves; LDS holds one W chunk [32 k][64 columns] fp32 (8 KB, static) and the
// spike words of the tile: words[chunk % 64][plane 0..4][position].  Wave w owns positions w, w + 8, ... (PPW of them), lane = column;
// accumulators acc[p][plane] in registers (PPW x 5).  Per chunk and (position, plane): read the word (scalar, via readfirstlane of a
// broadcast LDS read), then for every set bit one ds_read_b32 of W[k][lane] + one v_add_f32.
template <int PPW>
__global__ __launch_bounds__(512) void event_probe(const uint32_t* __restrict__ seed, float* out, unsigned long long* clk, int chunks,
                                                   const int d0, const int d1, const int d2, const int d3, const int d4) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* W = reinterpret_cast<float*>(smem);                                  // [32][64]
    uint32_t* words = reinterpret_cast<uint32_t*>(smem + 8192);                 // [64 chunk slots][5][8 * PPW]
    const int P = 8 * PPW;
    const int dens[5] = {d0, d1, d2, d3, d4};                                   // densities in 1 / 65536
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) W[i] = (float)((seed[i & 1023] >> 9) & 0xffff) * 1e-5f;
    for (int i = threadIdx.x; i < 64 * 5 * P; i += blockDim.x) {
        const int plane = (i / P) % 5;
        uint32_t w = 0;
        for (int b = 0; b < 32; ++b) {
            uint32_t x = seed[(i + b * 131) & 1023] * 2654435761u + i * 40503u + b * 977u + blockIdx.x * 7919u;
            x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
            if ((x & 0xffff) < (uint32_t)dens[plane]) w |= 1u << b;
        }
        words[i] = w;
    }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc[PPW][5];
    for (int p = 0; p < PPW; ++p) for (int n = 0; n < 5; ++n) acc[p][n] = 0.f;
    const float* Wl = W + lane;
    for (int c = 0; c < chunks; ++c) {
        const uint32_t* wc = words + (size_t)(c & 63) * 5 * P;
#pragma unroll
        for (int p = 0; p < PPW; ++p) {
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                uint32_t m = __builtin_amdgcn_readfirstlane(wc[n * P + wave + 8 * p]);
                while (m) {                                                     // wave-uniform loop over the set bits
                    const int k = __builtin_ctz(m);
                    m &= m - 1;
                    acc[p][n] += Wl[k * 64];
                }
            }
        }
    }
    float s = 0.f;
    for (int p = 0; p < PPW; ++p) for (int n = 0; n < 5; ++n) s += acc[p][n];
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

int main(int argc, char** argv) {
    uint32_t* seed; float* out; unsigned long long* clk;
    CHECK(hipMalloc(&seed, 4096)); CHECK(hipMalloc(&out, 2048 * 64 * 4 * 4)); CHECK(hipMalloc(&clk, 4096 * 8));
    uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i * 747796405u + 2891336453u;
    CHECK(hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));

    // ---- A1 layout
    {
        std::vector<float> r0(2048 * 64 * 4), r1(2048 * 64 * 4);
        hipLaunchKernelGGL(layout_probe, dim3(2048), dim3(64), 0, 0, out, 0);
        CHECK(hipMemcpy(r0.data(), out, r0.size() * 4, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(layout_probe, dim3(2048), dim3(64), 0, 0, out, 1);
        CHECK(hipMemcpy(r1.data(), out, r1.size() * 4, hipMemcpyDeviceToHost));
        // accumulator layout of the 16x16 result: lane -> column lane & 15, rows 4 * (lane >> 4) + r
        printf("A1: v_smfmac_f32_16x16x64_bf16 operand layout (experiment = one-hot A value in (lane l, slot j), every index = q)\n");
        int bad = 0;
        for (int e = 0; e < 2048; ++e) {
            const int l = e >> 5, j = (e >> 2) & 7, q = e & 3;
            int row = -1, bi = -1, bl[16];
            for (int c = 0; c < 16; ++c) bl[c] = -1;
            for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
                const float v0 = r0[((size_t)e * 64 + lane) * 4 + r], v1 = r1[((size_t)e * 64 + lane) * 4 + r];
                if (v0 != 0.f) { row = 4 * (lane >> 4) + r; bi = (int)v0 - 1; bl[lane & 15] = (int)v1 - 1; }
            }
            // hypothesis: row = l & 15; the value names k = 16 * (l >> 4) + 4 * (j >> 1) + q ... the B element holding that k is element
            // (k % 16) of lane 16 * (k / 16) + column
            const int k_h = 16 * (l >> 4) + 4 * (j >> 1) + q;
            // (B: element i < 8 of lane group g holds k = 8 g + i; elements 8 .. 15 hold k = 32 + 8 g + i - 8: two stacked 16x16x32 layouts)
            const int bg = (k_h & 31) >> 3, be = (k_h & 7) + 8 * (k_h >> 5);
            const bool ok = row == (l & 15) && bi == be && bl[0] == 16 * bg;
            if (!ok) ++bad;
            if (e < 8 || (!ok && bad < 24)) printf("  lane %2d slot %d idx %d -> row %2d, B element %2d of lanes %2d.. (hypothesis k = %2d: %s)\n", l, j, q, row, bi, bl[0], k_h, ok ? "ok" : "NO");
        }
        printf("A1: %d of 2048 experiments off the hypothesis [A: row = l & 15, k = 16 (l >> 4) + 4 (j >> 1) + q;  B: element (k & 7) + 8 (k >> 5) of lane 16 ((k & 31) >> 3) + column]\n", bad);
    }
    // ---- A4 functional check
    {
        const int NT = 256;
        std::vector<unsigned long long> rows(NT * 16);
        std::vector<float> bm(64 * 16);
        uint32_t x = 12345u;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
        for (auto& r : rows) {
            r = 0;
            for (int nb = 0; nb < 16; ++nb) {
                const uint32_t c = rnd() % 11;                                  // 0 .. 10: the 11 nibbles with <= 2 bits
                static const uint32_t pat[11] = {0, 1, 2, 4, 8, 3, 5, 9, 6, 10, 12};
                r |= (unsigned long long)pat[c] << (4 * nb);
            }
        }
        for (auto& v : bm) v = (float)((int)(rnd() % 17) - 8);
        unsigned long long* d_rows; float* d_b;
        CHECK(hipMalloc(&d_rows, rows.size() * 8)); CHECK(hipMalloc(&d_b, bm.size() * 4));
        CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_b, bm.data(), bm.size() * 4, hipMemcpyHostToDevice));
        for (int hi = 0; hi < 2; ++hi) {
            hipLaunchKernelGGL(func_probe, dim3(NT), dim3(64), 0, 0, d_rows, d_b, out, hi);
            std::vector<float> got(NT * 16 * 16);
            CHECK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int t = 0; t < NT * 16; ++t) for (int c = 0; c < 16; ++c) {
                float e = 0.f;
                for (int k = 0; k < 64; ++k) if ((rows[t] >> k) & 1ull) e += bm[k * 16 + c];
                if (got[t * 16 + c] != e) ++bad;
            }
            printf("A4: nibble encoding on %d random 2:4-compliant rows x 16 columns, index bits 31:16 %s: %d wrong results\n", NT * 16, hi ? "garbage" : "zero", bad);
        }
    }
    // ---- A2 rate
    for (int sp = 0; sp < 2; ++sp) {
        const int iters = 4000, grid = 2048;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            if (sp) hipLaunchKernelGGL(rate_probe<1>, dim3(grid), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(rate_probe<0>, dim3(grid), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = (double)grid * 4 * iters * 8 * 16.0 * 16.0 * (sp ? 64.0 : 32.0) * 2.0;
            if (rep == 2) printf("A2: register loop, %s: %.0f TFLOP/s dense-equivalent (%.3f ms)\n", sp ? "v_smfmac_f32_16x16x64_bf16" : "v_mfma_f32_16x16x32_bf16  ", flop / (ms * 1e-3) / 1e12, ms);
        }
    }
    // ---- A3 LDS-fed loop, sustained
    for (int pass = 0; pass < 2; ++pass)
    for (int sp = 0; sp < 2; ++sp) {
        const int iters = 1000, grid = 512;
        const void* kern = sp ? (const void*)lds_probe<1> : (const void*)lds_probe<0>;
        CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        // dense operand at the workload's plane density (e_1 .. e_2: ~0.18); the sparse form's slots are occupied twice as often per slot
        const int a_rate = sp ? 20 : 10;       // of 256: sparse planes e_3.. have density ~0.04 -> compressed slot occupancy ~0.08
        const double flop = (double)grid * 8 * iters * 64.0 * 64.0 * 64.0 * 2.0 * 3.0;
        auto t0 = std::chrono::steady_clock::now();
        double last = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) {
                if (sp) hipLaunchKernelGGL(lds_probe<1>, dim3(grid), dim3(512), 98304, 0, seed, out, clk, iters, a_rate);
                else hipLaunchKernelGGL(lds_probe<0>, dim3(grid), dim3(512), 98304, 0, seed, out, clk, iters, a_rate);
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            last = 20 * flop / (ms * 1e-3) / 1e12;
        }
        printf("A3: LDS-fed 8-wave loop, 3 weight planes, %s: %.0f TFLOP/s dense-equivalent sustained, in-kernel clock %.3f GHz\n",
               sp ? "sparse 16x16x64" : "dense 16x16x32 ", last, clock_ghz(clk, grid));
    }
    // ---- B event-driven accumulation
    {
        const int dens[5] = {(int)(0.0386 * 65536), (int)(0.0188 * 65536), (int)(0.0111 * 65536), (int)(0.0073 * 65536), (int)(0.0051 * 65536)};
        const int chunks = 72 * 40, grid = 512;
        const size_t lds = 8192 + 64 * 5 * 8 * 9 * 4;
        for (int pass = 0; pass < 2; ++pass) {
            const void* kern = (const void*)event_probe<9>;
            CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(event_probe<9>, dim3(grid), dim3(512), lds, 0, seed, out, clk, chunks, dens[0], dens[1], dens[2], dens[3], dens[4]);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            // units = (position, chunk) pairs of one 64-column block: grid x 72 positions x chunks; two work-groups share a CU (grid = 2 x 256)
            const double units = (double)grid * 72 * chunks;
            const double ghz = clock_ghz(clk, grid);
            printf("B: event-driven accumulate, planes e_3..e_7 (densities .0386 .0188 .0111 .0073 .0051), 72 positions x 64 columns per work-group, 2 work-groups per CU:\n"
                   "   %.3f ms for %.3g (position, chunk) units -> %.2f CU-cycles per unit at %.3f GHz (the MFMA loop spends ~13 CU-cycles per unit on these 5 row groups)\n",
                   best, units, best * 1e-3 * ghz * 1e9 * 256.0 / units, ghz);
        }
    }
    return 0;
}
