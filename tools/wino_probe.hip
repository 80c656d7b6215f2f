// probe 6: the inner structure a Winograd F(2x2,3x3) form of the shared conv would have, on static LDS contents (no global
// traffic): 16 transformed coordinates xi, 8 chunks of 32 channels each; per (xi, chunk) the work-group (8 waves = 2 row-waves
// x 4 column-waves, wave tile = 64 tile-rows x 32 columns = 4 x 2 MFMA tiles, 3 weight planes) builds the transformed-spike
// fragments U = +-T[b1] +- T[b2] +- T[b3] +- T[b4] (byte -> 8 bf16 table lookups + packed bf16 adds; the 4 column-waves share
// the 4 M-tiles' fragments through LDS), then 4 A-fragment reads + 6 B-fragment reads feed 24 MFMAs into M; after the 8 chunks M
// is folded into the four output accumulators Y (+-M).  Prints the sustained executed MFMA rate against mfma_probe5's
// direct-loop rate - the Winograd form needs 2.25x fewer MFMAs, so it wins if it sustains more than 1 / 2.25 of that.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

#define LUT_OFF 0              // 4 KB: byte -> 8 bf16
#define BYTES_OFF 4096         // 8 KB: spike bytes of the 4x4 patches: [128 rows][16 pixels][4 bytes]
#define U_OFF (4096 + 8192)    // 2 x 8 KB: U image [buf][4 m-tiles... rows 128][64 B]
#define V_OFF (U_OFF + 16384)  // 3 slots x 24 KB: [slot][plane 3][128 cols][64 B]
#define LDS_BYTES (V_OFF + 3 * 24576)

__device__ __forceinline__ bf16x8 pk_addsub(bf16x8 a, bf16x8 b, bool sub) {
    // 4 packed bf16 adds (v_pk_add_bf16 on gfx950); the sign rides on a sign-bit flip of b
    union { bf16x8 v; uint32_t u[4]; bf16x2v h[4]; } x, y, r;
    x.v = a; y.v = b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (sub) y.u[i] ^= 0x80008000u;
        r.h[i] = x.h[i] + y.h[i];
    }
    return r.v;
}

// BUILD == 3: the same structure with (a) the fragment built in INTEGER arithmetic - byte -> 8 nibbles table (4 B), nibble-wise
// sum of the four pixels with a bias of 2 (one 32-bit add/sub each), nibble-pair -> 2 bf16 table (4 B, 4 reads) - because gfx950 has
// no packed bf16 add (hipcc turns one into shift / v_add_f32 / v_cvt_pk_bf16_f32: ~50 vector instructions per fragment), and
// (b) software pipelining: the fragment of step s+1 is built and stored while the MFMAs of step s run (U image double-buffered).
#define NIB_OFF LDS_BYTES            // 1 KB: byte -> 8 nibbles (0 / 1)
#define CODE_OFF (LDS_BYTES + 1024)  // 1 KB: two nibble codes (0..4 each) -> 2 bf16 of (code - 2)
#define LDS_BYTES3 (LDS_BYTES + 2048)
__global__ __launch_bounds__(512, 2) void probe3(const uint32_t* __restrict__ seed, float* out, unsigned long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* l32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = threadIdx.x; i < LDS_BYTES3 / 4; i += blockDim.x) {
        uint32_t x = seed[i & 1023] * 2654435761u + i * 40503u + blockIdx.x * 977u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        if (i >= NIB_OFF / 4 && i < CODE_OFF / 4) { const int e = i - NIB_OFF / 4; uint32_t v = 0; for (int b = 0; b < 8; ++b) v |= ((e >> b) & 1u) << (4 * b); l32[i] = v; }
        else if (i >= CODE_OFF / 4) {
            const int e = i - CODE_OFF / 4, c0 = min(e & 15, 4), c1 = min(e >> 4, 4);
            const uint32_t t[5] = {0xC000u, 0xBF80u, 0u, 0x3F80u, 0x4000u};
            l32[i] = t[c0] | (t[c1] << 16);
        }
        else if (i >= BYTES_OFF / 4 && i < (BYTES_OFF + 8192) / 4) l32[i] = x & (x >> 8) & 0x7F7F7F7Fu;
        else if (i < V_OFF / 4) l32[i] = ((x & 1) ? 0x3F80u : 0u) | ((x & 2) ? 0xBF800000u : 0u);
        else l32[i] = (x & 0x807F807Fu) | 0x3F003F00u;
    }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;
    f32x4 M[4][2], Y[4][4][2];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) { M[a][b] = f32x4{0, 0, 0, 0}; for (int o = 0; o < 4; ++o) Y[o][a][b] = f32x4{0, 0, 0, 0}; }
    const int brow = wm * 64 + wn * 16 + lr;
    const unsigned char* bytes = smem + BYTES_OFF + brow * 64 + lg;
    unsigned char* const uimg = smem + U_OFF;
    const unsigned char* Ab = uimg + (wm * 64 + lr) * 64 + (lg << 4);
    const unsigned char* Bb = smem + V_OFF + (wn * 32 + lr) * 64 + (lg << 4);
    const uint32_t* nib = reinterpret_cast<const uint32_t*>(smem + NIB_OFF);
    const uint32_t* code = reinterpret_cast<const uint32_t*>(smem + CODE_OFF);
    auto build = [&](int step) {                         // fragment of (xi, chunk) = step -> U image buffer step & 1
        const int xi = (step >> 3) & 15, ch = step & 7, i = xi >> 2, j = xi & 3;
        const int r1 = i == 0 ? 0 : 1, r2 = i == 3 ? 3 : 2;
        const int c1 = j == 0 ? 0 : 1, c2 = j == 3 ? 3 : 2;
        const uint32_t w11 = nib[bytes[(r1 * 4 + c1) * 4 + (ch & 3)]], w12 = nib[bytes[(r1 * 4 + c2) * 4 + (ch & 3)]];
        const uint32_t w21 = nib[bytes[(r2 * 4 + c1) * 4 + (ch & 3)]], w22 = nib[bytes[(r2 * 4 + c2) * 4 + (ch & 3)]];
        const bool sr = i != 1, sc = j != 1;             // (wave-uniform signs)
        uint32_t s = 0x22222222u + w11;
        s = sc ? s - w12 : s + w12;
        s = sr ? s - w21 : s + w21;
        s = (sr != sc) ? s - w22 : s + w22;
        uint4 u;
        u.x = code[s & 255]; u.y = code[(s >> 8) & 255]; u.z = code[(s >> 16) & 255]; u.w = code[s >> 24];
        *reinterpret_cast<uint4*>(uimg + (step & 1) * 8192 + brow * 64 + (lg << 4)) = u;
    };
    const int n_steps = iters * 128;
    build(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
#pragma unroll 2
    for (int step = 0; step < n_steps; ++step) {
        const int buf = step & 1, slot = step % 3;
        bf16x8 a[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(Ab + buf * 8192 + mt * 1024);
        build(step + 1);                                 // next step's fragment, under this step's MFMAs
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bf16x8 b[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(Bb + slot * 24576 + pl * 8192 + nt * 1024);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    M[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[pl], M[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if ((step & 7) == 7) {                           // end of a transformed coordinate: fold M into the outputs
            const int xi = (step >> 3) & 15, i = xi >> 2, j = xi & 3;
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const int sa = a2 == 0 ? (i < 3 ? 1 : 0) : (i == 0 ? 0 : i == 1 ? 1 : -1);
                    const int sb = b2 == 0 ? (j < 3 ? 1 : 0) : (j == 0 ? 0 : j == 1 ? 1 : -1);
                    const int sg = sa * sb;
                    if (sg != 0) {
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                Y[a2 * 2 + b2][mt][nt] = sg > 0 ? Y[a2 * 2 + b2][mt][nt] + M[mt][nt] : Y[a2 * 2 + b2][mt][nt] - M[mt][nt];
                    }
                }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) M[mt][nt] = f32x4{0, 0, 0, 0};
        }
    }
    float sacc = 0.f;
    for (int o = 0; o < 4; ++o) for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) sacc += Y[o][a][b][r] + M[a][b][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
    if (threadIdx.x == 0) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int BUILD>   // 0: no U build (A fragments read from a static image), 1: build shared through LDS, 2: no Y fold either
__global__ __launch_bounds__(512, 2) void probe(const uint32_t* __restrict__ seed, float* out, unsigned long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* l32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += blockDim.x) {
        uint32_t x = seed[i & 1023] * 2654435761u + i * 40503u + blockIdx.x * 977u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        if (i < 1024) { const int e = i >> 2, q = i & 3; l32[i] = (((e >> (2 * q)) & 1) ? 0x3F80u : 0u) | (((e >> (2 * q + 1)) & 1) ? 0x3F800000u : 0u); }
        else if (i < (BYTES_OFF + 8192) / 4) l32[i] = x & (x >> 8) & 0x7F7F7F7Fu;       // spike bytes, rate ~ 1/4 .. 1/3
        else if (i < V_OFF / 4) l32[i] = ((x & 1) ? 0x3F80u : 0u) | ((x & 2) ? 0xBF800000u : 0u);
        else l32[i] = (x & 0x807F807Fu) | 0x3F003F00u;
    }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 15, lg = lane >> 4;
    f32x4 M[4][2], Y[4][4][2];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) { M[a][b] = f32x4{0, 0, 0, 0}; for (int o = 0; o < 4; ++o) Y[o][a][b] = f32x4{0, 0, 0, 0}; }
    // U build role: wave (wm, wn) builds M-tile wn of row-wave wm: row = wm*64 + wn*16 + lr, k-group lg
    const int brow = wm * 64 + wn * 16 + lr;
    const unsigned char* bytes = smem + BYTES_OFF + brow * 64 + lg;             // + pixel * 4
    unsigned char* const uimg = smem + U_OFF;
    const unsigned char* Ab = uimg + (wm * 64 + lr) * 64 + (lg << 4);           // + buf * 8192 + mt * 1024
    const unsigned char* Bb = smem + V_OFF + (wn * 32 + lr) * 64 + (lg << 4);   // + slot * 24576 + plane * 8192 + nt * 1024
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int xi = 0; xi < 16; ++xi) {
            const int i = xi >> 2, j = xi & 3;
            // pixels (2 rows x 2 cols of the 4x4 patch) and signs of B^T d B for this xi (wave-uniform)
            const int r1 = i == 0 ? 0 : i == 3 ? 1 : 1, r2 = i == 0 ? 2 : i == 3 ? 3 : 2;
            const int c1 = j == 0 ? 0 : j == 3 ? 1 : 1, c2 = j == 0 ? 2 : j == 3 ? 3 : 2;
            const bool sr = (i == 0 || i == 3) ? true : (i == 2), sc = (j == 0 || j == 3) ? true : (j == 2);
#pragma unroll 2
            for (int ch = 0; ch < 8; ++ch) {
                const int buf = ch & 1, slot = (xi * 8 + ch) % 3;
                if (BUILD) {
                    const uint32_t b11 = bytes[(r1 * 4 + c1) * 4 + ch % 4], b12 = bytes[(r1 * 4 + c2) * 4 + ch % 4];
                    const uint32_t b21 = bytes[(r2 * 4 + c1) * 4 + ch % 4], b22 = bytes[(r2 * 4 + c2) * 4 + ch % 4];
                    const bf16x8 t11 = *reinterpret_cast<const bf16x8*>(smem + (b11 << 4)), t12 = *reinterpret_cast<const bf16x8*>(smem + (b12 << 4));
                    const bf16x8 t21 = *reinterpret_cast<const bf16x8*>(smem + (b21 << 4)), t22 = *reinterpret_cast<const bf16x8*>(smem + (b22 << 4));
                    const bf16x8 ra = pk_addsub(t11, t12, sc), rb = pk_addsub(t21, t22, sc);
                    const bf16x8 u = pk_addsub(ra, rb, sr);
                    *reinterpret_cast<bf16x8*>(uimg + buf * 8192 + brow * 64 + (lg << 4)) = u;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
                bf16x8 a[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(Ab + buf * 8192 + mt * 1024);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    bf16x8 b[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8*>(Bb + slot * 24576 + pl * 8192 + nt * 1024);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            M[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[pl], M[mt][nt], 0, 0, 0);
                }
            }
            if (BUILD < 2) {
                // fold M into the outputs it feeds: Y_ab += At[a][i] At[b][j] M   (1, 2 or 4 outputs, signs by (i, j))
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) {
                        const int sa = a2 == 0 ? (i < 3 ? 1 : 0) : (i == 0 ? 0 : i == 1 ? 1 : -1);
                        const int sb = b2 == 0 ? (j < 3 ? 1 : 0) : (j == 0 ? 0 : j == 1 ? 1 : -1);
                        const int sg = sa * sb;              // wave-uniform
                        if (sg != 0) {
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                                for (int nt = 0; nt < 2; ++nt)
                                    Y[a2 * 2 + b2][mt][nt] = sg > 0 ? Y[a2 * 2 + b2][mt][nt] + M[mt][nt] : Y[a2 * 2 + b2][mt][nt] - M[mt][nt];
                        }
                    }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) M[mt][nt] = BUILD < 2 ? f32x4{0, 0, 0, 0} : M[mt][nt];
        }
    }
    float s = 0.f;
    for (int o = 0; o < 4; ++o) for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) s += Y[o][a][b][r] + M[a][b][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int BUILD>
static void run(const char* name, const uint32_t* seed, float* out, unsigned long long* clk, double seconds) {
    int iters = 40;
    const int grid = 256;                                 // one work-group per CU (256 VGPRs, ~101 KB of LDS)
    const void* kern = BUILD == 3 ? (const void*)probe3 : (const void*)probe<BUILD == 3 ? 0 : BUILD>;
    const int lds = BUILD == 3 ? LDS_BYTES3 : LDS_BYTES;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double flop = (double)grid * 8 * iters * 16.0 * 8.0 * 64.0 * 32.0 * 32.0 * 2.0 * 3.0;
    auto t0 = std::chrono::steady_clock::now();
    double last = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) {
            void* args[] = {(void*)&seed, (void*)&out, (void*)&clk, (void*)&iters};
            (void)hipLaunchKernel(kern, dim3(grid), dim3(512), args, lds, 0);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        last = 20 * flop / (ms * 1e-3) / 1e12;
    }
    static unsigned long long h[512];
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0;
    for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; ticks += (double)h[2 * i + 1]; }
    printf("%-64s %5.0f TF executed after %.0f s, in-kernel clock %.3f GHz  (direct-conv equivalent x2.25: %5.0f TF)\n", name, last, seconds,
           cyc / ticks * 0.1, last * 2.25);
    fflush(stdout);
}

int main() {
    uint32_t* seed; float* out; unsigned long long* clk;
    (void)hipMalloc(&seed, 4096); (void)hipMalloc(&out, 512 * 512 * 4); (void)hipMalloc(&clk, 1024 * 8);
    uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i * 747796405u + 2891336453u;
    (void)hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
    run<2>("MFMA + fragment reads only (4 x 2 tiles per wave, 1 WG per CU)", seed, out, clk, 2.0);
    run<0>("+ fold of M into the four outputs after every 8 chunks", seed, out, clk, 2.0);
    run<1>("+ transformed-spike fragments built through LDS (full structure)", seed, out, clk, 3.0);
    run<3>("full structure, integer fragment build, software-pipelined", seed, out, clk, 3.0);
    return 0;
}
