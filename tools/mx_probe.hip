// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950: operand layout of fp4 (A) x fp6 e2m3 (B) with E8M0 block
// scales, exactness of integer-digit products, and the sustained rate of a register-resident loop.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mx_probe tools/mx_probe.hip && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// out[lane][4] = one MFMA with per-lane operands from memory
__global__ void k_one(const uint32_t* a, const uint32_t* b, const uint32_t* sa, const uint32_t* sb, float* out) {
    const int l = threadIdx.x;
    v8i va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = a[l * 8 + i]; vb[i] = b[l * 8 + i]; }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 4 /*A fp4*/, 2 /*B fp6 e2m3*/, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

__global__ void k_rate(const uint32_t* a, const uint32_t* b, float* out, int iters) {
    const int l = threadIdx.x & 63;
    v8i va[4], vb[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) { va[j][i] = a[(l * 4 + j) * 8 + i]; vb[j][i] = b[(l * 4 + j) * 8 + i]; }
    v4f c[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                c[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va[i], vb[j], c[i][j], 4, 2, 0, 127, 0, 127);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += c[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static uint32_t fp6_code(int units) {      // units of 1/8, |units| <= 15 exactly representable; returns 6-bit e2m3 code
    const int s = units < 0, u = abs(units);
    uint32_t code;
    if (u < 8) code = u;                   // subnormal: e = 0, m = u
    else code = (1u << 3) | (u - 8);       // e = 1: (1 + m/8) * 2^0, u = 8 + m
    return code | (s << 5);
}
static void put6(uint32_t* w, int j, uint32_t code) {       // element j of a lane's 32 x 6-bit fragment (192 bits)
    const int bit = 6 * j;
    w[bit >> 5] |= code << (bit & 31);
    if ((bit & 31) > 26) w[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
}

int main() {
    uint32_t *a, *b, *sa, *sb; float* out;
    hipMallocManaged(&a, 64 * 8 * 4); hipMallocManaged(&b, 64 * 8 * 4); hipMallocManaged(&sa, 256); hipMallocManaged(&sb, 256);
    hipMallocManaged(&out, 64 * 4 * 4);
    // ---- layout test: A[i][k] = 1 only at (i = ai, k = ak); B[k][n] = digit(k, n): C[ai][n] must equal B[ak][n]
    int bad = 0, tests = 0;
    srand(1);
    for (int trial = 0; trial < 40; ++trial) {
        const int ai = rand() % 16, ak = rand() % 128;
        memset(a, 0, 64 * 32); memset(b, 0, 64 * 32);
        for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127 + (l % 5) - 2; }   // scale per lane: 2^(-2..2)
        // hypothesis: lane l holds row/col (l & 15), k = 32 * (l >> 4) + j, element j at bits [4j, 4j+4) / [6j, 6j+6)
        a[(ai + 16 * (ak / 32)) * 8 + ((ak % 32) / 8)] |= 0x2u << (4 * ((ak % 32) % 8));     // fp4 1.0 = 0b0010
        int digit[128][16];
        for (int k = 0; k < 128; ++k) for (int n = 0; n < 16; ++n) {
            digit[k][n] = rand() % 31 - 15;
            put6(&b[(n + 16 * (k / 32)) * 8], k % 32, fp6_code(digit[k][n]));
        }
        k_one<<<1, 64>>>(a, b, sa, sb, out);
        hipDeviceSynchronize();
        for (int n = 0; n < 16; ++n) for (int i = 0; i < 16; ++i) {
            // C layout: col = lane & 15, row = (lane >> 4) * 4 + reg
            const float got = out[(n + 16 * (i / 4)) * 4 + (i % 4)];
            const int lb = n + 16 * (ak / 32);
            const float exp = i == ai ? digit[ak][n] / 8.0f * ldexpf(1.0f, (int)sb[lb] - 127) : 0.0f;
            ++tests; if (got != exp) { if (bad < 8) printf("  mismatch trial %d i %d n %d got %g exp %g\n", trial, i, n, got, exp); ++bad; }
        }
    }
    printf("layout/scale test: %d mismatches of %d\n", bad, tests);
    // ---- exact accumulation of many digit products: all spikes set, random digits, fp64 reference
    {
        memset(a, 0x22, 64 * 32);                          // every fp4 nibble = 1.0
        for (int l = 0; l < 64; ++l) for (int i = 4; i < 8; ++i) a[l * 8 + i] = 0;
        memset(b, 0, 64 * 32);
        double ref[16] = {0};
        for (int k = 0; k < 128; ++k) for (int n = 0; n < 16; ++n) {
            const int d = rand() % 31 - 15;
            put6(&b[(n + 16 * (k / 32)) * 8], k % 32, fp6_code(d));
            ref[n] += d / 8.0;
        }
        for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127; }
        k_one<<<1, 64>>>(a, b, sa, sb, out);
        hipDeviceSynchronize();
        double e = 0;
        for (int n = 0; n < 16; ++n) e = fmax(e, fabs(out[n * 4] - ref[n]));
        printf("sum of 128 digit products: max |err| vs fp64 = %g\n", e);
    }
    // ---- rate
    {
        uint32_t *ra, *rb; float* ro;
        const int blocks = 256 * 8, threads = 256;
        hipMalloc(&ra, 64 * 4 * 32); hipMalloc(&rb, 64 * 4 * 32); hipMalloc(&ro, blocks * threads * 4);
        uint32_t* h = (uint32_t*)malloc(64 * 4 * 32);
        for (int i = 0; i < 64 * 4 * 8; ++i) h[i] = ((uint32_t)rand() << 16) ^ rand();
        hipMemcpy(rb, h, 64 * 4 * 32, hipMemcpyHostToDevice);
        for (int i = 0; i < 64 * 4 * 8; ++i) h[i] = (((uint32_t)rand() << 16) ^ rand()) & 0x22222222u & (((uint32_t)rand() << 16) ^ rand());   // sparse spikes
        hipMemcpy(ra, h, 64 * 4 * 32, hipMemcpyHostToDevice);
        const int iters = 2000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_rate<<<blocks, threads>>>(ra, rb, ro, 10);
        hipEventRecord(e0);
        k_rate<<<blocks, threads>>>(ra, rb, ro, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 16 * 16 * 128 * 16.0 * iters * (double)blocks * (threads / 64);
        printf("register-resident fp4 x fp6 MFMA loop: %.3f ms, %.1f TFLOP/s (dense-equivalent)\n", ms, flop / ms / 1e9);
    }
    return 0;
}
