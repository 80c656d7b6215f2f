// Numerics probe: is an exact 3-way bf16 split of fp32 weights on v_mfma_f32_32x32x16_bf16 as accurate as the
// fp32 MFMA chain for binary (spike) A operands?  Compares both against an fp64 host reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

static inline uint16_t f2bf_rn(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// one wave: C[32][32] = A[32][K] (0/1) x W[K][32];  A given as bytes, W as fp32 [K][32]
__global__ void k_f32(const uint8_t* A, const float* W, float* C, int K) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 2) {
        float a = (float)A[i * K + k + h];
        float b = W[(k + h) * 32 + i];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
__global__ void k_bf16x3(const uint8_t* A, const uint16_t* Whi, const uint16_t* Wmid, const uint16_t* Wlo, float* C, int K, int order) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 16) {
        bf16x8 a, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            const int kk = k + 8 * h + j;
            a[j] = A[i * K + kk] ? (short)0x3F80 : (short)0;
            bh[j] = (short)Whi[kk * 32 + i]; bm[j] = (short)Wmid[kk * 32 + i]; bl[j] = (short)Wlo[kk * 32 + i];
        }
        if (order == 0) {   // lo first (small terms first)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

int main() {
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int K = cfg == 2 ? 12544 : 2304;
        const double rate = cfg == 1 ? 0.5 : 0.1;
        const float wscale = cfg == 2 ? 1.0f / sqrtf(12544.f) : 0.01f;
        std::mt19937 rng(123 + cfg);
        std::normal_distribution<float> nd(0.f, 1.f); std::uniform_real_distribution<float> ud(0.f, 1.f);
        std::vector<uint8_t> A(32 * K); std::vector<float> W(K * 32);
        std::vector<uint16_t> hi(K * 32), mid(K * 32), lo(K * 32);
        for (auto& a : A) a = ud(rng) < rate;
        double split_err = 0;
        for (int t = 0; t < K * 32; ++t) {
            float w = cfg == 2 ? (ud(rng) * 2 - 1) * wscale : nd(rng) * wscale; W[t] = w;
            hi[t] = f2bf_rn(w); float r1 = w - bf2f(hi[t]); mid[t] = f2bf_rn(r1); float r2 = r1 - bf2f(mid[t]); lo[t] = f2bf_rn(r2);
            split_err = fmax(split_err, fabs((double)w - ((double)bf2f(hi[t]) + bf2f(mid[t]) + bf2f(lo[t]))));
        }
        uint8_t* dA; float *dW, *dC; uint16_t *dh, *dm, *dl;
        hipMalloc(&dA, A.size()); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, 4096); hipMalloc(&dh, hi.size() * 2); hipMalloc(&dm, hi.size() * 2); hipMalloc(&dl, hi.size() * 2);
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dh, hi.data(), hi.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dm, mid.data(), hi.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dl, lo.data(), hi.size() * 2, hipMemcpyHostToDevice);
        std::vector<double> ref(1024, 0.0);
        for (int i = 0; i < 32; ++i) for (int k = 0; k < K; ++k) if (A[i * K + k]) for (int n = 0; n < 32; ++n) ref[i * 32 + n] += (double)W[k * 32 + n];
        std::vector<float> c32(1024), c0(1024), c1(1024);
        hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dW, dC, K); hipMemcpy(c32.data(), dC, 4096, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_bf16x3, dim3(1), dim3(64), 0, 0, dA, dh, dm, dl, dC, K, 0); hipMemcpy(c0.data(), dC, 4096, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_bf16x3, dim3(1), dim3(64), 0, 0, dA, dh, dm, dl, dC, K, 1); hipMemcpy(c1.data(), dC, 4096, hipMemcpyDeviceToHost);
        double e32 = 0, e0 = 0, e1 = 0, r32 = 0, r0 = 0, r1 = 0, mag = 0;
        for (int t = 0; t < 1024; ++t) {
            e32 = fmax(e32, fabs(c32[t] - ref[t])); e0 = fmax(e0, fabs(c0[t] - ref[t])); e1 = fmax(e1, fabs(c1[t] - ref[t]));
            r32 += pow(c32[t] - ref[t], 2); r0 += pow(c0[t] - ref[t], 2); r1 += pow(c1[t] - ref[t], 2); mag += ref[t] * ref[t];
        }
        printf("K=%d rate=%.2f |ref|rms=%.4f  max split residual %.3e\n", K, rate, sqrt(mag / 1024), split_err);
        printf("   fp32 MFMA chain      : max err %.3e  rms %.3e\n", e32, sqrt(r32 / 1024));
        printf("   bf16x3 (lo,mid,hi)   : max err %.3e  rms %.3e\n", e0, sqrt(r0 / 1024));
        printf("   bf16x3 (hi,mid,lo)   : max err %.3e  rms %.3e\n", e1, sqrt(r1 / 1024));
    }
    return 0;
}
