// Per-channel affine + residual + ReLU over an NCHW tensor in ONE pass (included by snn_kernels.hip).
//
// Not part of the spiking path: the detector's ResNet-50-FPN (stock torch modules, faster_rcnn.py:693-694) applies
// FrozenBatchNorm2d as  x * scale[c] + bias[c]  (two element-wise torch launches), then a residual add and a ReLU (two more):
// 4.3 of the backbone's 10.7 ms per batch are those passes (tools/prof_e2e.py).  This kernel performs the same
// operations in the same order with the same roundings (the library is built with -ffp-contract=off: mul, add, add stay
// separate), so its output is bit-identical to the torch sequence - an HBM-bound streaming kernel, 16 B per lane.
#pragma once

struct AffineArgs {
    const float* x;
    const float* scale;      // [C]
    const float* bias;       // [C]
    const float* residual;   // nullable, same shape as x
    float* y;                // may alias x
    int C, HW, relu;
    int chunks;              // work-groups per (n, c) plane
};

__global__ __launch_bounds__(256) void k_affine_act(const AffineArgs a) {
    const int plane = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
    const int c = plane % a.C;
    const float s = a.scale[c], b = a.bias[c];
    const size_t base = (size_t)plane * a.HW;
    const bool relu = a.relu != 0;
    auto f = [&](float v, float r) {
        float o = v * s;
        o = o + b;
        if (a.residual) o = o + r;
        return relu ? ((o <= 0.0f) ? 0.0f : o) : o;      // torch's relu = threshold(x, 0, 0): NaN stays NaN, -0.0 becomes +0.0 (fmaxf would swallow NaN)
    };
    // planes whose start is 16-B aligned (HW % 4 == 0) go as float4, anything else element by element
    if ((a.HW & 3) == 0) {
        const int n4 = a.HW >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(a.x + base);
        const float4* r4 = a.residual ? reinterpret_cast<const float4*>(a.residual + base) : nullptr;
        float4* y4 = reinterpret_cast<float4*>(a.y + base);
        for (int i = chunk * 256 + threadIdx.x; i < n4; i += a.chunks * 256) {
            const float4 v = x4[i];
            const float4 r = r4 ? r4[i] : float4{0.f, 0.f, 0.f, 0.f};
            y4[i] = float4{f(v.x, r.x), f(v.y, r.y), f(v.z, r.z), f(v.w, r.w)};
        }
    } else {
        for (int i = chunk * 256 + threadIdx.x; i < a.HW; i += a.chunks * 256)
            a.y[base + i] = f(a.x[base + i], a.residual ? a.residual[base + i] : 0.0f);
    }
}
