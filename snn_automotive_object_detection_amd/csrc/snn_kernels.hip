// gfx950 (MI355X, CDNA4) kernels of the spiking RPN / detector heads + their C ABI (include/snn_hip.h).
//
// Data flow (bit-planes everywhere a spike tensor would be; see include/snn_hip.h):
//   RPN  : k_encode_nchw -> k_conv3x3_lif (fp32-MFMA implicit GEMM, T-loop inside, LIF state in
//          registers, ballot -> spike planes) -> k_li_heads (both 1x1 LI heads, time-collapsed)
//   DET  : k_encode_rows -> k_spike_gemm (time-batched fc6) -> k_lif_scan -> k_spike_gemm (fc7)
//          -> k_lif_scan -> k_li_heads
// Reference loops replaced: rpn.py:84-121 (+126-200), faster_rcnn.py:470-516 (+520-618).
#include "snn_common.h"
#include "snn_hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// ------------------------------------------------------------------------------------------------
// error plumbing (thread-local; no exceptions, no abort)
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
static int check_launch(const char* name) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(-3, "launch of %s failed: %s", name, hipGetErrorString(e));
    return 0;
}
#define SNN_CHECK_LAUNCH(name)                 \
    do {                                       \
        const int rc_ = check_launch(name);    \
        if (rc_) return rc_;                   \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static NeuronP make_p(const snn_params* p, float v_th) {
    NeuronP q;
    q.ca = p->dt_tau_mem; q.cb = p->neg_dt_tau_syn; q.v_leak = p->v_leak; q.v_reset = p->v_reset;
    q.v_th = v_th;
    return q;
}

// ------------------------------------------------------------------------------------------------
// weight packing: GEMM operand B[k][n] -> fragment-major
//   packed[((kc*Nw + nt)*4 + qq)*64 + lane][r] = B[k = kc*32 + 4*qq + r + 16*(lane>>5)][n = nt*32 + (lane&31)]
// so that one ds_read_b128 per lane yields the B operands of 4 consecutive MFMAs and both the
// global->LDS copy and the LDS read are perfectly linear (no bank conflicts, no swizzle needed).
// ------------------------------------------------------------------------------------------------
enum { PACK_CONV3X3 = 0, PACK_LINEAR = 1 };

__global__ void k_pack_gemm_b(const float* __restrict__ src, float* __restrict__ dst, int mode,
                              int K, int N, int Kc, int Nw, int Cin, int Cp) {
    const size_t total = (size_t)Kc * Nw * 1024;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int r = idx & 3;
        const int lane = (idx >> 2) & 63;
        const int qq = (idx >> 8) & 3;
        const size_t blk = idx >> 10;
        const int nt = (int)(blk % Nw);
        const int kc = (int)(blk / Nw);
        const int k = kc * 32 + 4 * qq + r + 16 * (lane >> 5);
        const int n = nt * 32 + (lane & 31);
        float v = 0.0f;
        if (n < N) {
            if (mode == PACK_CONV3X3) {            // k = tap*Cp + ci ; src OIHW [N][Cin][3][3]
                const int tap = k / Cp, ci = k % Cp;
                if (ci < Cin) v = src[((size_t)n * Cin + ci) * 9 + tap];
            } else {                               // src [N][K] (nn.Linear weight)
                if (k < K) v = src[(size_t)n * K + k];
            }
        }
        dst[idx] = v;
    }
}

// both LI heads -> transposed [Kp][NOp] (NOp = NA+NB rounded up to 16)
__global__ void k_pack_heads(const float* __restrict__ wa, int NA, const float* __restrict__ wb, int NB,
                             int K, int Kp, int NOp, float* __restrict__ dst) {
    const int total = Kp * NOp;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int k = idx / NOp, j = idx % NOp;
        float v = 0.0f;
        if (k < K) {
            if (j < NA) v = wa[(size_t)j * K + k];
            else if (j < NA + NB) v = wb[(size_t)(j - NA) * K + k];
        }
        dst[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// K1a: constant-current LIF encoder, NCHW fp32 -> bit-planes [T][N*HW][Cw]
// Block = 32 positions x 8 channel words; thread = one position x 32 channels (reads coalesced along W, two 128-B
// segments per wave instruction).  The T words of a thread go through LDS so that every plane store of the block is
// one contiguous run of 32 positions x 8 words (1 KB at C = 256) instead of 4-byte pieces at a 32-byte pitch.
// VALU-bound (32 x T encoder steps per thread): packed fp32 arithmetic, see enc_step2_word.
// ------------------------------------------------------------------------------------------------
#define ENC_PB 32                                   // positions per block
// Wpad > 0: the planes carry a one-position zero halo around every image (row (n, y, x) -> (n*(H+2) + y+1)*(W+2) + x+1,
// W = Wpad); the halo itself is zeroed by the caller.  The fp4 x fp6 conv kernel reads its 3x3 taps from such planes
// without any border logic.
template <bool ZR>
__device__ __forceinline__ void encode_block(const float* __restrict__ feat, int C, int HW, int Cw, int T, const NeuronP& p,
                                             uint32_t* __restrict__ planes, size_t plane_stride, int n, int bx, int by,
                                             int Wpad = 0) {
    __shared__ uint32_t wbuf[SNN_MAX_STEPS * ENC_PB * 9];      // [t][position][8 words + 1 pad]
    const int pl = threadIdx.x & 31, cgl = threadIdx.x >> 5;
    const int pos = bx * ENC_PB + pl;
    const int cg = by * 8 + cgl;
    float x[32], v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int c = cg * 32 + j;
        x[j] = (pos < HW && c < C) ? feat[((size_t)n * C + c) * HW + pos] : 0.0f;
        v[j] = 0.0f;                              // rpn.py:93  v = zeros
    }
    for (int t = 0; t < T; ++t) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 31; j >= 0; --j) enc_step_word<ZR>(x[j], v[j], p, word);   // bit 31 first ... bit 0 last
        wbuf[(t * ENC_PB + pl) * 9 + cgl] = word;
    }
    __syncthreads();
    // store: thread -> (position tid >> 3, word tid & 7): consecutive threads write consecutive plane words
    const int sp = threadIdx.x >> 3, sw = threadIdx.x & 7;
    const int spos = bx * ENC_PB + sp, scg = by * 8 + sw;
    if (spos < HW && scg < Cw) {
        size_t row = (size_t)n * HW + spos;
        if (Wpad) {
            const int H = HW / Wpad, y = spos / Wpad, x = spos % Wpad;
            row = ((size_t)n * (H + 2) + y + 1) * (Wpad + 2) + x + 1;
        }
        uint32_t* out = planes + row * Cw + scg;
        for (int t = 0; t < T; ++t) out[(size_t)t * plane_stride] = wbuf[(t * ENC_PB + sp) * 9 + sw];
    }
}

template <bool ZR>
__global__ __launch_bounds__(256) void k_encode_nchw(const float* __restrict__ feat, int C, int HW, int Cw,
                                                     int T, NeuronP p, uint32_t* __restrict__ planes,
                                                     size_t plane_stride) {
    encode_block<ZR>(feat, C, HW, Cw, T, p, planes, plane_stride, blockIdx.z, blockIdx.x, blockIdx.y);
}

// all pyramid levels of the RPN head in one launch (the small levels are launch-latency bound on their own)
struct EncLevels {
    const float* feat[SNN_MAX_LEVELS];
    int HW[SNN_MAX_LEVELS], bpi[SNN_MAX_LEVELS];   // positions / blocks per image
    int blk_base[SNN_MAX_LEVELS + 1];               // first block of the level
    int pos_base[SNN_MAX_LEVELS];                   // first plane row of the level
    int Wpad[SNN_MAX_LEVELS];                       // 0, or the level's width when the planes carry a zero halo
    int n_levels;
};
template <bool ZR>
__global__ __launch_bounds__(256) void k_encode_levels(const EncLevels lv, int C, int Cw, int T, NeuronP p,
                                                       uint32_t* __restrict__ planes, size_t plane_stride) {
    int l = 0;
    while (l + 1 < lv.n_levels && (int)blockIdx.x >= lv.blk_base[l + 1]) ++l;
    const int local = blockIdx.x - lv.blk_base[l];
    encode_block<ZR>(lv.feat[l], C, lv.HW[l], Cw, T, p, planes + (size_t)lv.pos_base[l] * Cw, plane_stride,
                     local / lv.bpi[l], local % lv.bpi[l], blockIdx.y, lv.Wpad[l]);
}

// K1b: encoder on row-major x[R][D] -> bit-planes [T][R][Dw]; a wave covers 64 consecutive reduction indices per
// slot, so one ballot per step IS two plane words.  Each thread runs ENC_U independent elements (64 apart) to keep
// several loads and scan chains in flight.
#define ENC_U 4
template <bool ZR>
__global__ __launch_bounds__(256) void k_encode_rows(const float* __restrict__ x, int R, int D, int Dw, int T,
                                                     NeuronP p, uint32_t* __restrict__ planes,
                                                     size_t plane_stride) {
    const size_t Dp = (size_t)Dw * 32;
    const size_t total = (size_t)R * Dp;
    const int lane = threadIdx.x & 63;
    const size_t wave_base = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * ENC_U);
    float xv[ENC_U], v[ENC_U];
    size_t e[ENC_U];
#pragma unroll
    for (int u = 0; u < ENC_U; ++u) {
        e[u] = wave_base + (size_t)u * 64 + lane;
        xv[u] = 0.0f;
        v[u] = 0.0f;                              // faster_rcnn.py:484
        if (e[u] < total) {
            const size_t r = e[u] / Dp, k = e[u] % Dp;
            if (k < (size_t)D) xv[u] = x[r * D + k];
        }
    }
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int u = 0; u < ENC_U; ++u) {
            const bool z = enc_step_t<ZR>(xv[u], v[u], p);
            const unsigned long long m = __ballot(z);
            if ((lane & 31) == 0 && e[u] < total)
                planes[(size_t)t * plane_stride + (e[u] >> 5)] = (lane < 32) ? (uint32_t)m : (uint32_t)(m >> 32);
        }
    }
}

// K1b': the same encoder when D is a multiple of 32 (the detector: 12544): the [R][D] elements are then one contiguous
// run of plane words, and a LANE owns a whole word.  A work-group copies 256 words (32 KB) with 16-byte coalesced loads
// through LDS (row pitch 36 dwords: the 8 ds_read_b128 of a lane are conflict-free), then every lane scans its 32
// neurons with the 6-instruction step whose carry chain builds the plane word (enc_step_word) and the wave stores 64
// consecutive words per time step.  Against the ballot form: 6 instead of 11 vector instructions per neuron-step and 8
// times the bytes in flight per thread.
#define ENC_W_PITCH 36
template <bool ZR>
__global__ __launch_bounds__(256) void k_encode_rows_w(const float* __restrict__ x, size_t n_words, int T, NeuronP p,
                                                       uint32_t* __restrict__ planes, size_t plane_stride) {
    __shared__ __attribute__((aligned(16))) float tile[256 * ENC_W_PITCH];
    const int tid = threadIdx.x;
    const size_t w0 = (size_t)blockIdx.x * 256;                 // first plane word of the work-group
    const size_t nw = n_words - w0 < 256 ? n_words - w0 : 256;
    const f32x4* src = reinterpret_cast<const f32x4*>(x + w0 * 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = tid + 256 * j;                            // 16-byte piece q of the work-group: word q / 8, elements 4 (q % 8) ..
        f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
        if ((size_t)(q >> 3) < nw) v4 = src[q];
        *reinterpret_cast<f32x4*>(tile + (q >> 3) * ENC_W_PITCH + (q & 7) * 4) = v4;
    }
    __syncthreads();
    if ((size_t)tid >= nw) return;
    float xv[32], v[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(tile + tid * ENC_W_PITCH + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) { xv[4 * q + r] = t4[r]; v[4 * q + r] = 0.0f; }      // v = 0: faster_rcnn.py:484
    }
    uint32_t* dst = planes + w0 + tid;
    for (int t = 0; t < T; ++t) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 31; b >= 0; --b) enc_step_word<ZR>(xv[b], v[b], p, word);          // bit 31 first
        dst[(size_t)t * plane_stride] = word;
    }
}

// ------------------------------------------------------------------------------------------------
// K1c: MultiScaleRoIAlign (7x7, sampling_ratio 2, aligned=False) fused with the detector's constant-current
// encoder (roi_heads.py:1217 -> faster_rcnn.py:473,494): the [R,C,7,7] fp32 RoI features (100 MB at R=2000) are
// never materialised; each thread pools one (RoI, channel, bin) element - 4 samples x 4 bilinear taps - runs its
// T encoder steps in registers and the wave ballots straight into the bit-planes [T][R][Dw] (flatten order
// d = c*49 + ph*7 + pw).  Arithmetic follows torchvision's roi_align / the stock-torch stand-in op for op
// (explicit roundings, no fma): sample = (hy*hx)*v1 + (hy*lx)*v2 + (ly*hx)*v3 + (ly*lx)*v4, bin = mean of 4.
// ------------------------------------------------------------------------------------------------
struct RoiLevel { const float* feat; int H, W; float scale; };
struct RoiArgs {
    RoiLevel lv[4];
    const float* rois;        // [R][4] x1,y1,x2,y2 in image coordinates
    const int* roi_batch;     // [R]
    const int* roi_level;     // [R] index into lv
    float* pooled;            // nullable test hook: [R][C*49] fp32
    uint32_t* planes;
    unsigned long long plane_stride;
    int R, C, T, Dw;
    NeuronP p;
};

__device__ __forceinline__ float roi_bilinear(const float* __restrict__ f, int H, int W, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
    y = fmaxf(y, 0.0f);
    x = fmaxf(x, 0.0f);
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = __fsub_rn(y, (float)y_low), lx = __fsub_rn(x, (float)x_low);
    const float hy = __fsub_rn(1.0f, ly), hx = __fsub_rn(1.0f, lx);
    const float v1 = f[y_low * W + x_low], v2 = f[y_low * W + x_high];
    const float v3 = f[y_high * W + x_low], v4 = f[y_high * W + x_high];
    float acc = __fmul_rn(__fmul_rn(hy, hx), v1);
    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(hy, lx), v2));
    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ly, hx), v3));
    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(ly, lx), v4));
    return acc;
}

__global__ __launch_bounds__(256) void k_roi_align_encode(const RoiArgs a) {
    const int r = blockIdx.y;
    const int D = a.C * 49;
    const int d = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float val = 0.0f;
    if (d < D) {
        const int c = d / 49, ph = (d % 49) / 7, pw = d % 7;
        const RoiLevel L = a.lv[a.roi_level[r]];
        const float* roi = a.rois + (size_t)r * 4;
        const float x1 = __fmul_rn(roi[0], L.scale), y1 = __fmul_rn(roi[1], L.scale);
        const float rw = fmaxf(__fsub_rn(__fmul_rn(roi[2], L.scale), x1), 1.0f);
        const float rh = fmaxf(__fsub_rn(__fmul_rn(roi[3], L.scale), y1), 1.0f);
        const float bh = __fdiv_rn(rh, 7.0f), bw = __fdiv_rn(rw, 7.0f);
        const float* f = L.feat + ((size_t)a.roi_batch[r] * a.C + c) * (size_t)(L.H * L.W);
        // sample coordinate: start + (p + (i + .5)/2) * bin   (the stock op's grid form)
        float s[2][2];
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                const float gy = __fadd_rn((float)ph, __fdiv_rn((float)iy + 0.5f, 2.0f));
                const float gx = __fadd_rn((float)pw, __fdiv_rn((float)ix + 0.5f, 2.0f));
                s[iy][ix] = roi_bilinear(f, L.H, L.W, __fadd_rn(y1, __fmul_rn(gy, bh)), __fadd_rn(x1, __fmul_rn(gx, bw)));
            }
        // mean over the 2x2 samples (torch .mean(dim=(3,5)): sum in (iy, ix) order, then / 4)
        val = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(s[0][0], s[0][1]), s[1][0]), s[1][1]), 4.0f);
        if (a.pooled) a.pooled[(size_t)r * D + d] = val;
    }
    // encoder over T steps; the wave's 64 consecutive d are two plane words
    float v = 0.0f;
    const size_t e = (size_t)r * a.Dw * 32 + d;
    const bool in = d < a.Dw * 32;
    for (int t = 0; t < a.T; ++t) {
        const bool z = enc_step(val, v, a.p) && d < D;
        const unsigned long long m = __ballot(z);
        if ((lane & 31) == 0 && in)
            a.planes[(size_t)t * a.plane_stride + (e >> 5)] = (lane < 32) ? (uint32_t)m : (uint32_t)(m >> 32);
    }
}

// ------------------------------------------------------------------------------------------------
// K2: fused 3x3 spike convolution (implicit GEMM on the fp32 matrix cores) + LIF over the T loop.
//
// Work-group = 512 threads = 8 waves; tile = 64 positions (an 8x8 patch of one image of one level)
// x 256 output channels; wave w owns all 64 positions x channels [32w, 32w+32)  (2 x 1 MFMA tiles).
// Per lane: 32 accumulators + 32 membrane voltages + 32 synaptic currents stay in registers for the
// whole T loop; nothing but spike bits is written to HBM.
//   A operand: the encoder spikes of the (8+2)x(8+2) halo of the patch are expanded ONCE per time
//              step from bit-planes to an fp32 LDS image [halo position][channel] (104 KB at C=256);
//              every (tap, channel-chunk) operand is then a plain ds_read_b128 - the MFMA loop holds
//              no VALU instruction (on gfx950 each one costs ~6.5 cycles of fp32 matrix-pipe time).
//   B operand: every wave streams its own 4-KiB weight fragment per 32-deep chunk straight from the
//              packed global image (L2-resident: 2.4 MB) into registers, one chunk ahead of the
//              MFMAs (ping-pong register sets) - no LDS staging, no per-chunk barrier.
// Spatial tiles are independent for the whole T loop (the only coupling is the conv halo on the
// *encoder* spikes), so there is no inter-work-group synchronisation.
// ------------------------------------------------------------------------------------------------
struct ConvLevelDev {
    int pos_base;        // first row (position) of this level in the plane buffers
    int N, H, W;
    int tiles_x, tiles_per_img;
    int tile_begin;      // first blockIdx.x of this level
    int pad;
};
struct ConvArgs {
    const uint32_t* enc;
    uint32_t* spk;
    const float* wpk;
    unsigned long long* counts;
    float* dbg_cur;                              // nullable: input currents [T][P][Nw*32] (parity tests)
    unsigned long long enc_stride, spk_stride;   // words per time plane
    int Cw, Nw, T, n_levels, max_n, pad;
    NeuronP p;
    ConvLevelDev lv[SNN_MAX_LEVELS];
};

#define CONV_PH 8
#define CONV_PW 8
#define CONV_HALO ((CONV_PH + 2) * (CONV_PW + 2))
#define CONV_BNT 8                                // n-tiles (waves) per block
#define CONV_APAD 4                               // floats of padding per halo position (bank spread)
#define CONV_MAX_CW 12                            // 100 x (384+4) x 4 B = 155 KB of LDS
#define CONV_HW ((CONV_HALO * CONV_MAX_CW + 511) / 512)   // halo words per thread

template <bool DBG>
__global__ __launch_bounds__(512) void k_conv3x3_lif(const ConvArgs args) {
    constexpr int MT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ldsA = reinterpret_cast<float*>(smem);                  // [CONV_HALO][CST]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // ---- which tile ----
    int l = 0;
    const int b = blockIdx.x;
    while (l + 1 < args.n_levels && b >= args.lv[l + 1].tile_begin) ++l;
    const ConvLevelDev L = args.lv[l];
    const int local = b - L.tile_begin;
    const int n = local / L.tiles_per_img;
    const int rem = local % L.tiles_per_img;
    const int y0 = (rem / L.tiles_x) * CONV_PH, x0 = (rem % L.tiles_x) * CONV_PW;
    const int H = L.H, W = L.W;
    const size_t img_base = (size_t)L.pos_base + (size_t)n * H * W;
    const int Cw = args.Cw, Nw = args.Nw;
    const int CST = Cw * 32 + CONV_APAD;             // floats per halo position
    const int ntg = blockIdx.y * CONV_BNT + wave;    // this wave's 32-channel output tile
    const bool active = ntg < Nw;                    // wave-uniform

    // this lane's A rows: LDS offset (in 16-byte units: CST is a multiple of 4 floats) of position
    // (py, px) of the patch for the two M-tiles
    const f32x4* lds16 = reinterpret_cast<const f32x4*>(smem);
    uint32_t a_q[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        a_q[mt] = (uint32_t)(((mt * 4 + (li >> 3)) * (CONV_PW + 2) + (li & 7)) * (CST / 4) + 4 * lh);

    // which (mt, r) accumulator rows of this lane are inside the image -> spike mask
    uint32_t valid_bits = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lh);
            const int yy = y0 + mt * 4 + (row >> 3), xx = x0 + (row & 7);
            valid_bits |= (uint32_t)(yy < H && xx < W) << (mt * 16 + r);
        }

    f32x16 acc[MT];
    float v[MT][16], cur_i[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { v[mt][r] = args.p.v_leak; cur_i[mt][r] = 0.0f; }   // state fallback

    const int KC = 9 * Cw;
    // weight fragment of (chunk kc, tile ntg): 256 float4, lane reads [qq*64 + lane]
    // (wave-uniform base pointer + lane index: global_load with an SGPR base, no per-chunk VALU address math)
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(args.wpk) + (size_t)(active ? ntg : 0) * 256;
    const size_t wchunk = (size_t)Nw * 256;          // float4 per reduction chunk (all n-tiles)

    // ---- halo spike words of one time step: thread owns words idx = tid + 512*j ----
    uint32_t hw[CONV_HW];
    auto fetch_halo = [&](int t) {
        const uint32_t* enc_t = args.enc + (size_t)t * args.enc_stride;
#pragma unroll
        for (int j = 0; j < CONV_HW; ++j) {
            const int idx = tid + 512 * j;
            hw[j] = 0;
            if (idx < CONV_HALO * Cw) {
                const int cc = idx % Cw, hp = idx / Cw;
                const int y = y0 - 1 + hp / (CONV_PW + 2), x = x0 - 1 + hp % (CONV_PW + 2);
                if (y >= 0 && y < H && x >= 0 && x < W) hw[j] = enc_t[(img_base + (size_t)y * W + x) * Cw + cc];
            }
        }
    };
    auto expand_halo = [&]() {
#pragma unroll
        for (int j = 0; j < CONV_HW; ++j) {
            const int idx = tid + 512 * j;
            if (idx < CONV_HALO * Cw) expand_word(hw[j], ldsA + (idx / Cw) * CST + (idx % Cw) * 32);
        }
    };

    // ---- software pipeline state: chunk whose operands are fetched NEXT ----
    const f32x4* wnext = wsrc;       // weights: wraps around at KC (next step re-reads the same image)
    int kcB = 0;
    auto load_b = [&](f32x4 (&dst)[4]) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) dst[qq] = wnext[qq * 64 + lane];
        if (++kcB == KC) { kcB = 0; wnext = wsrc; } else wnext += wchunk;
    };
    int tapA = 0, ccA = 0;           // spike operands: (tap, channel word) of the chunk being fetched
    uint32_t a_cur[MT];              // lane's LDS index (16-B units) of the chunk in flight: one v_add per M-tile
    auto load_a_lo = [&](f32x4 (&dst)[MT][2]) {      // first half (qq 0,1) of the next chunk
        const uint32_t cq = (uint32_t)(((tapA / 3) * (CONV_PW + 2) + (tapA % 3)) * (CST / 4) + ccA * 8);   // scalar
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            a_cur[mt] = a_q[mt] + cq;
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) dst[mt][q2] = lds16[a_cur[mt] + q2];
        }
        if (++ccA == Cw) { ccA = 0; ++tapA; }
    };
    auto load_a_hi = [&](f32x4 (&dst)[MT][2]) {      // second half (qq 2,3) of the same chunk
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) dst[mt][q2] = lds16[a_cur[mt] + 2 + q2];
    };

    f32x4 b0[4], b1[4];
    f32x4 alo[MT][2], ahi[MT][2];
    fetch_halo(0);
    if (active) load_b(b0);
    unsigned long long n_spikes = 0;

    for (int t = 0; t < args.T; ++t) {
        __syncthreads();                                   // everyone done reading the previous image
        expand_halo();
        if (t + 1 < args.T) fetch_halo(t + 1);             // latency hidden behind this step's MFMAs
        __syncthreads();                                   // image of step t complete
        if (active) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
            tapA = 0; ccA = 0;
            load_a_lo(alo);
            // A operands run half a chunk ahead of the MFMAs, B operands one chunk ahead
            auto chunk = [&](int kc, const f32x4 (&bcur)[4], f32x4 (&bnext)[4]) {
                load_b(bnext);                 // chunk kc+1 (or chunk 0 of the next time step)
                load_a_hi(ahi);
                __builtin_amdgcn_sched_barrier(0);
                mma_half<MT, 0>(acc, alo, bcur);
                __builtin_amdgcn_sched_barrier(0);
                if (kc + 1 < KC) load_a_lo(alo);
                __builtin_amdgcn_sched_barrier(0);
                mma_half<MT, 1>(acc, ahi, bcur);
                __builtin_amdgcn_sched_barrier(0);
            };
            int kc = 0;
            for (; kc + 1 < KC; kc += 2) {     // two chunks per trip: ping-pong weight registers, no copies
                chunk(kc, b0, b1);
                chunk(kc + 1, b1, b0);
            }
            if (kc < KC) {                     // odd chunk count (C_in = 32 * odd)
                chunk(kc, b0, b1);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) b0[qq] = b1[qq];
            }
            if (DBG) {                         // test-hook instantiation: dump the step's input currents
                float* d = args.dbg_cur + (size_t)t * (args.spk_stride * 32);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = acc_row(r, lh);
                        const int yy = y0 + mt * 4 + (row >> 3), xx = x0 + (row & 7);
                        if (yy < H && xx < W)
                            d[(img_base + (size_t)yy * W + xx) * (Nw * 32) + ntg * 32 + li] = acc[mt][r];
                    }
            }
            // ---- LIF epilogue in registers; spikes leave as ballots ----
            uint32_t myword = 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    bool z = lif_step(acc[mt][r], v[mt][r], cur_i[mt][r], args.p);
                    z = z && ((valid_bits >> (mt * 16 + r)) & 1u);
                    const unsigned long long m = __ballot(z);
                    n_spikes += __popcll(m);
                    const int L0 = mt * 32 + r * 2;
                    myword = (lane == L0) ? (uint32_t)m : myword;
                    myword = (lane == L0 + 1) ? (uint32_t)(m >> 32) : myword;
                }
            }
            {   // lane -> (mt, r, half): one 32-channel word of one position
                const int mt = lane >> 5, r = (lane >> 1) & 15, hh = lane & 1;
                const int row = acc_row(r, hh);
                const int yy = y0 + mt * 4 + (row >> 3), xx = x0 + (row & 7);
                if (yy < H && xx < W)
                    args.spk[(size_t)t * args.spk_stride + (img_base + (size_t)yy * W + xx) * Nw + ntg] = myword;
            }
        }
    }
    if (args.counts != nullptr && lane == 0 && n_spikes != 0)
        atomicAdd(&args.counts[l * args.max_n + n], n_spikes);
}

// ------------------------------------------------------------------------------------------------
// K3: time-batched spike GEMM  cur[M][ldo] = A_bits[M][K] x W[K][N]   (rows m = t*R + r)
// Work-group = 8 waves = 128 rows x 256 columns; wave w owns the 128 rows x columns of n-tile w
// (4 x 1 MFMA tiles).  Per 32-deep chunk the 512 threads expand the 128 spike words of the tile to
// an fp32 LDS image [row][32 k] (double-buffered, one barrier per chunk; 16 bit->float conversions
// per thread per 64 MFMAs per wave), A operands are ds_read_b128, B fragments stream from global
// one chunk ahead exactly as in K2.
// blockIdx -> (row tile, panel) puts all work-groups of one XCD (blockIdx % 8) on the same weight
// panel whenever the panel count divides 8, so a panel is fetched from HBM once per XCD and then
// served from that XCD's L2.
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    const uint32_t* A;
    const float* wpk;
    float* out;
    int M, Kw, Nw, ldo, n_blocks, pad;
};

#define GEMM_AST 36                                 // floats per LDS row: 32 + 4 (conflict-free b128 reads)

template <int MT>
__global__ __launch_bounds__(512) void k_spike_gemm(const GemmArgs args) {
    static_assert(MT == 4, "512 threads expand 128 rows x 4 bytes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ldsA = reinterpret_cast<float*>(smem);                  // 2 x [128][GEMM_AST]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nb = blockIdx.x % args.n_blocks;
    const int mb = blockIdx.x / args.n_blocks;
    const int m0 = mb * (MT * 32);
    const int Kw = args.Kw, Nw = args.Nw, M = args.M;
    const int ntg = nb * 8 + wave;
    const bool active = ntg < Nw;                     // wave-uniform

    // expansion role: thread -> (row, byte) of the tile's spike words
    const int xrow = tid >> 2, xbyte = tid & 3;
    const uint32_t* xsrc = args.A + (size_t)min(m0 + xrow, M - 1) * Kw;
    float* xdst = ldsA + xrow * GEMM_AST + xbyte * 8;
    auto expand = [&](uint32_t w, int buf) {
        const uint32_t byte = (w >> (8 * xbyte)) & 0xffu;
        f32x4 lo, hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) { lo[r] = (float)((byte >> r) & 1u); hi[r] = (float)((byte >> (4 + r)) & 1u); }
        float* d = xdst + buf * (128 * GEMM_AST);
        *reinterpret_cast<f32x4*>(d) = lo;
        *reinterpret_cast<f32x4*>(d + 4) = hi;
    };

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;

    const f32x4* wsrc = reinterpret_cast<const f32x4*>(args.wpk) + (size_t)(active ? ntg : 0) * 256;   // uniform
    const size_t wchunk = (size_t)Nw * 256;
    const float* abase = ldsA + li * GEMM_AST + 16 * lh;

    f32x4 b0[4], b1[4];
    auto load_b = [&](f32x4 (&dst)[4], int kc) {
        const f32x4* wn = wsrc + (size_t)min(kc, Kw - 1) * wchunk;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) dst[qq] = wn[qq * 64 + lane];
    };
    auto chunk = [&](int kc, const f32x4 (&bcur)[4], f32x4 (&bnext)[4], uint32_t& wnext) {
        // stage chunk kc+1 (spike image + weights) while chunk kc is multiplied
        if (kc + 1 < Kw) expand(wnext, (kc + 1) & 1);
        wnext = xsrc[min(kc + 2, Kw - 1)];
        if (active) {
            load_b(bnext, kc + 1);
            f32x4 a[MT][4];
            const float* ab = abase + (kc & 1) * (128 * GEMM_AST);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) a[mt][qq] = *reinterpret_cast<const f32x4*>(ab + mt * 32 * GEMM_AST + 4 * qq);
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk<MT>(acc, a, bcur);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    uint32_t wnext = xsrc[0];
    expand(wnext, 0);
    wnext = xsrc[min(1, Kw - 1)];
    if (active) load_b(b0, 0);
    __syncthreads();
    int kc = 0;
    for (; kc + 1 < Kw; kc += 2) {
        chunk(kc, b0, b1, wnext);
        chunk(kc + 1, b1, b0, wnext);
    }
    if (kc < Kw) chunk(kc, b0, b1, wnext);
    if (!active) return;
    // ---- store currents (lanes 0-31 / 32-63 write two 128-B row segments per instruction) ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mt * 32 + acc_row(r, lh);
            if (m < M) args.out[(size_t)m * args.ldo + ntg * 32 + li] = acc[mt][r];
        }
}

// ------------------------------------------------------------------------------------------------
// K3b: exact bf16x3 spike GEMM on the bf16 matrix cores (16x the fp32 MFMA rate, 3 MFMAs per product).
//
// Spikes are exactly {0,1} and every fp32 weight is exactly hi + mid + lo with three bf16 values, so
//   A_bits x W  ==  A_bf16 x W_hi + A_bf16 x W_mid + A_bf16 x W_lo     (every product exact)
// with fp32 accumulation inside v_mfma_f32_32x32x16_bf16.  Measured against fp64 the result is as accurate
// as the fp32 MFMA chain (tools/bf16x3_numerics.hip: rms error 4.1e-8 vs 4.2e-8 at K=2304).
//
// cur[M][ldo] = A_bits[M][K] x W[K][N];   work-group = 8 waves = 256 rows x 128 columns, wave = 64 x 64
// = 4 x 4 tiles of v_mfma_f32_16x16x32_bf16 (one 32-deep chunk = one k-step; in an LDS-fed loop with this
// kernel's traffic the 16x16x32 shape sustains 2.47 PF against 1.90 PF for 32x32x16: tools/mfma_probe4.hip).
// Per 32-deep chunk: B = 3 planes x 128 x 32 bf16 (24 KB) copied global->LDS, A = 256 spike words expanded
// to bf16 in LDS (two 16-bit halves per row, one per thread); both double-buffered, one barrier per chunk.
// LDS rows are 64 B (no padding) with an XOR swizzle of the 16-B units: conflict-free ds_read_b128 fragment
// reads at 80 KB of LDS per work-group, so TWO work-groups share a CU and one's barrier/staging phase hides
// behind the other's MFMAs.
// CONV = true: row m = (t, position) and the chunk (tap, channel word) is gathered straight from the
// encoder bit-planes (9 taps, zero outside the image) - the un-fused time-batched 3x3 convolution.
// ------------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// LDS-DMA: 16 B per lane straight from global memory (wave-uniform 64-bit base + the lane's 32-bit byte offset)
// into LDS at (wave-uniform byte address in M0) + 16*lane; no VGPR destination, completion is counted on vmcnt.
// Issued as inline asm on purpose: once hipcc sees an LDS-DMA in flight it degrades every LDS wait of the loop to
// s_waitcnt lgkmcnt(0) and puts vmcnt(0) in front of every ds_write (possible alias); hidden from it, the fragment
// reads keep their exact counted waits.  The kernel waits for the DMA by hand (vmcnt(0) before the chunk barrier).
// a wave-uniform pointer as an SGPR pair (inline asm "s" operands are not legalised by hipcc)
__device__ __forceinline__ const void* sgpr_ptr(const void* p) {
    const unsigned long long x = (unsigned long long)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
// Three pieces (the three weight planes of a chunk) per call; s_nop 4 / s_nop 0: SGPR -> VMEM-base and M0 -> LDS-DMA
// wait states, which hipcc's hazard recogniser does not insert inside inline asm.
__device__ __forceinline__ void glds16(const void* p0, uint32_t voff, uint32_t d0) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(p0), "s"(d0) : "memory", "m0");
}
__device__ __forceinline__ void glds16x3(const void* p0, const void* p1, const void* p2, uint32_t voff,
                                         uint32_t d0, uint32_t d1, uint32_t d2) {
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3"
                 :: "v"(voff), "s"(p0), "s"(p1), "s"(p2), "s"(d0), "s"(d1), "s"(d2) : "memory", "m0");
}

// Work-group tile: the 8 waves form (8 / WN) x WN; a wave is 16*MT rows x 64 columns.
//   WN = 2: 64*MT rows x 128 columns (256 x 128 at MT = 4): 24 KB of weight planes per chunk
//   WN = 1: 128*MT rows x 64 columns (512 x 64):            12 KB per chunk for the same MFMA work - half the L2 -> LDS
//           copies per FLOP (the 256 x 128 tile pulls 14 B/clk/CU = 7 TB/s chip-wide out of the L2).  Measured equal.
#define G3_BM(wn, mt) ((8 / (wn)) * 16 * (mt))
#define G3_BN(wn) (64 * (wn))
#define G3_ROWB 64                                  // bytes per LDS weight row: 32 bf16, four 16-B units, XOR-swizzled
#define G3_AW_BYTES(wn) (G3_BM(wn, 4) * 4)          // raw spike words of one chunk (one per row): 1 or 2 KB
#define G3_B_BYTES(wn) (3 * G3_BN(wn) * G3_ROWB)    // three weight planes of one chunk: 24 or 12 KB
#define G3_SLOT(wn) (G3_AW_BYTES(wn) + G3_B_BYTES(wn))   // one ring slot = one 32-deep chunk
#define G3_LUT_BYTES 4096                           // byte -> 8 bf16 (0 / 1.0) expansion table
#define G3_STATE_BYTES (512 * 64)                   // register-fused variant: 16 LIF state values per thread
#define G3_LDS(nb, wn) (G3_LUT_BYTES + (nb) * G3_SLOT(wn))   // table at offset 0, then the ring (3 slots: 80896 / 47104 B)
// unit u (= k-group 8u..8u+7) of weight row r lives at physical unit u ^ swz(r), swz = [0,3,2,1][(r >> 2) & 3].  A
// 16x16x32 fragment read has lane l on row l&15, unit l>>4; the four 16-lane groups of a ds_read_b128
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) then each hit 16 distinct 16-B slots of the 256-B bank row.
#define G3_SWZ(r) ((0 - ((r) >> 2)) & 3)

struct Gemm3Args {
    const uint32_t* A;           // fc: [M][Kw] spike words;  conv: encoder planes [T][P][Cw]
    const uint16_t* wpk;         // [3][Kc][Np][32] bf16
    float* out;                  // [M][ldo]
    unsigned long long plane_elems;     // Kc*Np*32
    unsigned long long enc_stride;      // conv: words per time plane
    int M, Kc, Np, ldo, n_blocks;
    int Cw, P_total, n_levels;          // conv only
    // conv + LIF: spikes leave as bit-planes.  G3_CONV_LIF_REG: rows are positions, the T loop runs inside;
    // G3_CONV_LIF_TILE: a 256-row tile = all T time steps of pb = 256/T positions (row = t*pb + position)
    int T, pb;
    uint32_t* spk;
    unsigned long long spk_stride;
    NeuronP p;
    ConvLevelDev lv[SNN_MAX_LEVELS];
};

__device__ __forceinline__ uint32_t bf16_pair(uint32_t w, int j) {      // bits 2j, 2j+1 -> two bf16 (0 / 1.0)
    const uint32_t t = (w >> (2 * j)) & 3u;
    return ((t | (t << 15)) & 0x10001u) * 0x3F80u;
}

// FUSE (conv only): M = positions; per time step the 9*Cw chunks are accumulated, then the LIF update runs
// on the accumulators in registers and only spike bits are written.
//
// LDS: a 4-KB table byte -> 8 bf16, a ring of NB slots (one 32-deep chunk each: 256 raw spike words + 3 weight
// planes), and for FUSE 16 LIF state values per thread.  Staging runs ahead, so the slot of chunk c+1 is already
// complete during chunk c and the first fragments of chunk c+1 are read from LDS BEFORE the barrier that ends chunk
// c - the matrix pipe does not drain at the barrier.
// NB = 3 (80 KB, two work-groups per CU): the weight planes of chunk c+2 are copied during chunk c and must have
// landed at its end (s_waitcnt vmcnt(0)).  NB = 4 (fused variant, which owns its CU): the planes of chunk c+3 are
// copied during chunk c and need to land only by the end of chunk c+1 (s_waitcnt vmcnt(3): the three youngest
// operations, this chunk's copies, stay in flight across the barrier) - no L2 latency is ever waited for.
// The A (spike) fragment of a lane is 8 consecutive k of one row = ONE BYTE of that row's spike word: it is fetched
// as table[byte] by a single ds_read_b128.  No expanded spike image is ever built (the first version spent 36 VALU
// instructions + 2 ds_write_b128 per thread and chunk on it, 10 % of the kernel: every VALU instruction beside
// v_mfma_f32_16x16x32_bf16 competes for the SIMD's vector issue, which the MFMAs alone hold half of the time).
// PD = fragment prefetch distance inside a chunk, in groups of 4 MFMAs (one weight fragment per group).
enum { G3_FC = 0, G3_CONV = 1, G3_CONV_LIF_REG = 2, G3_CONV_LIF_TILE = 3, G3_FC_LIF_TILE = 4 };
// LIF_TILE epilogue: the current tile goes through LDS in two passes of CG = 32*WN columns, row pitch CG + 4 floats
// (conflict-free for the accumulator writes and the column reads)
#define G3_TILE_CG(wn) (32 * (wn))
#define G3_TILE_BYTES(wn) (G3_BM(wn, 4) * (G3_TILE_CG(wn) + 4) * 4)     // 69632 / 73728 B

// MT = 16-row M-tiles per wave: the work-group tile is 64*MT rows (256; 192 / 128 only where a small problem spreads
// better over the CUs that way - per unit of work the smaller tiles are slower: fc6 1.03 / 1.07 / 1.21 ms at MT 4 / 3 / 2).
template <int MODE, int NB, int MT, int WN>
__global__ __launch_bounds__(512, MODE == G3_CONV_LIF_REG ? 2 : 4) void k_gemm_bf16x3(const Gemm3Args args) {
    constexpr bool CONV = MODE == G3_CONV || MODE == G3_CONV_LIF_REG || MODE == G3_CONV_LIF_TILE;
    constexpr bool FUSE = MODE == G3_CONV_LIF_REG, TILE = MODE == G3_CONV_LIF_TILE || MODE == G3_FC_LIF_TILE;
    static_assert(MT >= 2 && MT <= 4 && (MT == 4 || !FUSE), "M-tiles per wave");
    static_assert(WN == 1 || (WN == 2 && true), "waves along N");
    static_assert(WN == 2 || !FUSE, "the register-fused variant keeps the 4 x 2 wave grid");
    constexpr int BM = G3_BM(WN, MT), BN = G3_BN(WN), WROWS = 16 * MT;   // rows, columns per work-group; rows per wave
    constexpr int AW_BYTES = G3_AW_BYTES(WN);
    static_assert(NB == 3 || NB == 4, "ring depth");
    constexpr int SLOT = G3_SLOT(WN);
    constexpr int PD = (CONV && !FUSE) ? 2 : 3, RING = PD + 1;  // 12 groups per chunk: RING must divide 12 (the 128-register conv
                                                               // rows have 128 registers: one fragment less in flight)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned char* const lut = smem;                // table at LDS offset 0: a fragment address is just (byte << 4)
    unsigned char* const ring = smem + G3_LUT_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // Plain row-major tile order, column block fastest: work-group b runs on XCD b % 8 (round-robin dispatch), so with
    // 2 (or 4, 8) column blocks every XCD only ever sees ONE weight panel - half of the 3.5 MB of conv weight planes,
    // which then stay resident in its 4-MB L2 beside the streaming spike planes.  Both re-orderings tried (each XCD a
    // contiguous eighth of the tiles; both column blocks of a tile on one XCD) put all panels on every XCD and
    // multiplied the L2 fills: FETCH_SIZE 102 -> 323 / 535 MB per launch at unchanged kernel time (profiles/r1_h_*).
    const int nb = blockIdx.x % args.n_blocks;
    const int mb = blockIdx.x / args.n_blocks;
    const int m0 = TILE ? mb * args.pb : mb * BM;            // first row (TILE: first position) of the tile
    const int Kc = args.Kc, Np = args.Np, M = args.M;

    if (tid < 256) {                                // table entry e: element j = bit j of e as bf16
        uint4 q;
        q.x = bf16_pair(tid, 0); q.y = bf16_pair(tid, 1); q.z = bf16_pair(tid, 2); q.w = bf16_pair(tid, 3);
        *reinterpret_cast<uint4*>(lut + tid * 16) = q;
    }

    // ---- A staging role: thread -> row (the first BM threads).  A spike word is addressed as (wave-uniform 64-bit
    // base in SGPRs) + (32-bit byte offset of the lane's row): no per-chunk 64-bit vector arithmetic ----
    const bool a_role = wave * 64 < BM;
    const int xrow = tid & (G3_BM(WN, 4) - 1);
    const int xt = TILE ? xrow / args.pb : 0;       // TILE: time step of the row
    const int xm = TILE ? (xt < args.T ? m0 + xrow % args.pb : M) : m0 + xrow;
    uint32_t a_off = 0;                             // bytes: fc row / conv centre tap, channel word 0
    int a_pitch = 0;                                // conv: bytes per image row of the lane's pyramid level
    uint32_t a_valid = 0;                           // conv: 9-bit tap validity
    if (CONV) {
        if (xm < M) {
            const int t = FUSE ? 0 : (TILE ? xt : xm / args.P_total), p = (FUSE || TILE) ? xm : xm % args.P_total;
            int l = 0;
            while (l + 1 < args.n_levels && p >= args.lv[l + 1].pos_base) ++l;
            const int H = args.lv[l].H, W = args.lv[l].W;
            const int local = p - args.lv[l].pos_base;
            const int rem = local % (H * W);
            const int y = rem / W, x = rem % W;
            a_off = (uint32_t)(((size_t)t * args.enc_stride + (size_t)p * args.Cw) * 4);
            a_pitch = W * args.Cw * 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                a_valid |= (uint32_t)(yy >= 0 && yy < H && xx >= 0 && xx < W) << tap;
            }
        }
    } else if (TILE) {                              // fc rows of the spike planes [T][M][Kc]; unused tile rows read row 0
        a_off = xm < M ? (uint32_t)(((size_t)xt * M + xm) * Kc * 4) : 0u;
    } else {
        a_off = (uint32_t)((size_t)min(xm, M - 1) * Kc * 4);
    }
    // Spike-word loads are issued as inline asm: hipcc must not see them, or it drains the LDS-DMA queue
    // (vmcnt(0)) at their first use while weight planes are still in flight.  A word is consumed only after the
    // s_waitcnt vmcnt(0) that ends the chunk it was issued in.
    // The fetch stream walks the chunk sequence (t, tap dy, tap dx, channel word) with scalar counters.
    // (s_nop 4: an SGPR written by SALU / v_readfirstlane needs 5 wait states before a VMEM instruction reads it as
    // its base address, and hipcc's hazard recogniser does not look into inline asm.)
    // The stream is a running scalar pointer: within one tap row (dy) the wave-uniform word offset dx*Cw + cc just
    // increments by one per chunk; every 3*Cw chunks the lanes step one image row down, every Kc chunks one time
    // step on.  Past the last chunk the stream wraps to the start (staged, never multiplied).
    const int n_steps = FUSE ? args.T : 1;
    const uint32_t* f_tbase = args.A;               // scalar: A + t*enc_stride
    int f_off = CONV ? -args.Cw : 0;                // scalar: dx*Cw + cc   (fc: kc)
    uint32_t f_voff = CONV ? a_off - (uint32_t)a_pitch : a_off;     // lane: row offset of tap row dy
    uint32_t f_mask = 1u;                           // conv: bit of the current tap
    int f_t = 0, f_kc = 0, f_cc = 0, f_dx = 0;
    auto fetch_next = [&](uint32_t& w) {
        w = 0u;
        const void* sbase = sgpr_ptr(f_tbase + f_off);
        if (CONV) {
            if (a_role && (a_valid & f_mask))
                asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "+v"(w) : "v"(f_voff), "s"(sbase) : "memory");
        } else {
            if (a_role) asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "+v"(w) : "v"(f_voff), "s"(sbase) : "memory");
        }
        ++f_off;
        if (CONV && ++f_cc == args.Cw) {
            f_cc = 0;
            f_mask <<= 1;
            if (++f_dx == 3) { f_dx = 0; f_off -= 3 * args.Cw; f_voff += (uint32_t)a_pitch; }
        }
        if (++f_kc == Kc) {
            f_kc = 0; f_cc = 0; f_dx = 0; f_mask = 1u;
            f_off = CONV ? -args.Cw : 0;
            if (++f_t == n_steps) f_t = 0;
            f_tbase = args.A + (FUSE ? (size_t)f_t * args.enc_stride : 0);
            f_voff = CONV ? a_off - (uint32_t)a_pitch : a_off;
        }
    };
    auto store_w = [&](uint32_t w, uint32_t slot_off) {     // the raw spike word of the thread's row
        if (a_role) {
            uint32_t l;                             // lane id, re-derived (2 VALU) instead of a register held all loop
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            *reinterpret_cast<uint32_t*>(ring + slot_off + wave * 256 + l * 4) = w;
        }
    };

    // ---- B staging: LDS-DMA, one 1-KiB piece (16 rows of one plane) per instruction; the 3 * BN/16 pieces of a chunk
    // go round the 8 waves (piece p = wave, wave + 8, wave + 16: all of them on row block wave % (BN/16)).  Lane L
    // lands in physical unit L&3 of row (L>>2) of the piece, so it fetches logical unit (L&3) ^ swz(row): the swizzle
    // is applied on the SOURCE address, the LDS image stays lane-linear ----
    constexpr int RBLK = BN / 16, NPIECE = 3 * RBLK;            // row blocks per plane, pieces per chunk (24 / 12)
    const int brow = (wave % RBLK) * 16 + (lane >> 2);
    const int bcol = min(nb * BN + brow, Np - 1);               // columns past Np: any valid row (never stored)
    const uint32_t b_off = (uint32_t)(bcol * 64 + (((lane & 3) ^ G3_SWZ(brow)) << 4));     // bytes within a chunk plane
    const unsigned long long b_chunk = (unsigned long long)Np * 64, b_plane = args.plane_elems * 2;   // bytes
    unsigned long long s_ptr = (unsigned long long)args.wpk;   // weight stream: plane 0 of the next chunk (scalar)
    int s_kc = 0;
    const uint32_t b_dst = smem_base + G3_LUT_BYTES + AW_BYTES + (wave % RBLK) * 1024;      // + slot offset, plane
    auto stage_next = [&](uint32_t slot_off) {
        const uint32_t d = __builtin_amdgcn_readfirstlane(b_dst + slot_off);                // wave-uniform LDS address
        if (WN == 2) {
            glds16x3(sgpr_ptr(reinterpret_cast<const void*>(s_ptr)), sgpr_ptr(reinterpret_cast<const void*>(s_ptr + b_plane)),
                     sgpr_ptr(reinterpret_cast<const void*>(s_ptr + 2 * b_plane)), b_off,
                     d, d + BN * G3_ROWB, d + 2 * BN * G3_ROWB);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int piece = wave + 8 * j;                 // wave-uniform
                if (piece < NPIECE) {
                    const int pl = piece / RBLK;
                    glds16(sgpr_ptr(reinterpret_cast<const void*>(s_ptr + pl * b_plane)), b_off, d + pl * (BN * G3_ROWB));
                }
            }
        }
        s_ptr += b_chunk;
        if (++s_kc == Kc) { s_kc = 0; s_ptr = (unsigned long long)args.wpk; }
    };

    // 16x16 tiles: lane holds column lane&15, rows (lane>>4)*4 + reg of each tile
    const int lr = lane & 15, lg = lane >> 4;
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // A fragment: row (wm*64 + mt*16 + lr), k = 8*lg .. 8*lg+7  ->  table[byte lg of the row's spike word]
    const unsigned char* const w_rd = ring + (wm * WROWS + lr) * 4;                          // + slot offset, mt*64
    const int lg8 = 8 * lg;
    auto rd_w = [&](uint32_t slot_off, int mt) { return *reinterpret_cast<const uint32_t*>(w_rd + slot_off + mt * 64); };
    auto rd_a = [&](uint32_t w) { return *reinterpret_cast<const bf16x8*>(lut + (__builtin_amdgcn_ubfe(w, lg8, 8) << 4)); };
    // B fragment: row (tile*16 + lr), logical unit lg; swz depends on lr only
    const unsigned char* const b_rd = ring + AW_BYTES + (wn * 64 + lr) * G3_ROWB + ((lg ^ G3_SWZ(lr)) << 4);
    // group g of a chunk = (N-tile g/3, plane 2 - g%3): per accumulator the small terms first (lo, mid, hi)
    auto rd_b = [&](uint32_t slot_off, int g) {
        return *reinterpret_cast<const bf16x8*>(b_rd + slot_off + (2 - g % 3) * (BN * G3_ROWB) + (g / 3) * 16 * G3_ROWB);
    };

    // LIF state of the fused variant, whole T loop: v (64 registers) and the synaptic current i (48 registers;
    // the 16 values of M-tile 3 live in LDS, private to the thread, touched once per time step - they are what
    // keeps the main loop free of scratch spills)
    f32x4 v[FUSE ? 4 : 1][FUSE ? 4 : 1], ci[FUSE ? 3 : 1][FUSE ? 4 : 1];
    f32x4* const ci_lds = reinterpret_cast<f32x4*>(smem + G3_LDS(NB, WN)) + tid;                    // [nt][512 threads]
    if (FUSE) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                v[mt][nt] = f32x4{args.p.v_leak, args.p.v_leak, args.p.v_leak, args.p.v_leak};
                if (mt < 3) ci[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                else ci_lds[nt * 512] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
    }
    const int n_total = n_steps * Kc;

    // Software pipeline over the chunk sequence c = (t, kc).  During chunk c:
    //   the spike word of chunk c+3 is fetched from global memory (register),
    //   the spike word of chunk c+2 (fetched during chunk c-1) and, by LDS-DMA, the weight planes of chunk c+2 go
    //   into ring slot (c-1) mod NB (spike words: slot of chunk c+2),
    //   the weight fragments of chunk c are read PD groups ahead of their MFMAs, the first ones of chunk c+1 and
    //   its spike words / table fragments at the end of chunk c (that slot has been complete since the last barrier).
    // One barrier per chunk; s_sched_barrier pins one fragment read + 4 MFMAs per group.
#ifdef SNN_EXP_FILL_RING           // timing only: random bf16 bits in the whole ring (for the no-copy experiment)
    for (int i = tid; i < NB * SLOT / 4; i += 512) {
        uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        reinterpret_cast<uint32_t*>(ring)[i] = (h & 0x3fff3fffu) | 0x38003800u;      // two bf16 of magnitude ~1e-5 .. 1
    }
    __syncthreads();
#endif
    uint32_t w_hold, w_new;
    {
        uint32_t w0[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fetch_next(w0[j]);  // spike words of chunks 0, 1 -> slots 0, 1
        fetch_next(w_hold);                             // chunk 2
#pragma unroll
        for (int j = 0; j < NB - 1; ++j) stage_next(j * SLOT);      // weight planes of chunks 0 .. NB-2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            asm volatile("" : "+v"(w0[j]));             // the loaded value is only defined from here on
            store_w(w0[j], j * SLOT);
        }
        asm volatile("" : "+v"(w_hold));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    bf16x8 af[2][MT], bq[RING];
    uint32_t wq[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wq[mt] = rd_w(0, mt);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[0][mt] = rd_a(wq[mt]);
#pragma unroll
    for (int g = 0; g < PD; ++g) bq[g] = rd_b(0, g);

    // ring slots (byte offsets) of chunks c, c+1, c+2 (receives its spike words now) and c+NB-1 (receives its planes)
    uint32_t o_cur = 0, o_nxt = SLOT, o_nn = 2 * SLOT, o_wr = (NB - 1) * SLOT;
    int kc = 0, t = 0;
    for (int c0 = 0; c0 < n_total; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (c0 + u >= n_total) break;
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                const int gp = g + PD;
                bq[gp % RING] = gp < 12 ? rd_b(o_cur, gp) : rd_b(o_nxt, gp - 12);
                if (g == 4) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) wq[mt] = rd_w(o_nxt, mt);
                }
                if (g >= 8 && g - 8 < MT) af[u ^ 1][g - 8] = rd_a(wq[g - 8]);
#ifndef SNN_EXP_NO_STORE_A
                if (g == 0) store_w(w_hold, o_nn);
#endif
                if (g == 2) {
                    fetch_next(w_new);
#ifndef SNN_EXP_NO_GLDS
                    stage_next(o_wr);
#endif
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][g / 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u][mt], bq[g % RING], acc[mt][g / 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef SNN_EXP_NO_BARRIER
            // weight planes landed (vmcnt), spike words written (lgkmcnt), everyone done reading slot `sl`.
            // The s_waitcnt builtin (not inline asm) so that hipcc's own wait-count bookkeeping knows the prefetched
            // fragments have arrived; the empty asm statements are compiler fences (neither builtin orders memory
            // accesses for hipcc, which otherwise moves LDS reads across the barrier).
            asm volatile("" ::: "memory");
            // NB = 4: this chunk's own copies (the youngest vector-memory operations of the wave: 3, or 2 / 1 on the
            // 8 x 1 wave grid) may stay in flight across the barrier
            if (NB == 3) __builtin_amdgcn_s_waitcnt(0x0070);        // vmcnt(0) lgkmcnt(0)
            else if (WN == 2) __builtin_amdgcn_s_waitcnt(0x0073);   // vmcnt(3) lgkmcnt(0)
            else if (wave + 8 < NPIECE) __builtin_amdgcn_s_waitcnt(0x0072);   // vmcnt(2)
            else __builtin_amdgcn_s_waitcnt(0x0071);                // vmcnt(1)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#endif
            asm volatile("" : "+v"(w_new));
            w_hold = w_new;
            if (NB == 3) { const uint32_t o = o_cur; o_cur = o_nxt; o_nxt = o_nn; o_nn = o; o_wr = o; }
            else { const uint32_t o = o_cur; o_cur = o_nxt; o_nxt = o_nn; o_nn = o_wr; o_wr = o; }
            const bool step_done = ++kc == Kc;
            if (step_done) kc = 0;
            if (FUSE && step_done) {
                // ---- LIF epilogue in registers.  A ballot over accumulator register (mt, nt, reg) holds, for each
                // of the 4 row groups rg, 16 channel bits of position mt*16 + rg*4 + reg; N-tiles (0,1) and (2,3)
                // pair up into the two 32-channel words of that position, which lane = position finally stores ----
                uint32_t my0 = 0, my1 = 0;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int np = 0; np < 2; ++np) {                   // N-tile pair -> word np of the position
                        f32x4 vd[2], d[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int nt = 2 * np + q;
                            if (mt < 3) {
                                lif_decay4(acc[mt][nt], v[mt][nt], ci[mt][nt], args.p, vd[q], d[q]);
                            } else {
                                f32x4 i3 = ci_lds[nt * 512];
                                lif_decay4(acc[mt][nt], v[mt][nt], i3, args.p, vd[q], d[q]);
                                ci_lds[nt * 512] = i3;
                            }
                            acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {                  // two ballots live at a time
                            const bool z0 = d[0][r] > 0.0f, z1 = d[1][r] > 0.0f;
                            const unsigned long long b0 = __ballot(z0), b1 = __ballot(z1);
                            v[mt][2 * np][r] = z0 ? args.p.v_reset : vd[0][r];
                            v[mt][2 * np + 1][r] = z1 ? args.p.v_reset : vd[1][r];
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                const uint32_t w = (uint32_t)((b0 >> (16 * rg)) & 0xffffull) | ((uint32_t)((b1 >> (16 * rg)) & 0xffffull) << 16);
                                // lane (mt*16 + rg*4 + r) keeps the two words of its position (rows >= M are never stored)
                                if (np == 0) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(my0) : "s"(w), "n"(mt * 16 + rg * 4 + r));
                                else         asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(my1) : "s"(w), "n"(mt * 16 + rg * 4 + r));
                            }
                        }
                    }
                {
                    const int row = m0 + wm * 64 + lane;                  // lane = position within the wave's 64 rows
                    const int word0 = (nb * BN + wn * 64) >> 5;
                    uint32_t* dst = args.spk + (size_t)t * args.spk_stride + (size_t)row * (Np >> 5) + word0;
                    if (row < M) {
                        if (word0 * 32 < Np) dst[0] = my0;
                        if ((word0 + 1) * 32 < Np) dst[1] = my1;
                    }
                }
                ++t;
            }
        }
    }
    if (FUSE) return;
    if (TILE) {
        // ---- LIF over the T time steps held in this tile.  The accumulators are the complete input currents
        // cur[t][position][column] of pb positions; in two passes of CG = 32*WN columns they go through LDS (the ring
        // is free now), where each thread runs neurons over t and the wave ballot of a step is the spike word(s):
        //   WN = 2 (CG = 64): lane = column, wave w takes positions w, w+8, ...; ballot = the word pair of (t, position)
        //   WN = 1 (CG = 32): lane = (position parity, column), wave w takes the position pairs; ballot = one word of
        //                     the even position (low half) and one of the odd position (high half)
        constexpr int CG = G3_TILE_CG(WN), PITCH = CG + 4;
        float* const tile = reinterpret_cast<float*>(smem);
        const int pb = args.pb, T = args.T;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // staged-ahead copies of chunks past the end have landed
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            __syncthreads();                               // ring reads done / previous pass consumed
            if (WN == 1 || wn == h) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nq = 0; nq < CG / 16; ++nq) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float val;
                            if (WN == 1) val = h == 0 ? acc[mt][nq][r] : acc[mt][2 + nq][r];
                            else val = acc[mt][nq][r];
                            tile[(wm * WROWS + mt * 16 + lg * 4 + r) * PITCH + nq * 16 + lr] = val;
                        }
                    }
            }
            __syncthreads();
            const int word0 = (nb * BN + h * CG) >> 5;     // first output word of this pass
            if (word0 * 32 >= Np) continue;                // block-uniform
            if (WN == 2) {
                const bool two = (word0 + 1) * 32 < Np;
                for (int pi = wave; pi < pb; pi += 8) {    // wave-uniform
                    const int pos = m0 + pi;
                    if (pos >= M) break;
                    float vv = args.p.v_leak, ii = 0.0f;
                    uint32_t my0 = 0, my1 = 0;             // lane t keeps the word pair of time step t
                    const float* src = tile + pi * PITCH + lane;
                    for (int t = 0; t < T; ++t) {
                        const bool z = lif_step(src[(size_t)t * pb * PITCH], vv, ii, args.p);
                        const unsigned long long b = __ballot(z);
                        my0 = lane == t ? (uint32_t)b : my0;
                        my1 = lane == t ? (uint32_t)(b >> 32) : my1;
                    }
                    if (lane < T) {
                        uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + (size_t)pos * (Np >> 5) + word0;
                        dst[0] = my0;
                        if (two) dst[1] = my1;
                    }
                }
            } else {
                const int par = lane >> 5, col = lane & 31;
                for (int pp = wave; 2 * pp < pb; pp += 8) {            // wave-uniform: position pair pp
                    const int pi = 2 * pp + par;                       // this half-wave's position
                    const bool live = pi < pb && m0 + pi < M;
                    if (m0 + 2 * pp >= M) break;
                    float vv = args.p.v_leak, ii = 0.0f;
                    uint32_t my0 = 0, my1 = 0;             // lane t keeps the words of (t, even position), (t, odd position)
                    const float* src = tile + (live ? pi : 2 * pp) * PITCH + col;
                    for (int t = 0; t < T; ++t) {
                        const bool z = lif_step(src[(size_t)t * pb * PITCH], vv, ii, args.p);
                        const unsigned long long b = __ballot(z);
                        my0 = lane == t ? (uint32_t)b : my0;
                        my1 = lane == t ? (uint32_t)(b >> 32) : my1;
                    }
                    if (lane < T) {
                        uint32_t* dst = args.spk + (size_t)lane * args.spk_stride + (size_t)(m0 + 2 * pp) * (Np >> 5) + word0;
                        dst[0] = my0;
                        if (2 * pp + 1 < pb && m0 + 2 * pp + 1 < M) dst[Np >> 5] = my1;
                    }
                }
            }
        }
        return;
    }
    // ---- store currents: per instruction 4 rows x 16 columns (64-B row segments) ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = nb * BN + wn * 64 + nt * 16 + lr;
            if (col >= Np) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * WROWS + mt * 16 + lg * 4 + r;
                if (m < M) args.out[(size_t)m * args.ldo + col] = acc[mt][nt][r];
            }
        }
}

// fp32 weights -> three bf16 planes [3][Kc][Np][32]  (hi = rn(w), mid = rn(w - hi), lo = rn(w - hi - mid): exact)
__device__ __forceinline__ uint16_t f2bf_rn(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }

__global__ void k_pack_bf16x3(const float* __restrict__ src, uint16_t* __restrict__ dst, int mode, int K, int N,
                              int Kc, int Np, int Cin, int Cp) {
    const size_t plane = (size_t)Kc * Np * 32;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < plane; idx += (size_t)gridDim.x * blockDim.x) {
        const int kk = idx & 31;
        const size_t rest = idx >> 5;
        const int n = (int)(rest % Np);
        const int kc = (int)(rest / Np);
        const int k = kc * 32 + kk;
        float w = 0.0f;
        if (n < N) {
            if (mode == PACK_CONV3X3) {
                const int tap = k / Cp, ci = k % Cp;
                if (ci < Cin) w = src[((size_t)n * Cin + ci) * 9 + tap];
            } else if (k < K) {
                w = src[(size_t)n * K + k];
            }
        }
        const uint16_t hi = f2bf_rn(w);
        const float r1 = __fsub_rn(w, bf2f(hi));
        const uint16_t mid = f2bf_rn(r1);
        const float r2 = __fsub_rn(r1, bf2f(mid));
        const uint16_t lo = f2bf_rn(r2);
        dst[idx] = hi; dst[plane + idx] = mid; dst[2 * plane + idx] = lo;
    }
}

// ------------------------------------------------------------------------------------------------
// K4: LIF scan over T of currents cur[T][R][ldc] -> spike planes [T][R][Nw]  (+ per-row counts)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lif_scan(const float* __restrict__ cur, int T, int R, int N, int Nw,
                                                  int ldc, NeuronP p, uint32_t* __restrict__ spk,
                                                  size_t spk_stride, uint32_t* __restrict__ row_counts) {
    const size_t Np = (size_t)Nw * 32;
    const size_t total = (size_t)R * Np;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = e < total;
    const size_t r = in ? e / Np : 0;
    const int n = in ? (int)(e % Np) : 0;
    const bool live = in && n < N;
    const float* c = cur + r * ldc + n;
    const size_t tstride = (size_t)R * ldc;
    float v = p.v_leak, i = 0.0f;
    const int lane = threadIdx.x & 63;
    uint32_t cnt = 0;
    for (int t = 0; t < T; ++t) {
        const float x = live ? c[(size_t)t * tstride] : 0.0f;
        const bool z = lif_step(x, v, i, p) && live;
        const unsigned long long m = __ballot(z);
        if ((lane & 31) == 0 && in) {
            const uint32_t w = (lane < 32) ? (uint32_t)m : (uint32_t)(m >> 32);
            spk[(size_t)t * spk_stride + (e >> 5)] = w;
            cnt += __popc(w);
        }
    }
    if (row_counts != nullptr && (lane & 31) == 0 && in && cnt != 0) atomicAdd(&row_counts[r], cnt);
}

// ------------------------------------------------------------------------------------------------
// K5: both leaky-integrator heads, time-collapsed.  The LI cell and the bias-free 1x1 conv / linear
// in front of it are linear and only the last membrane is used (rpn.py:118-119,
// faster_rcnn.py:513-514), so  mem_T = W . (sum_t kappa_last[t] * spk_t)  and the spike-rate
// variant's sum over t of the membranes is  W . (sum_t kappa_sum[t] * spk_t).
// ------------------------------------------------------------------------------------------------
struct Kappa { float last[SNN_MAX_STEPS]; float sum[SNN_MAX_STEPS]; };

// Block = 256 threads = RB rows; the reduction runs in slabs of HEADS_KS channels:
//   phase 1  thread = (row, 32-channel word): S[row][k] = sum_t kappa[t] * bit_t(row, k)  -> LDS (fp32)
//   phase 2  thread = (row, group of 4 outputs): acc4 += S[row][k] * Wt[k][4jg..4jg+3]   (Wt slab in LDS)
// NOp (outputs rounded up to 16) * RB / 4 <= 256.
#define HEADS_KS 128
template <int RB>
__global__ __launch_bounds__(256) void k_li_heads(const uint32_t* __restrict__ spk, size_t spk_stride, int T,
                                                  int M, int Kw, const float* __restrict__ wT, int NOp, int NA,
                                                  int NB, const Kappa kap, float* __restrict__ out_a,
                                                  float* __restrict__ out_b, float* __restrict__ sum_a,
                                                  float* __restrict__ sum_b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SST = HEADS_KS + 4;                          // padded row of S
    float* S_last = reinterpret_cast<float*>(smem);            // [RB][SST]
    const bool want_sum = (sum_a != nullptr);
    float* S_sum = S_last + RB * SST;                          // [RB][SST], present only if sums are requested
    float* Wl = S_last + (want_sum ? 2 : 1) * RB * SST;        // [HEADS_KS][NOp]
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * RB;
    const int JG = NOp / 4;                                    // output groups per row
    const int prow = tid / JG, pjg = tid % JG;                 // phase-2 role
    const bool pact = prow < RB;
    f32x4 acc_l = {0.f, 0.f, 0.f, 0.f}, acc_s = {0.f, 0.f, 0.f, 0.f};
    const int n_slabs = (Kw * 32 + HEADS_KS - 1) / HEADS_KS;
    for (int sl = 0; sl < n_slabs; ++sl) {
        const int w0 = sl * (HEADS_KS / 32);                   // first channel word of the slab
        const int nw = min(HEADS_KS / 32, Kw - w0);
        // weights of the slab -> LDS (coalesced float4 copy; rows beyond Kp are never touched)
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(wT + (size_t)w0 * 32 * NOp);
            f32x4* dst = reinterpret_cast<f32x4*>(Wl);
            for (int i = tid; i < nw * 32 * NOp / 4; i += 256) dst[i] = src[i];
        }
        // phase 1: item = (row, channel word, byte of the word) - 8 channels each, so that small row blocks (detector
        // heads: RB = 16) still give every thread an item
        for (int item = tid; item < RB * (HEADS_KS / 32) * 4; item += 256) {
            const int q = item & 3, wi = (item >> 2) % (HEADS_KS / 32), row = (item >> 2) / (HEADS_KS / 32);
            const int m = m0 + row;
            float sl_[8], ss_[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) { sl_[b] = 0.f; ss_[b] = 0.f; }
            if (m < M && wi < nw) {
                const uint32_t* wp = spk + (size_t)m * Kw + w0 + wi;
#pragma unroll 4
                for (int t = 0; t < T; ++t) {
                    const uint32_t w = wp[(size_t)t * spk_stride] >> (8 * q);
                    const float kl = kap.last[t], ks = kap.sum[t];
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const float bit = (float)((w >> b) & 1u);
                        sl_[b] = fmaf(bit, kl, sl_[b]);            // exact: bit is 0 or 1
                        if (want_sum) ss_[b] = fmaf(bit, ks, ss_[b]);
                    }
                }
            }
            f32x4* dl = reinterpret_cast<f32x4*>(S_last + row * SST + wi * 32 + q * 8);
            f32x4* ds = reinterpret_cast<f32x4*>(S_sum + row * SST + wi * 32 + q * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                dl[j] = f32x4{sl_[4 * j], sl_[4 * j + 1], sl_[4 * j + 2], sl_[4 * j + 3]};
                if (want_sum) ds[j] = f32x4{ss_[4 * j], ss_[4 * j + 1], ss_[4 * j + 2], ss_[4 * j + 3]};
            }
        }
        __syncthreads();
        // phase 2
        if (pact) {
            const float* sl_row = S_last + prow * SST;
            const float* ss_row = S_sum + prow * SST;
            const float* wcol = Wl + 4 * pjg;
            const int kn = nw * 32;
#pragma unroll 4
            for (int k = 0; k < kn; ++k) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wcol + (size_t)k * NOp);
                const float a = sl_row[k];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_l[r] = fmaf(a, w[r], acc_l[r]);
                if (want_sum) {
                    const float c = ss_row[k];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc_s[r] = fmaf(c, w[r], acc_s[r]);
                }
            }
        }
        __syncthreads();
    }
    const int m = m0 + prow;
    if (pact && m < M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 4 * pjg + r;
            if (j < NA) { out_a[(size_t)m * NA + j] = acc_l[r]; if (want_sum) sum_a[(size_t)m * NA + j] = acc_s[r]; }
            else if (j < NA + NB) {
                out_b[(size_t)m * NB + (j - NA)] = acc_l[r];
                if (want_sum) sum_b[(size_t)m * NB + (j - NA)] = acc_s[r];
            }
        }
    }
}

// K5b: the same heads on the bf16 matrix cores.  mem_T = sum_t kappa[t] * (spk_t . W): each  spk_t . W  is an exact
// bf16x3 spike GEMM (spikes {0,1}, W = hi + mid + lo, fp32 accumulate) with all NOp <= 64 outputs in 1-4 MFMA column
// tiles, and the kappa-weighted sum over t is an fma chain in the epilogue.  Wave = 16 rows x all T steps; the A
// fragments come from the byte -> 8 bf16 table (as in k_gemm_bf16x3), the weights are split into their three bf16
// planes while they are staged into LDS (no second packed copy): resident when all of W fits, else streamed per
// 32-deep chunk (double-buffered).  Time steps go in groups of 8 (8 x NT accumulators).
// RPN heads (196k rows, K=256, 15 outputs): 137 -> ~40 us;  detector heads (2000 rows, K=1024, 45 outputs): 88 -> ~15 us.
#define LIH_TG 8
struct LiHeadsArgs {
    const uint32_t* spk; unsigned long long spk_stride;
    const float* wT;              // [Kp][NOp] fp32 (snn_pack_heads_weight)
    float *out_a, *out_b, *sum_a, *sum_b;
    int T, M, Kw, NOp, NA, NB, n_groups, resident;
    Kappa kap;
};

template <int NT>
__global__ __launch_bounds__(256) void k_li_heads_mfma(const LiHeadsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lut = smem;                           // 4 KB
    unsigned char* const bbase = smem + G3_LUT_BYTES;          // [chunk slot][3][NOp][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4, lg8 = 8 * lg;
    const int NOp = a.NOp, Kc = a.Kw;
    const uint32_t slot_bytes = 3u * NOp * 64u;
    {
        uint4 q;
        q.x = bf16_pair(tid, 0); q.y = bf16_pair(tid, 1); q.z = bf16_pair(tid, 2); q.w = bf16_pair(tid, 3);
        *reinterpret_cast<uint4*>(lut + tid * 16) = q;
    }
    auto stage = [&](int kc, int slot) {                       // split W[32kc .. 32kc+31][NOp] into 3 bf16 planes
        unsigned char* dst = bbase + (size_t)slot * slot_bytes;
        for (int item = tid; item < 16 * NOp; item += 256) {
            const int n = item % NOp, kp = item / NOp;
            const float* src = a.wT + (size_t)(32 * kc + 2 * kp) * NOp + n;
            uint32_t pl[3] = {0u, 0u, 0u};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float w = src[(size_t)h * NOp];
                const uint16_t hi = f2bf_rn(w);
                const float r1 = __fsub_rn(w, bf2f(hi));
                const uint16_t mid = f2bf_rn(r1);
                const uint16_t lo = f2bf_rn(__fsub_rn(r1, bf2f(mid)));
                pl[0] |= (uint32_t)hi << (16 * h); pl[1] |= (uint32_t)mid << (16 * h); pl[2] |= (uint32_t)lo << (16 * h);
            }
            const int off = n * 64 + ((((kp >> 2) ^ G3_SWZ(n)) << 4) | ((kp & 3) << 2));
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<uint32_t*>(dst + q * NOp * 64 + off) = pl[q];
        }
    };
    if (a.resident)
        for (int kc = 0; kc < Kc; ++kc) stage(kc, kc);
    __syncthreads();
    const unsigned char* const b_rd = bbase + lr * 64 + ((lg ^ G3_SWZ(lr)) << 4);
    for (int g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
        const int m0 = (g * 4 + wave) * 16;
        const int mrow = min(m0 + lr, a.M - 1);                 // rows past M: recomputed, never stored
        const uint32_t* wsrc = a.spk + (size_t)mrow * a.Kw;
        f32x4 o_last[NT], o_sum[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { o_last[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; o_sum[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int tg0 = 0; tg0 < a.T; tg0 += LIH_TG) {
            const int tn = min(LIH_TG, a.T - tg0);              // block-uniform
            f32x4 acc[LIH_TG][NT];
#pragma unroll
            for (int t = 0; t < LIH_TG; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto chunk = [&](int kc, const uint32_t (&w_cur)[LIH_TG]) {
                const unsigned char* bs = b_rd + (size_t)(a.resident ? kc : (kc & 1)) * slot_bytes;
                bf16x8 b[3][NT];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        b[pl][nt] = *reinterpret_cast<const bf16x8*>(bs + (pl * NOp + nt * 16) * 64);
#pragma unroll
                for (int t = 0; t < LIH_TG; ++t) {
                    if (t < tn) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(lut + (__builtin_amdgcn_ubfe(w_cur[t], lg8, 8) << 4));
#pragma unroll
                        for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b[pl][nt], acc[t][nt], 0, 0, 0);
                    }
                }
            };
            if (a.resident && Kc == 8) {
                // C = 256: the 8 spike words of a (t, row) are one 32-byte line - all T x 8 words are requested up
                // front (one memory latency per row tile instead of one per chunk)
                uint4 wl[LIH_TG][2];
#pragma unroll
                for (int t = 0; t < LIH_TG; ++t) {
                    const uint4* q = reinterpret_cast<const uint4*>(wsrc + (size_t)(tg0 + (t < tn ? t : 0)) * a.spk_stride);
                    wl[t][0] = q[0]; wl[t][1] = q[1];
                }
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    uint32_t w_cur[LIH_TG];
#pragma unroll
                    for (int t = 0; t < LIH_TG; ++t) {
                        const uint4 v = wl[t][kc >> 2];
                        w_cur[t] = (kc & 3) == 0 ? v.x : (kc & 3) == 1 ? v.y : (kc & 3) == 2 ? v.z : v.w;
                    }
                    chunk(kc, w_cur);
                }
            } else {
                uint32_t w_nxt[LIH_TG];
#pragma unroll
                for (int t = 0; t < LIH_TG; ++t) w_nxt[t] = t < tn ? wsrc[(size_t)(tg0 + t) * a.spk_stride] : 0u;
                if (!a.resident) { stage(0, 0); __syncthreads(); }
                for (int kc = 0; kc < Kc; ++kc) {
                    uint32_t w_cur[LIH_TG];
#pragma unroll
                    for (int t = 0; t < LIH_TG; ++t) w_cur[t] = w_nxt[t];
                    if (kc + 1 < Kc) {
#pragma unroll
                        for (int t = 0; t < LIH_TG; ++t) w_nxt[t] = t < tn ? wsrc[(size_t)(tg0 + t) * a.spk_stride + kc + 1] : 0u;
                        if (!a.resident) stage(kc + 1, (kc + 1) & 1);
                    }
                    chunk(kc, w_cur);
                    if (!a.resident) __syncthreads();           // chunk kc+1 staged, chunk kc consumed
                }
            }
#pragma unroll
            for (int t = 0; t < LIH_TG; ++t)
                if (t < tn) {
                    const float kl = a.kap.last[tg0 + t], ks = a.kap.sum[tg0 + t];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o_last[nt][r] = fmaf(kl, acc[t][nt][r], o_last[nt][r]);
                            o_sum[nt][r] = fmaf(ks, acc[t][nt][r], o_sum[nt][r]);
                        }
                }
        }
        // lane holds rows lg*4 + r, output column nt*16 + lr
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int j = nt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + lg * 4 + r;
                if (m >= a.M) continue;
                if (j < a.NA) { a.out_a[(size_t)m * a.NA + j] = o_last[nt][r]; if (a.sum_a) a.sum_a[(size_t)m * a.NA + j] = o_sum[nt][r]; }
                else if (j < a.NA + a.NB) {
                    a.out_b[(size_t)m * a.NB + (j - a.NA)] = o_last[nt][r];
                    if (a.sum_b) a.sum_b[(size_t)m * a.NB + (j - a.NA)] = o_sum[nt][r];
                }
            }
        }
    }
}

// K5c: the same heads when W does not fit in LDS (detector: K = 1024, 45 outputs) and there are few rows (2000): one
// work-group per 16-row tile, its 4 waves split the reduction into quarters (so 125 work-groups of independent waves
// instead of 32 that march through K together).  A wave keeps the accumulators of ALL time steps (T <= 16; 12 with 4 column
// tiles), builds the
// three bf16 planes of its weight fragments in registers straight from the fp32 W^T (read through L2, next chunk's
// values requested before this chunk's MFMAs), and needs no barrier until the four partial results meet in LDS and are
// added in wave order (deterministic).  2000 x 1024 x 45, T = 12: 68 us (fp32 VALU kernel) -> ~15 us.
#define LIH_KS_TM(nt) ((nt) <= 3 ? 16 : 12)     // time steps whose accumulators fit in registers beside NT column tiles
template <int NT>
__global__ __launch_bounds__(256) void k_li_heads_ksplit(const LiHeadsArgs a) {
    constexpr int TM = LIH_KS_TM(NT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lut = smem;                           // 4 KB
    f32x4* const red = reinterpret_cast<f32x4*>(smem + G3_LUT_BYTES);      // [4 waves][2][NT][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4, lg8 = 8 * lg;
    const int NOp = a.NOp, Kc = a.Kw, T = a.T;
    {
        uint4 q;
        q.x = bf16_pair(tid, 0); q.y = bf16_pair(tid, 1); q.z = bf16_pair(tid, 2); q.w = bf16_pair(tid, 3);
        *reinterpret_cast<uint4*>(lut + tid * 16) = q;
    }
    __syncthreads();
    const int m0 = blockIdx.x * 16;
    const int mrow = min(m0 + lr, a.M - 1);                    // rows past M: recomputed, never stored
    const uint32_t* wsrc = a.spk + (size_t)mrow * a.Kw;
    const int c0 = wave * Kc / 4, c1 = (wave + 1) * Kc / 4;    // this wave's chunks
    f32x4 acc[TM][NT];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // element (k = 32 kc + 8 lg + j, n = nt*16 + lr) of W^T: the lane's B fragment is j = 0..7
    const float* const wlane = a.wT + (size_t)lg8 * NOp + lr;
    float wf[NT][8];
    uint32_t w_nxt[TM];
    auto request = [&](int kc) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[nt][j] = wlane[(size_t)(32 * kc + j) * NOp + nt * 16];
#pragma unroll
        for (int t = 0; t < TM; ++t) w_nxt[t] = t < T ? wsrc[(size_t)t * a.spk_stride + kc] : 0u;
    };
    if (c0 < c1) request(c0);
    for (int kc = c0; kc < c1; ++kc) {
        bf16x8 b[3][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = wf[nt][j];
                const uint16_t hi = f2bf_rn(w);
                const float r1 = __fsub_rn(w, bf2f(hi));
                const uint16_t mid = f2bf_rn(r1);
                const uint16_t lo = f2bf_rn(__fsub_rn(r1, bf2f(mid)));
                b[0][nt][j] = (short)hi; b[1][nt][j] = (short)mid; b[2][nt][j] = (short)lo;
            }
        uint32_t w_cur[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) w_cur[t] = w_nxt[t];
        if (kc + 1 < c1) request(kc + 1);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            if (t < T) {                                        // block-uniform
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(lut + (__builtin_amdgcn_ubfe(w_cur[t], lg8, 8) << 4));
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b[pl][nt], acc[t][nt], 0, 0, 0);
            }
        }
    }
    f32x4 o_last[NT], o_sum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { o_last[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; o_sum[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < TM; ++t)
        if (t < T) {
            const float kl = a.kap.last[t], ks = a.kap.sum[t];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o_last[nt][r] = fmaf(kl, acc[t][nt][r], o_last[nt][r]);
                    o_sum[nt][r] = fmaf(ks, acc[t][nt][r], o_sum[nt][r]);
                }
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        red[((wave * 2 + 0) * NT + nt) * 64 + lane] = o_last[nt];
        red[((wave * 2 + 1) * NT + nt) * 64 + lane] = o_sum[nt];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f32x4 ol = red[(0 * NT + nt) * 64 + lane], os = red[(1 * NT + nt) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 pl = red[((w * 2 + 0) * NT + nt) * 64 + lane], ps = red[((w * 2 + 1) * NT + nt) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) { ol[r] = __fadd_rn(ol[r], pl[r]); os[r] = __fadd_rn(os[r], ps[r]); }
        }
        // lane holds rows lg*4 + r, output column nt*16 + lr
        const int j = nt * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + lg * 4 + r;
            if (m >= a.M) continue;
            if (j < a.NA) { a.out_a[(size_t)m * a.NA + j] = ol[r]; if (a.sum_a) a.sum_a[(size_t)m * a.NA + j] = os[r]; }
            else if (j < a.NA + a.NB) {
                a.out_b[(size_t)m * a.NB + (j - a.NA)] = ol[r];
                if (a.sum_b) a.sum_b[(size_t)m * a.NB + (j - a.NA)] = os[r];
            }
        }
    }
}

// spikes per image of one level, counted from the bit-planes (spike-rate mode of the bf16x3 path): blockIdx.x = image,
// blockIdx.y = slice of the image's words; integer atomics, so the result does not depend on the order.
// counts must be zeroed by the caller.
__global__ __launch_bounds__(256) void k_count_spikes(const uint32_t* __restrict__ spk, unsigned long long spk_stride,
                                                      int T, int words_per_image,
                                                      unsigned long long* __restrict__ counts) {
    unsigned long long sum = 0;
    for (int t = 0; t < T; ++t) {
        const uint32_t* src = spk + (size_t)t * spk_stride + (size_t)blockIdx.x * words_per_image;
        for (int i = blockIdx.y * 256 + threadIdx.x; i < words_per_image; i += gridDim.y * 256) sum += __popc(src[i]);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&counts[blockIdx.x], part[0] + part[1] + part[2] + part[3]);
}

// spikes per row (RoI) over all T planes: one wave per row  (spike-rate mode of the fused linear layers)
__global__ __launch_bounds__(256) void k_count_rows(const uint32_t* __restrict__ spk, unsigned long long spk_stride, int T,
                                                    int R, int words, uint32_t* __restrict__ counts) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    uint32_t sum = 0;
    for (int t = 0; t < T; ++t) {
        const uint32_t* src = spk + (size_t)t * spk_stride + (size_t)row * words;
        for (int i = lane; i < words; i += 64) sum += __popc(src[i]);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0) counts[row] = sum;
}

// ------------------------------------------------------------------------------------------------
// Greedy (batched) NMS for the callers on either side of the heads (rpn.py:517, roi_heads.py:1160-1161):
// boxes arrive sorted by decreasing score; k_nms_mask builds the suppression bit-matrix (box j > i, same
// category, IoU > thr), k_nms_scan walks it in score order with the rows staged through LDS 64 at a time.
// ------------------------------------------------------------------------------------------------
// Batched form (blockIdx.z = image): per-image strides (elements) and a device-side candidate count; a plain call
// passes zero strides and n_dev = nullptr.
struct NmsBatch { const int* n_dev; long long boxes_stride, cat_stride, mask_stride, keep_stride; };

__global__ __launch_bounds__(64) void k_nms_mask(const float* __restrict__ boxes, const int* __restrict__ cat, int n,
                                                 float thr, unsigned long long* __restrict__ mask, int words,
                                                 const NmsBatch nb) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;                                  // only j > i matters
    if (nb.n_dev) n = nb.n_dev[blockIdx.z];
    if (rb * 64 >= n || cb * 64 >= n) return;
    boxes += (size_t)blockIdx.z * nb.boxes_stride;
    if (cat) cat += (size_t)blockIdx.z * nb.cat_stride;
    mask += (size_t)blockIdx.z * nb.mask_stride;
    __shared__ float cbx[64][4];
    __shared__ int ccat[64];
    const int t = threadIdx.x;
    const int j0 = cb * 64;
    if (j0 + t < n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cbx[t][q] = boxes[(size_t)(j0 + t) * 4 + q];
        ccat[t] = cat ? cat[j0 + t] : 0;
    }
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= n) return;
    const float x1 = boxes[(size_t)i * 4], y1 = boxes[(size_t)i * 4 + 1], x2 = boxes[(size_t)i * 4 + 2], y2 = boxes[(size_t)i * 4 + 3];
    const float area_i = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    const int ci = cat ? cat[i] : 0;
    unsigned long long bits = 0;
    const int jn = min(64, n - j0);
    for (int jj = (rb == cb ? t + 1 : 0); jj < jn; ++jj) {
        if (ccat[jj] != ci) continue;
        const float w = fmaxf(__fsub_rn(fminf(x2, cbx[jj][2]), fmaxf(x1, cbx[jj][0])), 0.0f);
        const float h = fmaxf(__fsub_rn(fminf(y2, cbx[jj][3]), fmaxf(y1, cbx[jj][1])), 0.0f);
        const float inter = __fmul_rn(w, h);
        const float area_j = __fmul_rn(__fsub_rn(cbx[jj][2], cbx[jj][0]), __fsub_rn(cbx[jj][3], cbx[jj][1]));
        const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_i, area_j), inter));     // box_iou's formula
        if (iou > thr) bits |= 1ull << jj;
    }
    mask[(size_t)i * words + cb] = bits;
}

// One work-group walks the candidates in score order, 64 (one mask word) per step.  Thread w < words owns word w of
// the "removed" set.  Per chunk c:
//   waves 1-3 copy the mask rows of chunk c+1 (words >= c+1 only: the upper triangle) into the other LDS buffer -
//     the rows do not depend on any decision, so the copy runs beside the walk;
//   wave 0 resolves the chunk's own 64x64 block in registers: lane b holds the diagonal word of row b, the walk is
//     64 scalar steps (v_readlane of a constant lane, s_or) with no memory access;
//   everyone ORs the rows of the kept boxes into the later words.
// (The first version staged each chunk with a blocking copy and read LDS inside the walk: 965 us for 4768 boxes,
// two thirds of the RPN post-processing time.)
// dbl = 0 (n > 9984: two buffers do not fit the LDS): one buffer, blocking copy at the top of each chunk.
__global__ __launch_bounds__(256) void k_nms_scan(const unsigned long long* __restrict__ mask, int n, int words,
                                                  int max_keep, int dbl, int* __restrict__ keep, int* __restrict__ n_keep,
                                                  const NmsBatch nb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* rows = reinterpret_cast<unsigned long long*>(smem);     // [1 + dbl][64][words]
    if (nb.n_dev) n = nb.n_dev[blockIdx.x];
    mask += (size_t)blockIdx.x * nb.mask_stride;
    keep += (size_t)blockIdx.x * nb.keep_stride;
    n_keep += blockIdx.x;
    if (n <= 0) { if (threadIdx.x == 0) *n_keep = 0; return; }
    __shared__ unsigned long long removed_cur, kept_cur;
    __shared__ int count_s;
    const int t = threadIdx.x;
    const int n_chunks = (n + 63) / 64;
    unsigned long long removed = 0;          // thread t (< words): removed bits of boxes [64t, 64t+64)
    auto copy_chunk = [&](int c, int first, int step) {       // rows of chunk c, words [c, words) -> buffer c & 1
        const int rn = min(64, n - c * 64), wn = words - c;
        unsigned long long* dst = rows + (size_t)(c & dbl) * 64 * words;
        const unsigned long long* src = mask + (size_t)c * 64 * words;
        for (int idx = first; idx < rn * wn; idx += step) {
            const int r = idx / wn, w = c + idx % wn;
            dst[r * words + w] = src[(size_t)r * words + w];
        }
    };
    copy_chunk(0, t, 256);
    if (t == 0) { count_s = 0; removed_cur = 0; }
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int rn = min(64, n - c * 64);
        const unsigned long long* cur = rows + (size_t)(c & dbl) * 64 * words;
        if (!dbl && c > 0) {
            copy_chunk(c, t, 256);
            __syncthreads();
        }
        if (t >= 64) {
            if (dbl && c + 1 < n_chunks) copy_chunk(c + 1, t - 64, 192);
        } else {                              // wave 0: the chunk's own block
            const unsigned long long diag = t < rn ? cur[t * words + c] : 0ull;
            const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
            unsigned long long rc = removed_cur;              // wave-uniform
            if (rn < 64) rc |= ~0ull << rn;                   // lanes past the end count as removed
            int count = count_s;
            const int base = count;
            unsigned long long kept_bits = 0;
#pragma unroll
            for (int b = 0; b < 64; ++b) {
                if (!((rc >> b) & 1ull) && count < max_keep) {
                    kept_bits |= 1ull << b;
                    ++count;
                    rc |= ((unsigned long long)__builtin_amdgcn_readlane(dhi, b) << 32) | __builtin_amdgcn_readlane(dlo, b);
                }
            }
            // kept boxes -> keep[] in order: lane b is kept box number base + popcount(kept_bits below b)
            if ((kept_bits >> t) & 1ull)
                keep[base + __popcll(kept_bits & ((1ull << t) - 1ull))] = c * 64 + t;
            if (t == 0) { count_s = count; kept_cur = kept_bits; }
        }
        __syncthreads();
        const unsigned long long kept = kept_cur;
        if (t < words && t > c) {             // later words: OR the rows of the kept boxes
            unsigned long long acc = removed, k = kept;
            while (k) {
                const int b = __ffsll((long long)k) - 1;
                k &= k - 1;
                acc |= cur[b * words + t];
            }
            removed = acc;
        }
        if (t == c + 1) removed_cur = removed;
        const bool done = count_s >= max_keep;
        __syncthreads();
        if (done) break;
    }
    if (t == 0) *n_keep = count_s;
}

// ------------------------------------------------------------------------------------------------
// RPN proposal selection (rpn.py:420-499 + 262-296 + the box coder), one call for the batch:
//   k_rpn_topk    per (level, image): the pre_nms_top_n largest logits by a 3-pass radix select (11+11+10 bits)
//   k_rpn_decode  per candidate: anchor from the level geometry, box decode, sigmoid, clip, size/score filters
//   k_rpn_sort    per image: bitonic sort of the candidates by decreasing score in LDS, gather into sorted order
//   k_nms_mask / k_nms_scan (batched over images, category = level)
//   k_rpn_output  kept boxes -> [N][post_nms_top_n] padded + counts
// ------------------------------------------------------------------------------------------------
#define RPN_MAX_ANCHORS 16
#define RPN_MAX_IMAGES 64
#define RPN_SORT_MAX 8192
struct RpnPostLevel {
    const float* logits;          // [N*H*W][A]   position-major (the head's own output buffer)
    const float* deltas;          // [N*H*W][4A]
    int H, W, n, k, koff;         // n = H*W*A elements per image, k = min(pre_nms_top_n, n), koff = first candidate slot
    float sh, sw;                 // anchor strides (image size // feature size)
    float base[RPN_MAX_ANCHORS * 4];
};
struct RpnPostArgs {
    RpnPostLevel lv[SNN_MAX_LEVELS];
    float img_h[RPN_MAX_IMAGES], img_w[RPN_MAX_IMAGES];
    int n_levels, N, A, Ktot, post_n;
    float score_thresh, min_size, clip;
    int* cand_idx;                // [N][Ktot] level-local element index
    float* cand_logit;            // [N][Ktot]
    float* boxes;                 // [N][Ktot][4] clipped
    float* pre;                   // [N][Ktot][4] decoded, un-clipped
    float* prob;                  // [N][Ktot]
    float* skey;                  // [N][Ktot] prob, or -1 for filtered candidates
    float* s_boxes; float* s_pre; float* s_prob; int* s_cat; int* n_valid;     // sorted by decreasing score
};

__device__ __forceinline__ uint32_t f2key(float f) {           // monotone: larger float -> larger key
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void k_rpn_topk(const RpnPostArgs a) {
    const RpnPostLevel& L = a.lv[blockIdx.x];
    const int img = blockIdx.y, tid = threadIdx.x;
    const float* src = L.logits + (size_t)img * L.n;
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t s_prefix, s_need, s_cnt_gt, s_cnt_eq;
    uint32_t prefix = 0, pmask = 0;           // bits of the k-th largest key decided so far
    uint32_t need = (uint32_t)L.k;            // how many are still to be taken among keys matching the prefix
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        for (int b = tid; b < 2048; b += 1024) hist[b] = 0;
        __syncthreads();
        const uint32_t bm = (1u << bits[pass]) - 1u;
        for (int e = tid; e < L.n; e += 1024) {
            const uint32_t key = f2key(src[e]);
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & bm], 1u);
        }
        __syncthreads();
        if (tid == 0) {                       // walk the bins from the top until `need` keys are covered
            uint32_t cum = 0;
            int b = (int)bm;
            for (; b > 0; --b) {
                if (cum + hist[b] >= need) break;
                cum += hist[b];
            }
            s_prefix = prefix | ((uint32_t)b << shifts[pass]);
            s_need = need - cum;              // taken from bin b (all bins above it are taken whole)
        }
        __syncthreads();
        prefix = s_prefix; need = s_need;
        pmask |= bm << shifts[pass];
        __syncthreads();
    }
    // prefix = key of the k-th largest logit; `need` of the keys equal to it are taken: those with the lowest element
    // index (deterministic; an ordered pass with a block scan, run only when there are more ties than needed)
    __shared__ uint32_t s_eq_total;
    if (tid == 0) { s_cnt_gt = 0; s_cnt_eq = 0; s_eq_total = hist[prefix & 1023u]; }
    __syncthreads();
    int* out_idx = a.cand_idx + (size_t)img * a.Ktot + L.koff;
    float* out_logit = a.cand_logit + (size_t)img * a.Ktot + L.koff;
    const uint32_t n_gt = (uint32_t)L.k - need;
    const bool ordered_ties = s_eq_total > need;
    for (int e = tid; e < L.n; e += 1024) {
        const float x = src[e];
        const uint32_t key = f2key(x);
        int slot = -1;
        if (key > prefix) slot = (int)atomicAdd(&s_cnt_gt, 1u);
        else if (key == prefix && !ordered_ties) slot = (int)(n_gt + atomicAdd(&s_cnt_eq, 1u));
        if (slot >= 0) { out_idx[slot] = e; out_logit[slot] = x; }
    }
    if (ordered_ties) {
        __shared__ uint32_t wsum[16];
        uint32_t taken = 0;                   // block-uniform
        for (int e0 = 0; e0 < L.n && taken < need; e0 += 1024) {
            const int e = e0 + tid;
            const bool tie = e < L.n && f2key(src[e]) == prefix;
            const unsigned long long bal = __ballot(tie);
            const int lane = tid & 63, wv = tid >> 6;
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int w = 0; w < 16; ++w) { if (w < wv) before += wsum[w]; total += wsum[w]; }
            const uint32_t rank = taken + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (tie && rank < need) { out_idx[n_gt + rank] = e; out_logit[n_gt + rank] = src[e]; }
            taken += total;
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void k_rpn_decode(const RpnPostArgs a) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= a.N * a.Ktot) return;
    const int img = g / a.Ktot, c = g % a.Ktot;
    int l = 0;
    while (l + 1 < a.n_levels && c >= a.lv[l + 1].koff) ++l;
    const RpnPostLevel& L = a.lv[l];
    const int e = a.cand_idx[g];
    const int pos = e / a.A, an = e % a.A;
    const int y = pos / L.W, x = pos % L.W;
    const float fx = (float)(x * (int)L.sw), fy = (float)(y * (int)L.sh);     // integer shifts, as the reference's
    const float ax1 = __fadd_rn(fx, L.base[4 * an]), ay1 = __fadd_rn(fy, L.base[4 * an + 1]);
    const float ax2 = __fadd_rn(fx, L.base[4 * an + 2]), ay2 = __fadd_rn(fy, L.base[4 * an + 3]);
    const float* d = L.deltas + ((size_t)img * L.H * L.W + pos) * 4 * a.A + 4 * an;
    // BoxCoder.decode_single, weights (1, 1, 1, 1)
    const float w = __fsub_rn(ax2, ax1), h = __fsub_rn(ay2, ay1);
    const float cx = __fadd_rn(ax1, __fmul_rn(0.5f, w)), cy = __fadd_rn(ay1, __fmul_rn(0.5f, h));
    const float dw = fminf(d[2], a.clip), dh = fminf(d[3], a.clip);
    const float pcx = __fadd_rn(__fmul_rn(d[0], w), cx), pcy = __fadd_rn(__fmul_rn(d[1], h), cy);
    const float hw = __fmul_rn(0.5f, __fmul_rn(expf(dw), w)), hh = __fmul_rn(0.5f, __fmul_rn(expf(dh), h));
    const float x1 = __fsub_rn(pcx, hw), y1 = __fsub_rn(pcy, hh), x2 = __fadd_rn(pcx, hw), y2 = __fadd_rn(pcy, hh);
    const float prob = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-a.cand_logit[g])));
    const float W_ = a.img_w[img], H_ = a.img_h[img];
    const float bx1 = fminf(fmaxf(x1, 0.0f), W_), by1 = fminf(fmaxf(y1, 0.0f), H_);
    const float bx2 = fminf(fmaxf(x2, 0.0f), W_), by2 = fminf(fmaxf(y2, 0.0f), H_);
    const bool valid = __fsub_rn(bx2, bx1) >= a.min_size && __fsub_rn(by2, by1) >= a.min_size && prob >= a.score_thresh;
    reinterpret_cast<float4*>(a.pre)[g] = make_float4(x1, y1, x2, y2);
    reinterpret_cast<float4*>(a.boxes)[g] = make_float4(bx1, by1, bx2, by2);
    a.prob[g] = prob;
    a.skey[g] = valid ? prob : -1.0f;
}

__global__ __launch_bounds__(1024) void k_rpn_sort(const RpnPostArgs a) {
    __shared__ unsigned long long v[RPN_SORT_MAX];
    __shared__ uint16_t slot_of[RPN_SORT_MAX];
    const int img = blockIdx.x, tid = threadIdx.x, K = a.Ktot;
    int np2 = 1;
    while (np2 < K) np2 <<= 1;
    // descending on (score, then level, then lower element index): the candidate's slot rides in the low 13 bits.
    // (The slots inside a level are filled in atomic order; the element index makes the result run-to-run identical.)
    for (int i = tid; i < np2; i += 1024) {
        unsigned long long key = 0ull;
        if (i < K) {
            int l = 0;
            while (l + 1 < a.n_levels && i >= a.lv[l + 1].koff) ++l;
            const uint32_t ident = ((uint32_t)l << 28) | (uint32_t)a.cand_idx[(size_t)img * K + i];    // e < 2^28
            key = ((unsigned long long)f2key(a.skey[(size_t)img * K + i]) << 32) | (uint32_t)(~ident);
        }
        v[i] = key;
        slot_of[i] = (uint16_t)i;
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long x = v[i], y = v[p];
                    const bool desc = (i & k) == 0;
                    if (desc ? x < y : x > y) {
                        v[i] = y; v[p] = x;
                        const uint16_t q = slot_of[i]; slot_of[i] = slot_of[p]; slot_of[p] = q;
                    }
                }
            }
            __syncthreads();
        }
    int valid = 0;
    for (int i = tid; i < K; i += 1024) {
        const int c = slot_of[i];
        const size_t src = (size_t)img * K + c, dst = (size_t)img * K + i;
        reinterpret_cast<float4*>(a.s_boxes)[dst] = reinterpret_cast<const float4*>(a.boxes)[src];
        reinterpret_cast<float4*>(a.s_pre)[dst] = reinterpret_cast<const float4*>(a.pre)[src];
        a.s_prob[dst] = a.prob[src];
        int l = 0;
        while (l + 1 < a.n_levels && c >= a.lv[l + 1].koff) ++l;
        a.s_cat[dst] = l;
        valid += a.skey[src] >= 0.0f;
    }
    __shared__ int s_valid;
    if (tid == 0) s_valid = 0;
    __syncthreads();
    if (valid) atomicAdd(&s_valid, valid);
    __syncthreads();
    if (tid == 0) a.n_valid[img] = s_valid;
}

__global__ __launch_bounds__(256) void k_rpn_output(const RpnPostArgs a, const int* __restrict__ keep, const int* __restrict__ n_keep,
                                                    float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                    int* __restrict__ out_counts) {
    const int img = blockIdx.x;
    const int cnt = min(n_keep[img], a.post_n);
    for (int r = threadIdx.x; r < a.post_n; r += 256) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        float sc = 0.f;
        if (r < cnt) {
            const size_t src = (size_t)img * a.Ktot + keep[(size_t)img * a.Ktot + r];
            b = reinterpret_cast<const float4*>(a.s_boxes)[src];
            sc = a.s_prob[src];
        }
        reinterpret_cast<float4*>(out_boxes)[(size_t)img * a.post_n + r] = b;
        out_scores[(size_t)img * a.post_n + r] = sc;
    }
    if (threadIdx.x == 0) out_counts[img] = cnt;
}

// ------------------------------------------------------------------------------------------------
// Detection post-processing (roi_heads.py:1075-1176, the reference's variant that also reports background boxes),
// one call for the batch.  Per image two candidate lists: foreground (RoI x class >= 1) and background (RoIs without
// any class above the score threshold, class-0 box); each list is sorted by decreasing score (ties: lower candidate
// index, as a stable sort), goes through NMS (category = class) and the kept boxes are written fg first, then bg.
//   k_det_candidates  per list slot: softmax, BoxCoder(10,10,5,5).decode, clip, score / size filters; also all_scores / all_boxes
//   k_sort_lists      per list: bitonic sort in LDS (<= 16384 slots), gather into score order
//   k_nms_mask / k_nms_scan (batched over the 2N lists),  k_det_output
// ------------------------------------------------------------------------------------------------
#define DET_SORT_MAX 16384
struct DetPostArgs {
    const float* logits;          // [R][K]
    const float* deltas;          // [R][4K]
    const float* props;           // [R][4]
    int roi_base[RPN_MAX_IMAGES + 1];
    float img_h[RPN_MAX_IMAGES], img_w[RPN_MAX_IMAGES];
    int N, K, Kcap, det_per_img, out_cap;
    float score_thresh, min_size, clip, wx, wy, ww, wh;
    float* all_scores; float* all_boxes;                       // [R][K], [R][K][4]
    float* boxes; float* skey; int* cat;                       // [2N][Kcap] candidate lists
    float* s_boxes; float* s_score; int* s_cat; int* n_valid;   // sorted
};

__global__ __launch_bounds__(256) void k_det_candidates(const DetPostArgs a) {
    const int list = blockIdx.y, img = list >> 1, bg = list & 1;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= a.Kcap) return;
    const int Ri = a.roi_base[img + 1] - a.roi_base[img];
    const int rl = bg ? slot : slot / (a.K - 1), k = bg ? 0 : slot % (a.K - 1) + 1;
    const size_t o = (size_t)list * a.Kcap + slot;
    if (rl >= Ri) { a.skey[o] = -1.0f; a.cat[o] = 0; reinterpret_cast<float4*>(a.boxes)[o] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const int r = a.roi_base[img] + rl;
    const float* lg = a.logits + (size_t)r * a.K;
    float mx = lg[0];
    for (int j = 1; j < a.K; ++j) mx = fmaxf(mx, lg[j]);
    float sum = 0.0f;
    bool has_fg_cand = false;                                   // filled below once the scores are known
    for (int j = 0; j < a.K; ++j) sum = __fadd_rn(sum, expf(__fsub_rn(lg[j], mx)));
    const float score = __fdiv_rn(expf(__fsub_rn(lg[k], mx)), sum);
    if (bg)
        for (int j = 1; j < a.K; ++j) has_fg_cand |= __fdiv_rn(expf(__fsub_rn(lg[j], mx)), sum) > a.score_thresh;
    // BoxCoder(weights).decode_single
    const float4 pb = reinterpret_cast<const float4*>(a.props)[r];
    const float* d = a.deltas + (size_t)r * 4 * a.K + 4 * k;
    const float w = __fsub_rn(pb.z, pb.x), h = __fsub_rn(pb.w, pb.y);
    const float cx = __fadd_rn(pb.x, __fmul_rn(0.5f, w)), cy = __fadd_rn(pb.y, __fmul_rn(0.5f, h));
    const float dx = __fdiv_rn(d[0], a.wx), dy = __fdiv_rn(d[1], a.wy);
    const float dw = fminf(__fdiv_rn(d[2], a.ww), a.clip), dh = fminf(__fdiv_rn(d[3], a.wh), a.clip);
    const float pcx = __fadd_rn(__fmul_rn(dx, w), cx), pcy = __fadd_rn(__fmul_rn(dy, h), cy);
    const float hw = __fmul_rn(0.5f, __fmul_rn(expf(dw), w)), hh = __fmul_rn(0.5f, __fmul_rn(expf(dh), h));
    const float W_ = a.img_w[img], H_ = a.img_h[img];
    const float x1 = fminf(fmaxf(__fsub_rn(pcx, hw), 0.0f), W_), y1 = fminf(fmaxf(__fsub_rn(pcy, hh), 0.0f), H_);
    const float x2 = fminf(fmaxf(__fadd_rn(pcx, hw), 0.0f), W_), y2 = fminf(fmaxf(__fadd_rn(pcy, hh), 0.0f), H_);
    a.all_scores[(size_t)r * a.K + k] = score;
    reinterpret_cast<float4*>(a.all_boxes)[(size_t)r * a.K + k] = make_float4(x1, y1, x2, y2);
    const bool big = __fsub_rn(x2, x1) >= a.min_size && __fsub_rn(y2, y1) >= a.min_size;
    const bool valid = big && (bg ? !has_fg_cand : score > a.score_thresh);
    reinterpret_cast<float4*>(a.boxes)[o] = make_float4(x1, y1, x2, y2);
    a.skey[o] = valid ? score : -1.0f;
    a.cat[o] = k;
}

// one block per list: order = decreasing (score, then lower slot); n_valid = candidates with score >= 0
__global__ __launch_bounds__(1024) void k_sort_lists(const float* __restrict__ skey, const float* __restrict__ boxes,
                                                     const int* __restrict__ cat, int Kcap, float* __restrict__ s_boxes,
                                                     float* __restrict__ s_score, int* __restrict__ s_cat,
                                                     int* __restrict__ n_valid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* v = reinterpret_cast<unsigned long long*>(smem);
    const int list = blockIdx.x, tid = threadIdx.x;
    int np2 = 1;
    while (np2 < Kcap) np2 <<= 1;
    for (int i = tid; i < np2; i += 1024)
        v[i] = i < Kcap ? ((unsigned long long)f2key(skey[(size_t)list * Kcap + i]) << 32) | (uint32_t)(~(uint32_t)i) : 0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long x = v[i], y = v[p];
                    const bool desc = (i & k) == 0;
                    if (desc ? x < y : x > y) { v[i] = y; v[p] = x; }
                }
            }
            __syncthreads();
        }
    int valid = 0;
    for (int i = tid; i < Kcap; i += 1024) {
        const int c = (int)(~(uint32_t)v[i]);
        const size_t src = (size_t)list * Kcap + c, dst = (size_t)list * Kcap + i;
        reinterpret_cast<float4*>(s_boxes)[dst] = reinterpret_cast<const float4*>(boxes)[src];
        const float sc = skey[src];
        s_score[dst] = sc;
        s_cat[dst] = cat[src];
        valid += sc >= 0.0f;
    }
    for (int off = 32; off > 0; off >>= 1) valid += __shfl_down(valid, off);
    __syncthreads();                                            // v[] is free now
    int* part = reinterpret_cast<int*>(smem);
    if ((tid & 63) == 0) part[tid >> 6] = valid;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += part[w];
        n_valid[list] = t;
    }
}

__global__ __launch_bounds__(256) void k_det_output(const DetPostArgs a, const int* __restrict__ keep, const int* __restrict__ n_keep,
                                                    float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                    int* __restrict__ out_labels, int* __restrict__ out_counts) {
    const int img = blockIdx.x;
    const int n_fg = min(n_keep[2 * img], a.det_per_img), n_bg = n_keep[2 * img + 1];
    for (int r = threadIdx.x; r < n_fg + n_bg; r += 256) {
        const int list = r < n_fg ? 2 * img : 2 * img + 1;
        const size_t src = (size_t)list * a.Kcap + keep[(size_t)list * a.Kcap + (r < n_fg ? r : r - n_fg)];
        const size_t dst = (size_t)img * a.out_cap + r;
        reinterpret_cast<float4*>(out_boxes)[dst] = reinterpret_cast<const float4*>(a.s_boxes)[src];
        out_scores[dst] = a.s_score[src];
        out_labels[dst] = a.s_cat[src];
    }
    if (threadIdx.x == 0) { out_counts[2 * img] = n_fg; out_counts[2 * img + 1] = n_bg; }
}

// impulse responses of the LI cell (norse leaky_integrator.py: li_feed_forward_step; v_leak = 0)
static void li_kappa(const snn_params* p, int T, Kappa* k) {
    const double a = (double)p->dt_tau_mem, cb = (double)p->neg_dt_tau_syn;
    for (int s = 0; s < SNN_MAX_STEPS; ++s) { k->last[s] = 0.f; k->sum[s] = 0.f; }
    for (int s = 0; s < T; ++s) {
        double v = 0.0, i = 0.0, acc = 0.0;
        for (int t = s; t < T; ++t) {
            const double x = (t == s) ? 1.0 : 0.0;
            if (p->li_order == 0) {          // jump-first
                const double in = i + x;
                v = v + a * (in - v);
                i = in + cb * in;
            } else {                         // voltage-first
                v = v + a * (i - v);
                i = i + cb * i + x;
            }
            acc += v;
        }
        k->last[s] = (float)v;
        k->sum[s] = (float)acc;
    }
}

#include "snn_mx.h"

// ================================================================================================
// C ABI
// ================================================================================================
// encoder fast path: Norse's default rest / reset potentials (SNN_ENC_GENERIC=1 forces the op-for-op kernels: test knob)
static bool enc_zero_rest(const NeuronP& p) {
    const char* g = getenv("SNN_ENC_GENERIC");
    return p.v_leak == 0.0f && p.v_reset == 0.0f && !(g && g[0] == '1');
}

static int g3_slots() {                       // CUs: two co-resident work-groups share a CU's matrix pipe, so the tail is
                                              // quantised per CU, not per work-group slot (fc6: MT=4 1.03 ms, MT=3 1.07 ms)
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        slots = cus;
    }
    return slots;
}

// M-tiles per wave (work-group rows = 64*MT): fewest rounds of work-groups x (tile work + per-chunk staging overhead)
template <typename F>
static int g3_pick_mt(F tiles_of) {
    int best = 0;
    double best_cost = 0;
    const char* force = getenv("SNN_BF16X3_MT");              // debug / A-B knob: 2, 3 or 4
    for (int mt = 4; mt >= 2; --mt) {
        const long long wgs = tiles_of(mt);
        if (wgs <= 0) continue;
        if (force && atoi(force) == mt) return mt;
        const double cost = (double)((wgs + g3_slots() - 1) / g3_slots()) * (mt + 0.5);
        if (best_cost == 0 || cost < best_cost * 0.97) { best = mt; best_cost = cost; }
    }
    return best;
}

#ifndef G3_NB1
#define G3_NB1 4                                    // ring slots of the 8 x 1 wave grid (14 KB each)
#endif
// waves along N of the bf16x3 tile: 2 (256 x 128 tile, default) or 1 (512 x 64 tile: half the weight copies per
// FLOP, 4-slot ring).  Both run at the same speed (conv+LIF 2.95 / 2.94 ms on one box): the copies are not what the
// kernel waits for - timing builds without them run faster because MFMAs on all-zero operands draw less power and
// the chip clocks up, not because the copies cost time.  SNN_BF16X3_WN=1|2 is an A-B / test knob.
static int g3_wn() {
    const char* f = getenv("SNN_BF16X3_WN");
    return (f && f[0] == '1') ? 1 : 2;
}

template <int MODE>
static const void* g3_kernel(int mt, int wn) {
    if (wn == 2)
        return mt == 4 ? (const void*)k_gemm_bf16x3<MODE, 3, 4, 2> : mt == 3 ? (const void*)k_gemm_bf16x3<MODE, 3, 3, 2>
                                                                              : (const void*)k_gemm_bf16x3<MODE, 3, 2, 2>;
    return mt == 4 ? (const void*)k_gemm_bf16x3<MODE, G3_NB1, 4, 1> : mt == 3 ? (const void*)k_gemm_bf16x3<MODE, G3_NB1, 3, 1>
                                                                                : (const void*)k_gemm_bf16x3<MODE, G3_NB1, 2, 1>;
}

// ------------------------------------------------------------------------------------------------
// Exchange payload of the data-parallel path: per image the max_det RoIs with the highest foreground score
// (softmax over K, best class >= 1) as rows (the 4 regression values of that class, score, label), by decreasing score,
// ties by RoI index.  One work-group per image: scores -> 64-bit keys (score bits | 0xFFFF - RoI | label) -> bitonic
// sort in LDS -> gather.  Replaces ~12 small torch launches per batch in front of the all-gather.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long det_payload_key(const float* __restrict__ l, int K, int r) {
    float m = l[0];
    for (int c = 1; c < K; ++c) m = fmaxf(m, l[c]);
    float sum = 0.0f, best = -1.0f;
    int lab = 1;
    for (int c = 0; c < K; ++c) {
        const float e = expf(l[c] - m);
        sum += e;
        if (c >= 1 && e > best) { best = e; lab = c; }
    }
    const float score = best / sum;                             // > 0
    return ((unsigned long long)__float_as_uint(score) << 32) | ((unsigned long long)(0xFFFFu - (unsigned)r) << 16) | (unsigned)lab;
}

// E = keys per thread (npad = E * blockDim.x).  E == 1: the key lives in a register, partners closer than a wave are
// exchanged by lane shuffles (45 of the 55 stages at 1024 keys need no barrier).
template <int E>
__global__ __launch_bounds__(1024) void k_det_payload(const float* __restrict__ cls, const float* __restrict__ reg, int Rn,
                                                      int K, int max_det, int npad, float* __restrict__ payload,
                                                      int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* const keys = reinterpret_cast<unsigned long long*>(smem);
    const int img = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const size_t base = (size_t)img * Rn;
    if (E == 1) {
        unsigned long long key = tid < Rn ? det_payload_key(cls + (base + tid) * K, K, tid) : 0ull;   // padding sorts last
        for (int k = 2; k <= npad; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                unsigned long long other;
                if (j < 64) {
                    other = __shfl_xor(key, j);
                } else {
                    __syncthreads();
                    keys[tid] = key;
                    __syncthreads();
                    other = keys[tid ^ j];
                }
                const bool desc = (tid & k) == 0, lower = (tid & j) == 0;
                const bool take_max = desc == lower;            // descending overall
                key = take_max ? (key > other ? key : other) : (key < other ? key : other);
            }
        __syncthreads();
        keys[tid] = key;
        __syncthreads();
    } else {
        for (int r = tid; r < npad; r += nthr) keys[r] = r < Rn ? det_payload_key(cls + (base + r) * K, K, r) : 0ull;
        __syncthreads();
        for (int k = 2; k <= npad; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < npad; i += nthr) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long a = keys[i], b = keys[ixj];
                        if ((i & k) == 0 ? a < b : a > b) { keys[i] = b; keys[ixj] = a; }
                    }
                }
                __syncthreads();
            }
    }
    const int n = min(max_det, Rn);
    for (int t = tid; t < max_det; t += nthr) {
        float* out = payload + ((size_t)img * max_det + t) * 6;
        if (t < n) {
            const unsigned long long key = keys[t];
            const int r = 0xFFFF - (int)((key >> 16) & 0xFFFFu), lab = (int)(key & 0xFFFFu);
            const float* d = reg + (base + r) * 4 * K + 4 * lab;
            out[0] = d[0]; out[1] = d[1]; out[2] = d[2]; out[3] = d[3];
            out[4] = __uint_as_float((uint32_t)(key >> 32));
            out[5] = (float)lab;
        } else {
            for (int q = 0; q < 6; ++q) out[q] = 0.0f;
        }
    }
    if (tid == 0) counts[img] = n;
}

extern "C" {

int snn_version(void) { return 1; }
const char* snn_last_error(void) { return g_err; }

size_t snn_packed_gemm_elems(int K_chunks32, int N) { return (size_t)K_chunks32 * cdiv(N, 32) * 1024; }
size_t snn_packed_conv3x3_elems(int C_out, int C_in) { return snn_packed_gemm_elems(9 * cdiv(C_in, 32), C_out); }
size_t snn_packed_linear_elems(int N, int K) { return snn_packed_gemm_elems(cdiv(K, 32), N); }
size_t snn_packed_heads_elems(int NA, int NB, int K) {
    return (size_t)cdiv(K, 32) * 32 * (size_t)(cdiv(NA + NB, 16) * 16);
}

int snn_pack_conv3x3_weight(const float* w, int C_out, int C_in, float* packed, snn_stream_t s) {
    if (!w || !packed || C_out <= 0 || C_in <= 0) return fail(-1, "snn_pack_conv3x3_weight: bad argument");
    const int Cp = cdiv(C_in, 32) * 32, Kc = 9 * (Cp / 32), Nw = cdiv(C_out, 32);
    const size_t total = (size_t)Kc * Nw * 1024;
    hipLaunchKernelGGL(k_pack_gemm_b, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)s, w, packed, (int)PACK_CONV3X3, 9 * Cp, C_out, Kc, Nw, C_in, Cp);
    SNN_CHECK_LAUNCH("k_pack_gemm_b");
    return 0;
}

int snn_pack_linear_weight(const float* w, int N, int K, float* packed, snn_stream_t s) {
    if (!w || !packed || N <= 0 || K <= 0) return fail(-1, "snn_pack_linear_weight: bad argument");
    const int Kc = cdiv(K, 32), Nw = cdiv(N, 32);
    const size_t total = (size_t)Kc * Nw * 1024;
    hipLaunchKernelGGL(k_pack_gemm_b, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)s, w, packed, (int)PACK_LINEAR, K, N, Kc, Nw, 0, 32);
    SNN_CHECK_LAUNCH("k_pack_gemm_b");
    return 0;
}

int snn_pack_heads_weight(const float* wa, int NA, const float* wb, int NB, int K, float* packed,
                          snn_stream_t s) {
    if (!wa || !wb || !packed || NA <= 0 || NB <= 0 || K <= 0)
        return fail(-1, "snn_pack_heads_weight: bad argument");
    const int Kp = cdiv(K, 32) * 32, NOp = cdiv(NA + NB, 16) * 16;
    hipLaunchKernelGGL(k_pack_heads, dim3(cdiv((long long)Kp * NOp, 256)), dim3(256), 0, (hipStream_t)s, wa, NA,
                       wb, NB, K, Kp, NOp, packed);
    SNN_CHECK_LAUNCH("k_pack_heads");
    return 0;
}

static int check_T(int T, const char* who);

// ---- bf16x3 (exact 3-way bf16 weight split on the bf16 matrix cores) ----------------------------------
size_t snn_packed_bf16x3_elems(int K_chunks32, int N) { return (size_t)3 * K_chunks32 * (cdiv(N, 32) * 32) * 32; }
size_t snn_packed_conv3x3_bf16x3_elems(int C_out, int C_in) { return snn_packed_bf16x3_elems(9 * cdiv(C_in, 32), C_out); }
size_t snn_packed_linear_bf16x3_elems(int N, int K) { return snn_packed_bf16x3_elems(cdiv(K, 32), N); }

int snn_pack_conv3x3_weight_bf16x3(const float* w, int C_out, int C_in, uint16_t* packed, snn_stream_t s) {
    if (!w || !packed || C_out <= 0 || C_in <= 0) return fail(-1, "snn_pack_conv3x3_weight_bf16x3: bad argument");
    const int Cp = cdiv(C_in, 32) * 32, Kc = 9 * (Cp / 32), Np = cdiv(C_out, 32) * 32;
    const size_t total = (size_t)Kc * Np * 32;
    hipLaunchKernelGGL(k_pack_bf16x3, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)s, w, packed, (int)PACK_CONV3X3, 9 * Cp, C_out, Kc, Np, C_in, Cp);
    SNN_CHECK_LAUNCH("k_pack_bf16x3");
    return 0;
}

int snn_pack_linear_weight_bf16x3(const float* w, int N, int K, uint16_t* packed, snn_stream_t s) {
    if (!w || !packed || N <= 0 || K <= 0) return fail(-1, "snn_pack_linear_weight_bf16x3: bad argument");
    const int Kc = cdiv(K, 32), Np = cdiv(N, 32) * 32;
    const size_t total = (size_t)Kc * Np * 32;
    hipLaunchKernelGGL(k_pack_bf16x3, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)s, w, packed, (int)PACK_LINEAR, K, N, Kc, Np, 0, 32);
    SNN_CHECK_LAUNCH("k_pack_bf16x3");
    return 0;
}

// ---- block-scaled fp6 digit planes (snn_mx.h) ---------------------------------------------------
size_t snn_packed_linear_mx_words(int N, int K) { return mx_words(cdiv(K, 128), cdiv(N, 32) * 32); }
size_t snn_packed_conv3x3_mx_words(int C_out, int C_in) { return mx_words(9 * cdiv(C_in, 128), cdiv(C_out, 32) * 32); }

int snn_pack_linear_weight_mx(const float* w, int N, int K, uint32_t* packed, snn_stream_t s) {
    if (!w || !packed || N <= 0 || K <= 0) return fail(-1, "snn_pack_linear_weight_mx: bad argument");
    const int Kc = cdiv(K, 128), Np = cdiv(N, 32) * 32;
    const size_t blocks = (size_t)Kc * Np * 4;
    hipLaunchKernelGGL(k_pack_mx, dim3((unsigned)min((size_t)8192, (blocks + 127) / 128)), dim3(128), 0, (hipStream_t)s, w, packed,
                       (int)PACK_LINEAR, K, N, Kc, Np, 0, 128);
    SNN_CHECK_LAUNCH("k_pack_mx");
    return 0;
}

int snn_pack_conv3x3_weight_mx(const float* w, int C_out, int C_in, uint32_t* packed, snn_stream_t s) {
    if (!w || !packed || C_out <= 0 || C_in <= 0) return fail(-1, "snn_pack_conv3x3_weight_mx: bad argument");
    const int Cp = cdiv(C_in, 128) * 128, Kc = 9 * Cp / 128, Np = cdiv(C_out, 32) * 32;
    const size_t blocks = (size_t)Kc * Np * 4;
    hipLaunchKernelGGL(k_pack_mx, dim3((unsigned)min((size_t)8192, (blocks + 127) / 128)), dim3(128), 0, (hipStream_t)s, w, packed,
                       (int)PACK_CONV3X3, 9 * Cp, C_out, Kc, Np, C_in, Cp);
    SNN_CHECK_LAUNCH("k_pack_mx");
    return 0;
}

// T-in-tile LIF fusion needs T <= 64 (one lane per time step keeps the spike words) and wastes rows % T rows per tile
static bool g3_tile_ok(int T, int rows) { return T >= 1 && T <= 64 && T <= rows && T * (rows / T) * 10 >= rows * 9; }

static int launch_gemm3(int mode, int mt, int wn, const Gemm3Args& a, hipStream_t s) {
    // LIF_REG owns its CU (256 registers per wave); the others run two work-groups per CU
    const void* kern;
    int lds = G3_LDS(wn == 1 ? G3_NB1 : 3, wn), tiles = cdiv(a.M, G3_BM(wn, mt));
    const int tile_lds = wn == 2 ? G3_TILE_BYTES(2) : G3_TILE_BYTES(1);       // the LIF_TILE epilogue reuses the ring
    switch (mode) {
    case G3_FC: kern = g3_kernel<G3_FC>(mt, wn); break;
    case G3_CONV: kern = g3_kernel<G3_CONV>(mt, wn); break;
    case G3_CONV_LIF_REG: kern = (const void*)k_gemm_bf16x3<G3_CONV_LIF_REG, 3, 4, 2>; lds = G3_LDS(3, 2) + G3_STATE_BYTES; break;
    case G3_CONV_LIF_TILE: kern = g3_kernel<G3_CONV_LIF_TILE>(mt, wn); tiles = cdiv(a.M, a.pb); lds = max(lds, tile_lds); break;
    default: kern = g3_kernel<G3_FC_LIF_TILE>(mt, wn); tiles = cdiv(a.M, a.pb); lds = max(lds, tile_lds); break;
    }
    static_assert(2 * G3_TILE_BYTES(1) <= 160 * 1024 && 2 * G3_LDS(3, 2) <= 160 * 1024, "two work-groups per CU");
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(-3, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    if (getenv("SNN_DEBUG_OCC")) {                             // debug: co-resident work-groups per CU
        int v = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kern, 512, lds);
        fprintf(stderr, "k_gemm_bf16x3 mode %d mt %d wn %d: lds %d B, %d work-groups per CU, grid %d\n", mode, mt, wn, lds, v, tiles * a.n_blocks);
    }
    void* kargs[] = {(void*)&a};
    e = hipLaunchKernel(kern, dim3(tiles * a.n_blocks), dim3(512), kargs, lds, s);
    if (e != hipSuccess) return fail(-3, "k_gemm_bf16x3 launch failed: %s", hipGetErrorString(e));
    SNN_CHECK_LAUNCH("k_gemm_bf16x3");
    return 0;
}

int snn_spike_gemm_bf16x3(const uint32_t* a_rows, int M, int K, int N, const uint16_t* w_packed, float* cur, int ldo,
                          snn_stream_t s) {
    if (!a_rows || !w_packed || !cur || M <= 0 || K <= 0 || N <= 0 || ldo < N)
        return fail(-1, "snn_spike_gemm_bf16x3: bad argument");
    Gemm3Args a;
    memset(&a, 0, sizeof(a));
    if ((long long)M * cdiv(K, 32) * 4 > 0xffffffffLL) return fail(-1, "snn_spike_gemm_bf16x3: spike rows over 4 GB");
    a.A = a_rows; a.wpk = w_packed; a.out = cur; a.M = M; a.Kc = cdiv(K, 32); a.Np = cdiv(N, 32) * 32; a.ldo = ldo;
    a.plane_elems = (unsigned long long)a.Kc * a.Np * 32;
    const int wn = g3_wn();
    a.n_blocks = cdiv(a.Np, G3_BN(wn));
    const int mt = g3_pick_mt([&](int m) { return (long long)cdiv(M, G3_BM(wn, m)) * a.n_blocks; });
    return launch_gemm3(G3_FC, mt, wn, a, (hipStream_t)s);
}

int snn_spike_gemm_lif_bf16x3(const uint32_t* a_planes, int T, int R, int K, int N, const snn_params* p,
                              const uint16_t* w_packed, uint32_t* spk, size_t spk_stride, snn_stream_t s) {
    if (!a_planes || !w_packed || !spk || !p || R <= 0 || K <= 0 || N <= 0)
        return fail(-1, "snn_spike_gemm_lif_bf16x3: bad argument");
    if (check_T(T, "snn_spike_gemm_lif_bf16x3")) return -1;
    if ((long long)T * R * cdiv(K, 32) * 4 > 0xffffffffLL) return fail(-1, "snn_spike_gemm_lif_bf16x3: input planes over 4 GB");
    Gemm3Args a;
    memset(&a, 0, sizeof(a));
    a.A = a_planes; a.wpk = w_packed; a.M = R; a.Kc = cdiv(K, 32); a.Np = cdiv(N, 32) * 32;
    a.plane_elems = (unsigned long long)a.Kc * a.Np * 32;
    const int wn = g3_wn();
    a.n_blocks = cdiv(a.Np, G3_BN(wn));
    a.T = T; a.spk = spk; a.spk_stride = spk_stride; a.p = make_p(p, p->v_th_lif);
    const int mt = g3_pick_mt([&](int m) { return g3_tile_ok(T, G3_BM(wn, m)) ? (long long)cdiv(R, G3_BM(wn, m) / T) * a.n_blocks : 0ll; });
    if (!mt) return fail(-4, "snn_spike_gemm_lif_bf16x3: T=%d does not fit a row tile (use snn_spike_gemm_bf16x3 + snn_lif_scan)", T);
    a.pb = G3_BM(wn, mt) / T;
    return launch_gemm3(G3_FC_LIF_TILE, mt, wn, a, (hipStream_t)s);
}

// ---- spike GEMMs on the block-scaled fp4 x fp6 path (snn_mx.h) ------------------------------------
static bool mx_tile_ok(int T) { return g3_tile_ok(T, MX_BM); }

// rows per wave: 4 M-tiles (8 waves, 128 registers; default) or 8 (4 waves, 256 registers: half the LDS fragment reads
// per MFMA - measured equal in the loop, slower in the LIF epilogue); SNN_MX_MW=4|8 overrides (debug / A-B switch)
static int mx_mw() {
    const char* e = getenv("SNN_MX_MW");
    return e && e[0] == '8' ? 8 : (e && e[0] == '4' ? 4 : MX_MW_DEFAULT);
}

#define MX_KERNEL_OF(MW, mode)                                                                        \
    ((mode) == G3_FC ? (const void*)k_gemm_mx<G3_FC, MW>                                              \
     : (mode) == G3_CONV ? (const void*)k_gemm_mx<G3_CONV, MW>                                        \
     : (mode) == G3_CONV_LIF_TILE ? (const void*)k_gemm_mx<G3_CONV_LIF_TILE, MW>                      \
                                  : (const void*)k_gemm_mx<G3_FC_LIF_TILE, MW>)

static int launch_gemm_mx(int mode, MxArgs& a, hipStream_t s) {
    const int mw = mx_mw();
    const void* kern = mw == 8 ? MX_KERNEL_OF(8, mode) : MX_KERNEL_OF(4, mode);
    const bool tile = mode == G3_CONV_LIF_TILE || mode == G3_FC_LIF_TILE;
    const int lds = tile ? max((int)MX_LDS, (int)G3_TILE_BYTES(1)) : MX_LDS;
    const int tiles = tile ? cdiv(a.g.M, a.g.pb) : cdiv(a.g.M, MX_BM);
    a.g.n_blocks = cdiv(a.g.Np, MX_BN);
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(-3, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    if (getenv("SNN_DEBUG_OCC")) {
        int v = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kern, 2048 / mw, lds);
        fprintf(stderr, "k_gemm_mx mode %d mw %d: lds %d B, %d work-groups per CU, grid %d\n", mode, mw, lds, v, tiles * a.g.n_blocks);
    }
    void* kargs[] = {(void*)&a};
    e = hipLaunchKernel(kern, dim3(tiles * a.g.n_blocks), dim3(2048 / mw), kargs, lds, s);
    if (e != hipSuccess) return fail(-3, "k_gemm_mx launch failed: %s", hipGetErrorString(e));
    SNN_CHECK_LAUNCH("k_gemm_mx");
    return 0;
}

int snn_spike_gemm_mx(const uint32_t* a_rows, int M, int K, int N, const uint32_t* w_packed, float* cur, int ldo,
                      snn_stream_t s) {
    if (!a_rows || !w_packed || !cur || M <= 0 || K <= 0 || N <= 0 || ldo < N) return fail(-1, "snn_spike_gemm_mx: bad argument");
    if (K % 128) return fail(-4, "snn_spike_gemm_mx: K=%d is not a multiple of 128 (use the bf16x3 kernels)", K);
    if ((long long)M * (K / 32) * 4 > 0xffffffffLL) return fail(-1, "snn_spike_gemm_mx: spike rows over 4 GB");
    MxArgs a;
    memset(&a, 0, sizeof(a));
    a.g.A = a_rows; a.g.out = cur; a.g.M = M; a.g.Np = cdiv(N, 32) * 32; a.g.ldo = ldo;
    a.wq = w_packed; a.Kc = K / 128;
    return launch_gemm_mx(G3_FC, a, (hipStream_t)s);
}

int snn_spike_gemm_lif_mx(const uint32_t* a_planes, int T, int R, int K, int N, const snn_params* p,
                          const uint32_t* w_packed, uint32_t* spk, size_t spk_stride, snn_stream_t s) {
    if (!a_planes || !w_packed || !spk || !p || R <= 0 || K <= 0 || N <= 0) return fail(-1, "snn_spike_gemm_lif_mx: bad argument");
    if (check_T(T, "snn_spike_gemm_lif_mx")) return -1;
    if (K % 128) return fail(-4, "snn_spike_gemm_lif_mx: K=%d is not a multiple of 128", K);
    if (!mx_tile_ok(T)) return fail(-4, "snn_spike_gemm_lif_mx: T=%d does not fit a row tile", T);
    if ((long long)T * R * (K / 32) * 4 > 0xffffffffLL) return fail(-1, "snn_spike_gemm_lif_mx: input planes over 4 GB");
    MxArgs a;
    memset(&a, 0, sizeof(a));
    a.g.A = a_planes; a.g.M = R; a.g.Np = cdiv(N, 32) * 32;
    a.g.T = T; a.g.spk = spk; a.g.spk_stride = spk_stride; a.g.p = make_p(p, p->v_th_lif); a.g.pb = MX_BM / T;
    a.wq = w_packed; a.Kc = K / 128;
    return launch_gemm_mx(G3_FC_LIF_TILE, a, (hipStream_t)s);
}

static int conv_mx_common(const char* who, const uint32_t* enc, size_t enc_stride, const snn_rpn_level* lv, int n_levels,
                          int C_in, int C_out, int T, const uint32_t* w_packed, MxArgs& a, long long* P_out) {
    if (!enc || !lv || !w_packed || n_levels <= 0 || n_levels > SNN_MAX_LEVELS || C_in <= 0 || C_out <= 0)
        return fail(-1, "%s: bad argument", who);
    if (C_in % 128) return fail(-4, "%s: C_in=%d is not a multiple of 128 (use the bf16x3 kernels)", who, C_in);
    if (check_T(T, who)) return -1;
    memset(&a, 0, sizeof(a));
    long long P = 0, Pp = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (lv[l].N <= 0 || lv[l].H <= 0 || lv[l].W <= 0) return fail(-1, "%s: bad level %d", who, l);
        a.g.lv[l].pos_base = (int)P; a.g.lv[l].N = lv[l].N; a.g.lv[l].H = lv[l].H; a.g.lv[l].W = lv[l].W;
        a.g.lv[l].tile_begin = (int)Pp;                // first row of the level in the padded (zero-halo) encoder planes
        P += (long long)lv[l].N * lv[l].H * lv[l].W;
        Pp += (long long)lv[l].N * (lv[l].H + 2) * (lv[l].W + 2);
    }
    if ((long long)T * P > 0x7fffffffLL) return fail(-1, "%s: T*P too large", who);
    if (enc_stride < (size_t)Pp * (C_in / 32)) return fail(-1, "%s: enc_stride %zu < %lld words (planes with a one-position zero halo)", who, enc_stride, Pp * (C_in / 32));
    if (((long long)(T - 1) * (long long)enc_stride + Pp * (C_in / 32)) * 4 > 0xffffffffLL) return fail(-1, "%s: encoder planes over 4 GB", who);
    a.g.A = enc; a.g.enc_stride = enc_stride; a.g.Cw = C_in / 32; a.g.Np = cdiv(C_out, 32) * 32;
    a.g.P_total = (int)P; a.g.n_levels = n_levels;
    a.wq = w_packed; a.Kc = 9 * C_in / 128;
    *P_out = P;
    return 0;
}

int snn_conv3x3_lif_mx(const uint32_t* enc, size_t enc_stride, const snn_rpn_level* lv, int n_levels, int C_in, int C_out,
                       int T, const snn_params* p, const uint32_t* w_packed, uint32_t* spk, size_t spk_stride,
                       snn_stream_t s) {
    if (!spk || !p) return fail(-1, "snn_conv3x3_lif_mx: bad argument");
    MxArgs a;
    long long P;
    int rc = conv_mx_common("snn_conv3x3_lif_mx", enc, enc_stride, lv, n_levels, C_in, C_out, T, w_packed, a, &P);
    if (rc) return rc;
    if (!mx_tile_ok(T)) return fail(-4, "snn_conv3x3_lif_mx: T=%d does not fit a row tile", T);
    a.g.M = (int)P; a.g.T = T; a.g.spk = spk; a.g.spk_stride = spk_stride; a.g.p = make_p(p, p->v_th_lif); a.g.pb = MX_BM / T;
    return launch_gemm_mx(G3_CONV_LIF_TILE, a, (hipStream_t)s);
}

int snn_spike_conv3x3_mx(const uint32_t* enc, size_t enc_stride, const snn_rpn_level* lv, int n_levels, int C_in, int C_out,
                         int T, const uint32_t* w_packed, float* cur, int ldo, snn_stream_t s) {
    if (!cur || ldo < C_out) return fail(-1, "snn_spike_conv3x3_mx: bad argument");
    MxArgs a;
    long long P;
    int rc = conv_mx_common("snn_spike_conv3x3_mx", enc, enc_stride, lv, n_levels, C_in, C_out, T, w_packed, a, &P);
    if (rc) return rc;
    a.g.M = (int)(T * P); a.g.out = cur; a.g.ldo = ldo;
    return launch_gemm_mx(G3_CONV, a, (hipStream_t)s);
}

static int conv3_common(const char* who, const uint32_t* enc, size_t enc_stride, const snn_rpn_level* lv, int n_levels,
                        int C_in, int C_out, int T, const uint16_t* w_packed, Gemm3Args& a, long long* P_out) {
    if (!enc || !lv || !w_packed || n_levels <= 0 || n_levels > SNN_MAX_LEVELS || C_in <= 0 || C_out <= 0)
        return fail(-1, "%s: bad argument", who);
    if (check_T(T, who)) return -1;
    memset(&a, 0, sizeof(a));
    long long P = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (lv[l].N <= 0 || lv[l].H <= 0 || lv[l].W <= 0) return fail(-1, "%s: bad level %d", who, l);
        a.lv[l].pos_base = (int)P; a.lv[l].N = lv[l].N; a.lv[l].H = lv[l].H; a.lv[l].W = lv[l].W;
        P += (long long)lv[l].N * lv[l].H * lv[l].W;
    }
    if ((long long)T * P > 0x7fffffffLL) return fail(-1, "%s: T*P too large", who);
    // the kernels address a spike word as 64-bit scalar base + 32-bit lane byte offset
    if (((long long)(T - 1) * (long long)enc_stride + P * cdiv(C_in, 32)) * 4 > 0xffffffffLL)
        return fail(-1, "%s: encoder planes over 4 GB", who);
    a.A = enc; a.wpk = w_packed; a.enc_stride = enc_stride;
    a.Cw = cdiv(C_in, 32); a.Kc = 9 * a.Cw; a.Np = cdiv(C_out, 32) * 32;
    a.plane_elems = (unsigned long long)a.Kc * a.Np * 32;
    a.P_total = (int)P; a.n_levels = n_levels;
    *P_out = P;
    return 0;
}

int snn_conv3x3_lif_bf16x3(const uint32_t* enc, size_t enc_stride, const snn_rpn_level* lv, int n_levels, int C_in,
                           int C_out, int T, const snn_params* p, const uint16_t* w_packed, uint32_t* spk,
                           size_t spk_stride, snn_stream_t s) {
    if (!spk || !p) return fail(-1, "snn_conv3x3_lif_bf16x3: bad argument");
    Gemm3Args a;
    long long P;
    int rc = conv3_common("snn_conv3x3_lif_bf16x3", enc, enc_stride, lv, n_levels, C_in, C_out, T, w_packed, a, &P);
    if (rc) return rc;
    a.M = (int)P; a.T = T; a.spk = spk; a.spk_stride = spk_stride; a.p = make_p(p, p->v_th_lif);
    // debug / A-B knob: SNN_BF16X3_LIF=reg forces the register-resident variant (the fallback for T > 64)
    const char* force = getenv("SNN_BF16X3_LIF");
    const int wn = g3_wn();
    int mt = 0;
    if (!(force && !strcmp(force, "reg"))) {
        a.n_blocks = cdiv(a.Np, G3_BN(wn));
        mt = g3_pick_mt([&](int m) { return g3_tile_ok(T, G3_BM(wn, m)) ? (long long)cdiv(P, G3_BM(wn, m) / T) * a.n_blocks : 0ll; });
    }
    if (!mt) {
        a.n_blocks = cdiv(a.Np, G3_BN(2));
        return launch_gemm3(G3_CONV_LIF_REG, 4, 2, a, (hipStream_t)s);
    }
    a.pb = G3_BM(wn, mt) / T;
    return launch_gemm3(G3_CONV_LIF_TILE, mt, wn, a, (hipStream_t)s);
}

int snn_spike_conv3x3_bf16x3(const uint32_t* enc, size_t enc_stride, const snn_rpn_level* lv, int n_levels, int C_in,
                             int C_out, int T, const uint16_t* w_packed, float* cur, int ldo, snn_stream_t s) {
    if (!enc || !lv || !w_packed || !cur || n_levels <= 0 || n_levels > SNN_MAX_LEVELS || C_in <= 0 || C_out <= 0 ||
        ldo < C_out)
        return fail(-1, "snn_spike_conv3x3_bf16x3: bad argument");
    if (check_T(T, "snn_spike_conv3x3_bf16x3")) return -1;
    Gemm3Args a;
    memset(&a, 0, sizeof(a));
    long long P = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (lv[l].N <= 0 || lv[l].H <= 0 || lv[l].W <= 0) return fail(-1, "snn_spike_conv3x3_bf16x3: bad level %d", l);
        a.lv[l].pos_base = (int)P; a.lv[l].N = lv[l].N; a.lv[l].H = lv[l].H; a.lv[l].W = lv[l].W;
        P += (long long)lv[l].N * lv[l].H * lv[l].W;
    }
    if ((long long)T * P > 0x7fffffffLL) return fail(-1, "snn_spike_conv3x3_bf16x3: T*P too large");
    if (((long long)(T - 1) * (long long)enc_stride + P * cdiv(C_in, 32)) * 4 > 0xffffffffLL)
        return fail(-1, "snn_spike_conv3x3_bf16x3: encoder planes over 4 GB");
    a.A = enc; a.wpk = w_packed; a.out = cur; a.enc_stride = enc_stride;
    a.Cw = cdiv(C_in, 32); a.Kc = 9 * a.Cw; a.Np = cdiv(C_out, 32) * 32; a.ldo = ldo;
    a.plane_elems = (unsigned long long)a.Kc * a.Np * 32;
    a.P_total = (int)P; a.n_levels = n_levels; a.M = (int)(T * P);
    const int wn = g3_wn();
    a.n_blocks = cdiv(a.Np, G3_BN(wn));
    const int mt = g3_pick_mt([&](int m) { return (long long)cdiv(a.M, G3_BM(wn, m)) * a.n_blocks; });
    return launch_gemm3(G3_CONV, mt, wn, a, (hipStream_t)s);
}

static int check_T(int T, const char* who) {
    if (T < 1 || T > SNN_MAX_STEPS) return fail(-1, "%s: num_steps %d outside [1, %d]", who, T, SNN_MAX_STEPS);
    return 0;
}

int snn_encode_nchw(const float* feat, int N, int C, int H, int W, int T, const snn_params* p,
                    uint32_t* planes, size_t plane_stride, snn_stream_t s) {
    if (!feat || !planes || !p || N <= 0 || C <= 0 || H <= 0 || W <= 0)
        return fail(-1, "snn_encode_nchw: bad argument");
    if (check_T(T, "snn_encode_nchw")) return -1;
    const int Cw = cdiv(C, 32), HW = H * W;
    const NeuronP np = make_p(p, p->v_th_enc);
    if (enc_zero_rest(np))
        hipLaunchKernelGGL(k_encode_nchw<true>, dim3(cdiv(HW, ENC_PB), cdiv(Cw, 8), N), dim3(256), 0, (hipStream_t)s, feat, C, HW, Cw, T, np, planes, plane_stride);
    else
        hipLaunchKernelGGL(k_encode_nchw<false>, dim3(cdiv(HW, ENC_PB), cdiv(Cw, 8), N), dim3(256), 0, (hipStream_t)s, feat, C, HW, Cw, T, np, planes, plane_stride);
    SNN_CHECK_LAUNCH("k_encode_nchw");
    return 0;
}

int snn_encode_rows(const float* x, int R, int D, int T, const snn_params* p, uint32_t* planes,
                    size_t plane_stride, snn_stream_t s) {
    if (!x || !planes || !p || R <= 0 || D <= 0) return fail(-1, "snn_encode_rows: bad argument");
    if (check_T(T, "snn_encode_rows")) return -1;
    const int Dw = cdiv(D, 32);
    const size_t total = (size_t)R * Dw * 32;
    const NeuronP np = make_p(p, p->v_th_enc);
    const char* force = getenv("SNN_ENC_ROWS");                // debug / A-B knob: "ballot" forces the element-per-lane kernel
    if (D % 32 == 0 && ((uintptr_t)x & 15) == 0 && !(force && !strcmp(force, "ballot"))) {
        const size_t n_words = (size_t)R * Dw;
        const dim3 gw((unsigned)((n_words + 255) / 256));
        if (enc_zero_rest(np)) hipLaunchKernelGGL(k_encode_rows_w<true>, gw, dim3(256), 0, (hipStream_t)s, x, n_words, T, np, planes, plane_stride);
        else hipLaunchKernelGGL(k_encode_rows_w<false>, gw, dim3(256), 0, (hipStream_t)s, x, n_words, T, np, planes, plane_stride);
        SNN_CHECK_LAUNCH("k_encode_rows_w");
        return 0;
    }
    const dim3 grid((unsigned)((total + 256 * ENC_U - 1) / (256 * ENC_U)));
    if (enc_zero_rest(np)) hipLaunchKernelGGL(k_encode_rows<true>, grid, dim3(256), 0, (hipStream_t)s, x, R, D, Dw, T, np, planes, plane_stride);
    else hipLaunchKernelGGL(k_encode_rows<false>, grid, dim3(256), 0, (hipStream_t)s, x, R, D, Dw, T, np, planes, plane_stride);
    SNN_CHECK_LAUNCH("k_encode_rows");
    return 0;
}


size_t snn_nms_workspace_bytes(int n) { return (size_t)n * cdiv(n, 64) * 8; }

int snn_nms_sorted(const float* boxes_sorted, const int* category_sorted, int n, float iou_threshold, int max_keep,
                   int* keep_out, int* n_keep_out, void* ws, size_t ws_bytes, snn_stream_t s) {
    if (!boxes_sorted || !keep_out || !n_keep_out || !ws || n <= 0 || max_keep <= 0)
        return fail(-1, "snn_nms_sorted: bad argument");
    const int words = cdiv(n, 64);
    if (words > 256) return fail(-1, "snn_nms_sorted: n=%d too large (max 16384)", n);
    if (ws_bytes < snn_nms_workspace_bytes(n)) return fail(-2, "snn_nms_sorted: workspace too small");
    unsigned long long* mask = (unsigned long long*)ws;
    NmsBatch nb;
    memset(&nb, 0, sizeof(nb));
    hipLaunchKernelGGL(k_nms_mask, dim3(words, words), dim3(64), 0, (hipStream_t)s, boxes_sorted, category_sorted, n,
                       iou_threshold, mask, words, nb);
    SNN_CHECK_LAUNCH("k_nms_mask");
    const int dbl = words <= 156 ? 1 : 0;                   // two chunk buffers fit the LDS up to n = 9984
    const size_t lds = (size_t)(1 + dbl) * 64 * words * 8;
    hipError_t e = hipFuncSetAttribute((const void*)k_nms_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(-3, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(256), lds, (hipStream_t)s, mask, n, words, max_keep, dbl, keep_out, n_keep_out, nb);
    SNN_CHECK_LAUNCH("k_nms_scan");
    return 0;
}

int snn_roi_align_encode(const snn_roi_level* levels_host, int n_levels, int C, const float* rois, const int* roi_batch,
                         const int* roi_level, int R, int T, const snn_params* p, uint32_t* planes,
                         size_t plane_stride, float* pooled_dbg, snn_stream_t s) {
    if (!levels_host || n_levels <= 0 || n_levels > 4 || C <= 0 || !rois || !roi_batch || !roi_level || R <= 0 || !p ||
        !planes)
        return fail(-1, "snn_roi_align_encode: bad argument");
    if (check_T(T, "snn_roi_align_encode")) return -1;
    RoiArgs a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l < n_levels; ++l) {
        if (!levels_host[l].feat || levels_host[l].H <= 0 || levels_host[l].W <= 0)
            return fail(-1, "snn_roi_align_encode: bad level %d", l);
        a.lv[l].feat = levels_host[l].feat; a.lv[l].H = levels_host[l].H; a.lv[l].W = levels_host[l].W;
        a.lv[l].scale = levels_host[l].spatial_scale;
    }
    a.rois = rois; a.roi_batch = roi_batch; a.roi_level = roi_level; a.pooled = pooled_dbg; a.planes = planes;
    a.plane_stride = plane_stride; a.R = R; a.C = C; a.T = T; a.Dw = cdiv(C * 49, 32);
    a.p = make_p(p, p->v_th_enc);
    hipLaunchKernelGGL(k_roi_align_encode, dim3(cdiv(a.Dw * 32, 256), R), dim3(256), 0, (hipStream_t)s, a);
    SNN_CHECK_LAUNCH("k_roi_align_encode");
    return 0;
}

// shared by snn_conv3x3_lif (one level) and snn_rpn_head_forward (all levels in one launch)
static int launch_conv(const snn_rpn_level* lv, int n_levels, int C_in, int C_out, int T, const snn_params* p,
                       const uint32_t* enc, size_t enc_stride, const float* wpk, uint32_t* spk,
                       size_t spk_stride, unsigned long long* counts, int max_n, float* dbg_cur,
                       hipStream_t s) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.enc = enc; a.spk = spk; a.wpk = wpk; a.counts = counts; a.dbg_cur = dbg_cur;
    a.enc_stride = enc_stride; a.spk_stride = spk_stride;
    a.Cw = cdiv(C_in, 32); a.Nw = cdiv(C_out, 32); a.T = T; a.n_levels = n_levels; a.max_n = max_n;
    a.p = make_p(p, p->v_th_lif);
    int tiles = 0, pos = 0;
    for (int l = 0; l < n_levels; ++l) {
        ConvLevelDev& d = a.lv[l];
        d.pos_base = pos; d.N = lv[l].N; d.H = lv[l].H; d.W = lv[l].W;
        d.tiles_x = cdiv(lv[l].W, CONV_PW);
        d.tiles_per_img = d.tiles_x * cdiv(lv[l].H, CONV_PH);
        d.tile_begin = tiles;
        tiles += d.tiles_per_img * lv[l].N;
        pos += lv[l].N * lv[l].H * lv[l].W;
    }
    if (a.Cw > CONV_MAX_CW) return fail(-1, "conv3x3_lif: C_in=%d > %d not supported", C_in, CONV_MAX_CW * 32);
    const size_t lds = (size_t)CONV_HALO * (a.Cw * 32 + CONV_APAD) * 4;
    auto kern = dbg_cur ? k_conv3x3_lif<true> : k_conv3x3_lif<false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(-3, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(tiles, cdiv(a.Nw, CONV_BNT)), dim3(512), lds, s, a);
    SNN_CHECK_LAUNCH("k_conv3x3_lif");
    return 0;
}

int snn_conv3x3_lif(const uint32_t* enc, size_t enc_stride, int N, int C_in, int C_out, int H, int W, int T,
                    const snn_params* p, const float* w_packed, uint32_t* spk, size_t spk_stride,
                    unsigned long long* counts, float* dbg_cur, snn_stream_t s) {
    if (!enc || !spk || !w_packed || !p || N <= 0 || C_in <= 0 || C_out <= 0 || H <= 0 || W <= 0)
        return fail(-1, "snn_conv3x3_lif: bad argument");
    if (check_T(T, "snn_conv3x3_lif")) return -1;
    snn_rpn_level lv; lv.feat = nullptr; lv.N = N; lv.H = H; lv.W = W; lv.reserved = 0;
    return launch_conv(&lv, 1, C_in, C_out, T, p, enc, enc_stride, w_packed, spk, spk_stride, counts, N, dbg_cur,
                       (hipStream_t)s);
}

int snn_spike_gemm(const uint32_t* a_rows, int M, int K, int N, const float* w_packed, float* cur, int ldo,
                   snn_stream_t s) {
    if (!a_rows || !w_packed || !cur || M <= 0 || K <= 0 || N <= 0 || ldo < N)
        return fail(-1, "snn_spike_gemm: bad argument");
    GemmArgs a;
    a.A = a_rows; a.wpk = w_packed; a.out = cur; a.M = M; a.Kw = cdiv(K, 32); a.Nw = cdiv(N, 32); a.ldo = ldo; a.pad = 0;
    // 128 rows x 256 columns per work-group: 8 waves, each 128 x 32
    constexpr int MT = 4;
    a.n_blocks = cdiv(a.Nw, 8);
    auto kern = k_spike_gemm<MT>;
    hipLaunchKernelGGL(kern, dim3(cdiv(M, MT * 32) * a.n_blocks), dim3(512), 2 * 128 * GEMM_AST * 4, (hipStream_t)s, a);
    SNN_CHECK_LAUNCH("k_spike_gemm");
    return 0;
}

int snn_lif_scan(const float* cur, int T, int R, int N, int ldc, const snn_params* p, uint32_t* spk,
                 size_t spk_stride, uint32_t* row_counts, snn_stream_t s) {
    if (!cur || !spk || !p || R <= 0 || N <= 0 || ldc < N) return fail(-1, "snn_lif_scan: bad argument");
    if (check_T(T, "snn_lif_scan")) return -1;
    const int Nw = cdiv(N, 32);
    const size_t total = (size_t)R * Nw * 32;
    hipLaunchKernelGGL(k_lif_scan, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, cur, T, R, N,
                       Nw, ldc, make_p(p, p->v_th_lif), spk, spk_stride, row_counts);
    SNN_CHECK_LAUNCH("k_lif_scan");
    return 0;
}

int snn_li_heads(const uint32_t* spk, size_t spk_stride, int T, int M, int K, const float* w_heads_packed,
                 int NA, int NB, const snn_params* p, float* out_a, float* out_b, float* sum_a, float* sum_b,
                 snn_stream_t s) {
    if (!spk || !w_heads_packed || !p || !out_a || !out_b || M <= 0 || K <= 0 || NA <= 0 || NB <= 0)
        return fail(-1, "snn_li_heads: bad argument");
    if ((sum_a == nullptr) != (sum_b == nullptr)) return fail(-1, "snn_li_heads: sum_a and sum_b go together");
    if (check_T(T, "snn_li_heads")) return -1;
    Kappa kap;
    li_kappa(p, T, &kap);
    const int Kw = cdiv(K, 32), NOp = cdiv(NA + NB, 16) * 16;
    if (NOp > 256) return fail(-1, "snn_li_heads: %d outputs per row not supported", NA + NB);
    const char* force = getenv("SNN_LI_HEADS");               // debug / A-B knob: "valu" forces the fp32 VALU kernel
    // matrix-core kernel where all of W (as three bf16 planes) stays resident in LDS; the streamed form is latency
    // bound on small row counts (detector heads: 164 us against 88 us for the VALU kernel) and only runs when forced
    const bool fits = (size_t)Kw * 3 * NOp * 64 <= 96 * 1024;
    // W too large for LDS: one work-group per 16 rows, the reduction split over its 4 waves ("ksplit" forces it)
    if (NOp <= 64 && T <= LIH_KS_TM(NOp / 16) && Kw >= 4 && (force ? !strcmp(force, "ksplit") : !fits)) {
        LiHeadsArgs a;
        memset(&a, 0, sizeof(a));
        a.spk = spk; a.spk_stride = spk_stride; a.wT = w_heads_packed; a.out_a = out_a; a.out_b = out_b;
        a.sum_a = sum_a; a.sum_b = sum_b; a.T = T; a.M = M; a.Kw = Kw; a.NOp = NOp; a.NA = NA; a.NB = NB; a.kap = kap;
        const int nt = NOp / 16;
        const size_t lds = G3_LUT_BYTES + (size_t)4 * 2 * nt * 64 * 16;
        const void* kern = nt == 1 ? (const void*)k_li_heads_ksplit<1> : nt == 2 ? (const void*)k_li_heads_ksplit<2>
                         : nt == 3 ? (const void*)k_li_heads_ksplit<3> : (const void*)k_li_heads_ksplit<4>;
        void* kargs[] = {(void*)&a};
        hipError_t e = hipLaunchKernel(kern, dim3(cdiv(M, 16)), dim3(256), kargs, lds, (hipStream_t)s);
        if (e != hipSuccess) return fail(-3, "k_li_heads_ksplit launch failed: %s", hipGetErrorString(e));
        SNN_CHECK_LAUNCH("k_li_heads_ksplit");
        return 0;
    }
    if (NOp <= 64 && (force ? !strcmp(force, "mfma") : fits)) {
        LiHeadsArgs a;
        memset(&a, 0, sizeof(a));
        a.spk = spk; a.spk_stride = spk_stride; a.wT = w_heads_packed; a.out_a = out_a; a.out_b = out_b;
        a.sum_a = sum_a; a.sum_b = sum_b; a.T = T; a.M = M; a.Kw = Kw; a.NOp = NOp; a.NA = NA; a.NB = NB; a.kap = kap;
        a.n_groups = cdiv(M, 64);
        const size_t all = (size_t)Kw * 3 * NOp * 64;
        a.resident = all <= 96 * 1024;
        const size_t lds = G3_LUT_BYTES + (a.resident ? all : (size_t)2 * 3 * NOp * 64);
        const int nt = NOp / 16;
        const void* kern = nt == 1 ? (const void*)k_li_heads_mfma<1> : nt == 2 ? (const void*)k_li_heads_mfma<2>
                         : nt == 3 ? (const void*)k_li_heads_mfma<3> : (const void*)k_li_heads_mfma<4>;
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(-3, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        // resident weights: persistent work-groups (staging once per work-group), exactly as many as are co-resident
        // (registers allow 3 per SIMD: a grid of 4 per CU ran a third of them as a second, mostly empty round);
        // streamed: one row group each
        static int per_cu_cache[5] = {0, 0, 0, 0, 0};           // by column tiles; the LDS size of the resident form follows from them and Kw
        static size_t per_cu_lds[5] = {0, 0, 0, 0, 0};
        if (per_cu_cache[nt] == 0 || per_cu_lds[nt] != lds) {
            int v = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kern, 256, lds) != hipSuccess || v <= 0) v = 3;
            per_cu_cache[nt] = v; per_cu_lds[nt] = lds;
        }
        const int per_cu = per_cu_cache[nt];
        const int grid = a.resident ? min(a.n_groups, per_cu * g3_slots()) : a.n_groups;
        void* kargs[] = {(void*)&a};
        e = hipLaunchKernel(kern, dim3(grid), dim3(256), kargs, lds, (hipStream_t)s);
        if (e != hipSuccess) return fail(-3, "k_li_heads_mfma launch failed: %s", hipGetErrorString(e));
        SNN_CHECK_LAUNCH("k_li_heads_mfma");
        return 0;
    }
    // rows per block: as many as 256 threads can own (row, 4-output group) pairs for
    const int jg = NOp / 4;
    const int rb = (256 / jg >= 64) ? 64 : (256 / jg >= 32) ? 32 : (256 / jg >= 16) ? 16 : (256 / jg >= 8) ? 8 : 4;
    const size_t lds = ((size_t)(sum_a ? 2 : 1) * rb * (HEADS_KS + 4) + (size_t)HEADS_KS * NOp) * 4;
    if (lds > 160 * 1024) return fail(-1, "snn_li_heads: LDS budget exceeded (NOp=%d)", NOp);
    void (*kern)(const uint32_t*, size_t, int, int, int, const float*, int, int, int, const Kappa, float*, float*,
                 float*, float*) = rb == 64 ? k_li_heads<64> : rb == 32 ? k_li_heads<32> : rb == 16 ? k_li_heads<16>
                                             : rb == 8 ? k_li_heads<8> : k_li_heads<4>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(-3, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(cdiv(M, rb)), dim3(256), lds, (hipStream_t)s, spk, spk_stride, T, M, Kw,
                       w_heads_packed, NOp, NA, NB, kap, out_a, out_b, sum_a, sum_b);
    SNN_CHECK_LAUNCH("k_li_heads");
    return 0;
}

// ---- whole heads ---------------------------------------------------------------------------------
static long long rpn_positions(const snn_rpn_level* lv, int n_levels, int* max_n) {
    long long P = 0; int mn = 0;
    for (int l = 0; l < n_levels; ++l) { P += (long long)lv[l].N * lv[l].H * lv[l].W; mn = lv[l].N > mn ? lv[l].N : mn; }
    if (max_n) *max_n = mn;
    return P;
}

static long long rpn_positions_padded(const snn_rpn_level* lv, int n_levels) {
    long long P = 0;
    for (int l = 0; l < n_levels; ++l) P += (long long)lv[l].N * (lv[l].H + 2) * (lv[l].W + 2);
    return P;
}

// Pe = rows of an encoder plane (positions; with the zero halo for the mxfp6 path)
static void rpn_ws_layout(long long P, long long Pe, int C, int T, int precision, size_t* o_spk, size_t* o_cur, size_t* o_cnt,
                          size_t* total) {
    const size_t plane = align_up((size_t)T * (Pe > P ? Pe : P) * cdiv(C, 32) * 4, 256);
    *o_spk = plane;
    *o_cur = 2 * plane;
    const size_t cur = 0;     // (only the un-fused snn_spike_conv3x3_bf16x3 + snn_lif_scan pair needs currents)
    (void)precision;
    *o_cnt = 2 * plane + cur;
    const size_t cnt = 0;
    *total = 2 * plane + cur + cnt;
}

size_t snn_rpn_head_workspace_bytes(const snn_rpn_level* lv, int n_levels, int C, int A, int T, int precision) {
    (void)A;
    if (!lv || n_levels <= 0 || n_levels > SNN_MAX_LEVELS || C <= 0 || T < 1) return 0;
    size_t a, b, c, tot;
    rpn_ws_layout(rpn_positions(lv, n_levels, nullptr), precision == SNN_PRECISION_MXFP6 ? rpn_positions_padded(lv, n_levels) : 0,
                  C, T, precision, &a, &b, &c, &tot);
    return tot;
}

int snn_rpn_head_forward_stages(const snn_rpn_level* lv, int n_levels, int C, int A, int T, const snn_params* p,
                                const void* w_shared_packed, const float* w_heads_packed, float* out_logits,
                                float* out_bbox, unsigned long long* spike_counts, float* sum_logits,
                                float* sum_bbox, void* ws, size_t ws_bytes, int stage_mask, snn_stream_t stream) {
    if (!lv || !p || !w_shared_packed || !w_heads_packed || !out_logits || !out_bbox || !ws)
        return fail(-1, "snn_rpn_head_forward: null argument");
    if (n_levels <= 0 || n_levels > SNN_MAX_LEVELS) return fail(-1, "snn_rpn_head_forward: n_levels=%d", n_levels);
    if (C <= 0 || A <= 0) return fail(-1, "snn_rpn_head_forward: bad C/A");
    if (p->precision != SNN_PRECISION_F32 && p->precision != SNN_PRECISION_BF16X3 && p->precision != SNN_PRECISION_MXFP6)
        return fail(-1, "snn_rpn_head_forward: unknown precision %d", p->precision);
    if (p->precision == SNN_PRECISION_MXFP6 && (C % 128 || !mx_tile_ok(T)))
        return fail(-4, "snn_rpn_head_forward: the mxfp6 kernels need C %% 128 == 0 and a T that fits a 512-row tile (C=%d, T=%d)", C, T);
    if (check_T(T, "snn_rpn_head_forward")) return -1;
    for (int l = 0; l < n_levels; ++l)
        if (!lv[l].feat || lv[l].N <= 0 || lv[l].H <= 0 || lv[l].W <= 0)
            return fail(-1, "snn_rpn_head_forward: bad level %d", l);
    int max_n = 0;
    const long long P = rpn_positions(lv, n_levels, &max_n);
    size_t o_spk, o_cur, o_cnt, need;
    const bool mxp = p->precision == SNN_PRECISION_MXFP6;          // encoder planes with a zero halo (k_gemm_mx)
    const long long Pe = mxp ? rpn_positions_padded(lv, n_levels) : P;
    rpn_ws_layout(P, mxp ? Pe : 0, C, T, p->precision, &o_spk, &o_cur, &o_cnt, &need);
    if (ws_bytes < need) return fail(-2, "snn_rpn_head_forward: workspace %zu < %zu bytes", ws_bytes, need);
    const int Cw = cdiv(C, 32);
    const size_t stride = (size_t)P * Cw;            // words per time plane (spike planes)
    const size_t enc_stride = (size_t)Pe * Cw;       // ... of the encoder planes
    uint32_t* enc = (uint32_t*)ws;
    uint32_t* spk = (uint32_t*)((char*)ws + o_spk);
    hipStream_t s = (hipStream_t)stream;
    if (stage_mask & SNN_STAGE_ENCODE) {                        // rpn.py:101, all levels in one launch
        EncLevels el;
        memset(&el, 0, sizeof(el));
        long long pos = 0;
        int blocks = 0;
        if (mxp && hipMemsetAsync(enc, 0, (size_t)T * enc_stride * 4, s) != hipSuccess) return fail(-3, "hipMemsetAsync failed");
        for (int l = 0; l < n_levels; ++l) {
            if (!lv[l].feat) return fail(-1, "snn_rpn_head_forward: level %d has no features", l);
            el.feat[l] = lv[l].feat; el.HW[l] = lv[l].H * lv[l].W; el.bpi[l] = cdiv(el.HW[l], ENC_PB);
            el.blk_base[l] = blocks; el.pos_base[l] = (int)pos; el.Wpad[l] = mxp ? lv[l].W : 0;
            blocks += lv[l].N * el.bpi[l];
            pos += mxp ? (long long)lv[l].N * (lv[l].H + 2) * (lv[l].W + 2) : (long long)lv[l].N * el.HW[l];
        }
        el.blk_base[n_levels] = blocks; el.n_levels = n_levels;
        const NeuronP np = make_p(p, p->v_th_enc);
        if (enc_zero_rest(np))
            hipLaunchKernelGGL(k_encode_levels<true>, dim3(blocks, cdiv(Cw, 8)), dim3(256), 0, s, el, C, Cw, T, np, enc, enc_stride);
        else
            hipLaunchKernelGGL(k_encode_levels<false>, dim3(blocks, cdiv(Cw, 8)), dim3(256), 0, s, el, C, Cw, T, np, enc, enc_stride);
        SNN_CHECK_LAUNCH("k_encode_levels");
    }
    if (stage_mask & SNN_STAGE_CONV_LIF) {
        if (p->precision == SNN_PRECISION_F32) {
            if (spike_counts) {
                hipError_t e = hipMemsetAsync(spike_counts, 0, sizeof(unsigned long long) * n_levels * max_n, s);
                if (e != hipSuccess) return fail(-3, "hipMemsetAsync failed: %s", hipGetErrorString(e));
            }
            int rc = launch_conv(lv, n_levels, C, C, T, p, enc, stride, (const float*)w_shared_packed, spk, stride,
                                 spike_counts, max_n, nullptr, s);
            if (rc) return rc;
        } else {
            // conv + LIF fused over T on the bf16 matrix cores (rpn.py:98-106): membrane state in registers, only
            // spike planes written (4.5 ms against 4.2 + 0.5 ms for snn_spike_conv3x3_bf16x3 + snn_lif_scan, which
            // give bit-identical planes)
            int rc = p->precision == SNN_PRECISION_MXFP6
                         ? snn_conv3x3_lif_mx(enc, enc_stride, lv, n_levels, C, C, T, p, (const uint32_t*)w_shared_packed, spk, stride, stream)
                         : snn_conv3x3_lif_bf16x3(enc, stride, lv, n_levels, C, C, T, p, (const uint16_t*)w_shared_packed,
                                                  spk, stride, stream);
            if (rc) return rc;
            if (spike_counts) {
                if (hipMemsetAsync(spike_counts, 0, sizeof(unsigned long long) * n_levels * max_n, s) != hipSuccess)
                    return fail(-3, "hipMemsetAsync failed");
                long long pb = 0;
                for (int l = 0; l < n_levels; ++l) {
                    const int hw = lv[l].H * lv[l].W;
                    hipLaunchKernelGGL(k_count_spikes, dim3(lv[l].N, max(1, min(256, hw * Cw / 2048))), dim3(256), 0, s, spk + (size_t)pb * Cw,
                                       (unsigned long long)stride, T, hw * Cw, spike_counts + (size_t)l * max_n);
                    SNN_CHECK_LAUNCH("k_count_spikes");
                    pb += (long long)lv[l].N * hw;
                }
            }
        }
    }
    if (!(stage_mask & SNN_STAGE_LI_HEADS)) return 0;
    return snn_li_heads(spk, stride, T, (int)P, C, w_heads_packed, A, 4 * A, p, out_logits, out_bbox, sum_logits,
                        sum_bbox, stream);
}

int snn_rpn_head_forward(const snn_rpn_level* lv, int n_levels, int C, int A, int T, const snn_params* p,
                         const void* w_shared_packed, const float* w_heads_packed, float* out_logits,
                         float* out_bbox, unsigned long long* spike_counts, float* sum_logits, float* sum_bbox,
                         void* ws, size_t ws_bytes, snn_stream_t stream) {
    return snn_rpn_head_forward_stages(lv, n_levels, C, A, T, p, w_shared_packed, w_heads_packed, out_logits,
                                       out_bbox, spike_counts, sum_logits, sum_bbox, ws, ws_bytes, SNN_STAGE_ALL,
                                       stream);
}

// ---- RPN proposal selection -------------------------------------------------------------------
int snn_rpn_proposals_candidates(const snn_rpn_post_level* lv, int n_levels, int A, int pre_nms_top_n) {
    if (!lv || n_levels <= 0 || n_levels > SNN_MAX_LEVELS || A <= 0 || pre_nms_top_n <= 0) return -1;
    long long k = 0;
    for (int l = 0; l < n_levels; ++l) k += min((long long)pre_nms_top_n, (long long)lv[l].H * lv[l].W * A);
    return k > 0x7fffffffLL ? -1 : (int)k;
}

static size_t rpn_post_layout(int N, int K, size_t off[14]) {
    const size_t nk = (size_t)N * K;
    const size_t sz[14] = {nk * 4, nk * 4, nk * 16, nk * 16, nk * 4, nk * 4, nk * 16, nk * 16, nk * 4, nk * 4,
                           (size_t)N * 4, nk * 4, (size_t)N * 4, nk * cdiv(K, 64) * 8};
    size_t o = 0;
    for (int i = 0; i < 14; ++i) { off[i] = o; o += align_up(sz[i], 256); }
    return o;
}

size_t snn_rpn_proposals_workspace_bytes(int N, int K_candidates) {
    size_t off[14];
    return (N > 0 && K_candidates > 0) ? rpn_post_layout(N, K_candidates, off) : 0;
}

int snn_rpn_proposals(const snn_rpn_post_level* lv, int n_levels, int N, int A, const float* image_hw_host,
                      int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float score_thresh, float min_size,
                      float* out_boxes, float* out_scores, int* out_counts, float* pre_boxes, float* pre_prob,
                      void* ws, size_t ws_bytes, snn_stream_t stream) {
    if (!lv || !image_hw_host || !out_boxes || !out_scores || !out_counts || !ws)
        return fail(-1, "snn_rpn_proposals: null argument");
    if (n_levels <= 0 || n_levels > SNN_MAX_LEVELS || N <= 0 || N > RPN_MAX_IMAGES || A <= 0 || A > RPN_MAX_ANCHORS ||
        pre_nms_top_n <= 0 || post_nms_top_n <= 0)
        return fail(-1, "snn_rpn_proposals: bad argument (levels <= %d, images <= %d, anchors <= %d)", SNN_MAX_LEVELS,
                    RPN_MAX_IMAGES, RPN_MAX_ANCHORS);
    const int K = snn_rpn_proposals_candidates(lv, n_levels, A, pre_nms_top_n);
    if (K <= 0 || K > RPN_SORT_MAX) return fail(-1, "snn_rpn_proposals: %d candidates per image (max %d)", K, RPN_SORT_MAX);
    size_t off[14];
    if (ws_bytes < rpn_post_layout(N, K, off)) return fail(-2, "snn_rpn_proposals: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    RpnPostArgs a;
    memset(&a, 0, sizeof(a));
    int koff = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (!lv[l].logits || !lv[l].deltas || lv[l].H <= 0 || lv[l].W <= 0) return fail(-1, "snn_rpn_proposals: bad level %d", l);
        if ((long long)lv[l].H * lv[l].W * A * N > 0x7fffffffLL || (long long)lv[l].H * lv[l].W * A >= (1 << 28))
            return fail(-1, "snn_rpn_proposals: level %d too large", l);
        RpnPostLevel& L = a.lv[l];
        L.logits = lv[l].logits; L.deltas = lv[l].deltas; L.H = lv[l].H; L.W = lv[l].W;
        L.n = lv[l].H * lv[l].W * A; L.k = min(pre_nms_top_n, L.n); L.koff = koff; koff += L.k;
        L.sh = lv[l].stride_h; L.sw = lv[l].stride_w;
        memcpy(L.base, lv[l].base_anchors, sizeof(float) * 4 * A);
    }
    for (int i = 0; i < N; ++i) { a.img_h[i] = image_hw_host[2 * i]; a.img_w[i] = image_hw_host[2 * i + 1]; }
    a.n_levels = n_levels; a.N = N; a.A = A; a.Ktot = K; a.post_n = post_nms_top_n;
    a.score_thresh = score_thresh; a.min_size = min_size; a.clip = (float)4.135166556742356;     // log(1000/16), boxes.py
    char* w = (char*)ws;
    a.cand_idx = (int*)(w + off[0]); a.cand_logit = (float*)(w + off[1]); a.boxes = (float*)(w + off[2]);
    a.pre = (float*)(w + off[3]); a.prob = (float*)(w + off[4]); a.skey = (float*)(w + off[5]);
    a.s_boxes = (float*)(w + off[6]); a.s_pre = pre_boxes ? pre_boxes : (float*)(w + off[7]);
    a.s_prob = pre_prob ? pre_prob : (float*)(w + off[8]); a.s_cat = (int*)(w + off[9]); a.n_valid = (int*)(w + off[10]);
    int* keep = (int*)(w + off[11]);
    int* n_keep = (int*)(w + off[12]);
    unsigned long long* mask = (unsigned long long*)(w + off[13]);
    hipLaunchKernelGGL(k_rpn_topk, dim3(n_levels, N), dim3(1024), 0, s, a);
    SNN_CHECK_LAUNCH("k_rpn_topk");
    hipLaunchKernelGGL(k_rpn_decode, dim3(cdiv((long long)N * K, 256)), dim3(256), 0, s, a);
    SNN_CHECK_LAUNCH("k_rpn_decode");
    hipLaunchKernelGGL(k_rpn_sort, dim3(N), dim3(1024), 0, s, a);
    SNN_CHECK_LAUNCH("k_rpn_sort");
    const int words = cdiv(K, 64);
    NmsBatch nb;
    nb.n_dev = a.n_valid; nb.boxes_stride = (long long)K * 4; nb.cat_stride = K; nb.mask_stride = (long long)K * words;
    nb.keep_stride = K;
    hipLaunchKernelGGL(k_nms_mask, dim3(words, words, N), dim3(64), 0, s, a.s_boxes, a.s_cat, K, nms_thresh, mask, words, nb);
    SNN_CHECK_LAUNCH("k_nms_mask");
    const int dbl = words <= 156 ? 1 : 0;
    const size_t lds = (size_t)(1 + dbl) * 64 * words * 8;
    hipError_t e = hipFuncSetAttribute((const void*)k_nms_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(-3, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_nms_scan, dim3(N), dim3(256), lds, s, mask, K, words, post_nms_top_n, dbl, keep, n_keep, nb);
    SNN_CHECK_LAUNCH("k_nms_scan");
    hipLaunchKernelGGL(k_rpn_output, dim3(N), dim3(256), 0, s, a, keep, n_keep, out_boxes, out_scores, out_counts);
    SNN_CHECK_LAUNCH("k_rpn_output");
    return 0;
}

// ---- detection post-processing ------------------------------------------------------------------
static size_t det_post_layout(int N, int Kcap, size_t off[11]) {
    const size_t L = 2 * (size_t)N, lk = L * Kcap;
    const size_t sz[11] = {lk * 16, lk * 4, lk * 4, lk * 16, lk * 4, lk * 4, L * 4, lk * 4, L * 4, lk * cdiv(Kcap, 64) * 8, 0};
    size_t o = 0;
    for (int i = 0; i < 11; ++i) { off[i] = o; o += align_up(sz[i], 256); }
    return o;
}

size_t snn_det_postprocess_workspace_bytes(int N, int max_rois_per_image, int K) {
    size_t off[11];
    if (N <= 0 || max_rois_per_image <= 0 || K < 2) return 0;
    return det_post_layout(N, max_rois_per_image * (K - 1), off);
}

int snn_det_postprocess(const float* class_logits, const float* box_regression, const float* proposals,
                        const int* rois_per_image_host, int N, int K, const float* image_hw_host,
                        const float* box_weights_host, float score_thresh, float nms_thresh, int detections_per_img,
                        float min_size, float* all_scores, float* all_boxes, float* out_boxes, float* out_scores,
                        int* out_labels, int* out_counts, int out_cap, void* ws, size_t ws_bytes, snn_stream_t stream) {
    if (!class_logits || !box_regression || !proposals || !rois_per_image_host || !image_hw_host || !box_weights_host || !all_scores ||
        !all_boxes || !out_boxes || !out_scores || !out_labels || !out_counts || !ws)
        return fail(-1, "snn_det_postprocess: null argument");
    if (N <= 0 || N > RPN_MAX_IMAGES || K < 2 || detections_per_img <= 0)
        return fail(-1, "snn_det_postprocess: bad argument (images <= %d)", RPN_MAX_IMAGES);
    DetPostArgs a;
    memset(&a, 0, sizeof(a));
    int rmax = 0;
    for (int i = 0; i < N; ++i) {
        if (rois_per_image_host[i] < 0) return fail(-1, "snn_det_postprocess: negative RoI count");
        a.roi_base[i + 1] = a.roi_base[i] + rois_per_image_host[i];
        rmax = max(rmax, rois_per_image_host[i]);
        a.img_h[i] = image_hw_host[2 * i]; a.img_w[i] = image_hw_host[2 * i + 1];
    }
    if (rmax == 0) {
        if (hipMemsetAsync(out_counts, 0, sizeof(int) * 2 * N, (hipStream_t)stream) != hipSuccess) return fail(-3, "hipMemsetAsync failed");
        return 0;
    }
    const int Kcap = rmax * (K - 1);
    if (Kcap > DET_SORT_MAX) return fail(-4, "snn_det_postprocess: %d candidates per image (max %d)", Kcap, DET_SORT_MAX);
    if (out_cap < detections_per_img + rmax) return fail(-1, "snn_det_postprocess: out_cap %d < %d", out_cap, detections_per_img + rmax);
    size_t off[11];
    if (ws_bytes < det_post_layout(N, Kcap, off)) return fail(-2, "snn_det_postprocess: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    a.logits = class_logits; a.deltas = box_regression; a.props = proposals;
    a.N = N; a.K = K; a.Kcap = Kcap; a.det_per_img = detections_per_img; a.out_cap = out_cap;
    a.score_thresh = score_thresh; a.min_size = min_size; a.clip = (float)4.135166556742356;
    a.wx = box_weights_host[0]; a.wy = box_weights_host[1]; a.ww = box_weights_host[2]; a.wh = box_weights_host[3];
    a.all_scores = all_scores; a.all_boxes = all_boxes;
    char* w = (char*)ws;
    a.boxes = (float*)(w + off[0]); a.skey = (float*)(w + off[1]); a.cat = (int*)(w + off[2]);
    a.s_boxes = (float*)(w + off[3]); a.s_score = (float*)(w + off[4]); a.s_cat = (int*)(w + off[5]); a.n_valid = (int*)(w + off[6]);
    int* keep = (int*)(w + off[7]);
    int* n_keep = (int*)(w + off[8]);
    unsigned long long* mask = (unsigned long long*)(w + off[9]);
    const int L = 2 * N;
    hipLaunchKernelGGL(k_det_candidates, dim3(cdiv(Kcap, 256), L), dim3(256), 0, s, a);
    SNN_CHECK_LAUNCH("k_det_candidates");
    int np2 = 1;
    while (np2 < Kcap) np2 <<= 1;
    const size_t sort_lds = (size_t)np2 * 8;
    hipError_t e = hipFuncSetAttribute((const void*)k_sort_lists, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds);
    if (e != hipSuccess) return fail(-3, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_sort_lists, dim3(L), dim3(1024), sort_lds, s, a.skey, a.boxes, a.cat, Kcap, a.s_boxes, a.s_score, a.s_cat, a.n_valid);
    SNN_CHECK_LAUNCH("k_sort_lists");
    const int words = cdiv(Kcap, 64);
    NmsBatch nb;
    nb.n_dev = a.n_valid; nb.boxes_stride = (long long)Kcap * 4; nb.cat_stride = Kcap; nb.mask_stride = (long long)Kcap * words;
    nb.keep_stride = Kcap;
    hipLaunchKernelGGL(k_nms_mask, dim3(words, words, L), dim3(64), 0, s, a.s_boxes, a.s_cat, Kcap, nms_thresh, mask, words, nb);
    SNN_CHECK_LAUNCH("k_nms_mask");
    const int dbl = words <= 156 ? 1 : 0;
    const size_t lds = (size_t)(1 + dbl) * 64 * words * 8;
    e = hipFuncSetAttribute((const void*)k_nms_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(-3, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_nms_scan, dim3(L), dim3(256), lds, s, mask, Kcap, words, Kcap, dbl, keep, n_keep, nb);
    SNN_CHECK_LAUNCH("k_nms_scan");
    hipLaunchKernelGGL(k_det_output, dim3(N), dim3(256), 0, s, a, keep, n_keep, out_boxes, out_scores, out_labels, out_counts);
    SNN_CHECK_LAUNCH("k_det_output");
    return 0;
}

static void det_ws_layout(int R, int D, int Hd, int T, size_t* o_enc, size_t* o_cur, size_t* o_s6, size_t* o_s7,
                          size_t* total) {
    const size_t enc = align_up((size_t)T * R * cdiv(D, 32) * 4, 256);
    const size_t cur = align_up((size_t)T * R * cdiv(Hd, 32) * 32 * 4, 256);
    const size_t sp = align_up((size_t)T * R * cdiv(Hd, 32) * 4, 256);
    *o_enc = 0; *o_cur = enc; *o_s6 = enc + cur; *o_s7 = enc + cur + sp; *total = enc + cur + 2 * sp;
}

size_t snn_det_head_workspace_bytes(int R, int D, int Hd, int K, int K4, int T, int precision) {
    (void)K; (void)K4; (void)precision;
    if (R <= 0 || D <= 0 || Hd <= 0 || T < 1) return 0;
    size_t a, b, c, d, tot;
    det_ws_layout(R, D, Hd, T, &a, &b, &c, &d, &tot);
    return tot;
}

static int det_head_from_planes(int R, int D, int Hd, int K, int K4, int T, const snn_params* p, const void* w6_packed,
                                const void* w7_packed, const float* w_heads_packed, float* out_cls, float* out_bbox,
                                uint32_t* spk6_count, uint32_t* spk7_count, float* sum_cls, float* sum_bbox, void* ws,
                                snn_stream_t stream) {
    size_t o_enc, o_cur, o_s6, o_s7, need;
    det_ws_layout(R, D, Hd, T, &o_enc, &o_cur, &o_s6, &o_s7, &need);
    hipStream_t s = (hipStream_t)stream;
    uint32_t* enc = (uint32_t*)((char*)ws + o_enc);
    float* cur = (float*)((char*)ws + o_cur);
    uint32_t* s6 = (uint32_t*)((char*)ws + o_s6);
    uint32_t* s7 = (uint32_t*)((char*)ws + o_s7);
    const int Hw = cdiv(Hd, 32), Hp = Hw * 32;
    int rc;
    if (spk6_count) { if (hipMemsetAsync(spk6_count, 0, sizeof(uint32_t) * R, s) != hipSuccess) return fail(-3, "hipMemsetAsync failed"); }
    if (spk7_count) { if (hipMemsetAsync(spk7_count, 0, sizeof(uint32_t) * R, s) != hipSuccess) return fail(-3, "hipMemsetAsync failed"); }
    const bool b3 = p->precision == SNN_PRECISION_BF16X3, mx = p->precision == SNN_PRECISION_MXFP6;
    if (p->precision != SNN_PRECISION_F32 && !b3 && !mx) return fail(-1, "snn_det_head_forward: unknown precision %d", p->precision);
    if (mx) {
        if (D % 128 || Hd % 128 || !mx_tile_ok(T))
            return fail(-4, "snn_det_head_forward: the mxfp6 kernels need D, Hd %% 128 == 0 and a T that fits a 512-row tile");
        if ((rc = snn_spike_gemm_lif_mx(enc, T, R, D, Hd, p, (const uint32_t*)w6_packed, s6, (size_t)R * Hw, stream))) return rc;
        if ((rc = snn_spike_gemm_lif_mx(s6, T, R, Hd, Hd, p, (const uint32_t*)w7_packed, s7, (size_t)R * Hw, stream))) return rc;
        if (spk6_count) {
            hipLaunchKernelGGL(k_count_rows, dim3(cdiv(R, 4)), dim3(256), 0, s, s6, (unsigned long long)R * Hw, T, R, Hw, spk6_count);
            SNN_CHECK_LAUNCH("k_count_rows");
        }
        if (spk7_count) {
            hipLaunchKernelGGL(k_count_rows, dim3(cdiv(R, 4)), dim3(256), 0, s, s7, (unsigned long long)R * Hw, T, R, Hw, spk7_count);
            SNN_CHECK_LAUNCH("k_count_rows");
        }
        return snn_li_heads(s7, (size_t)R * Hw, T, R, Hd, w_heads_packed, K, K4, p, out_cls, out_bbox, sum_cls,
                            sum_bbox, stream);
    }
    if (b3 && (g3_tile_ok(T, G3_BM(g3_wn(), 4)) || g3_tile_ok(T, G3_BM(g3_wn(), 3)) || g3_tile_ok(T, G3_BM(g3_wn(), 2)))) {
        // fc6 + LIF and fc7 + LIF, each one launch: a row tile holds all T steps of its RoIs, the currents never
        // leave the chip (faster_rcnn.py:498-501)
        if ((rc = snn_spike_gemm_lif_bf16x3(enc, T, R, D, Hd, p, (const uint16_t*)w6_packed, s6, (size_t)R * Hw, stream))) return rc;
        if ((rc = snn_spike_gemm_lif_bf16x3(s6, T, R, Hd, Hd, p, (const uint16_t*)w7_packed, s7, (size_t)R * Hw, stream))) return rc;
        if (spk6_count) {                                       // spike-rate mode: per-RoI counts from the planes
            hipLaunchKernelGGL(k_count_rows, dim3(cdiv(R, 4)), dim3(256), 0, s, s6, (unsigned long long)R * Hw, T, R, Hw, spk6_count);
            SNN_CHECK_LAUNCH("k_count_rows");
        }
        if (spk7_count) {
            hipLaunchKernelGGL(k_count_rows, dim3(cdiv(R, 4)), dim3(256), 0, s, s7, (unsigned long long)R * Hw, T, R, Hw, spk7_count);
            SNN_CHECK_LAUNCH("k_count_rows");
        }
        return snn_li_heads(s7, (size_t)R * Hw, T, R, Hd, w_heads_packed, K, K4, p, out_cls, out_bbox, sum_cls,
                            sum_bbox, stream);
    }
    // fc6 for all T steps at once: rows m = t*R + r   (faster_rcnn.py:498)
    rc = b3 ? snn_spike_gemm_bf16x3(enc, T * R, D, Hd, (const uint16_t*)w6_packed, cur, Hp, stream)
            : snn_spike_gemm(enc, T * R, D, Hd, (const float*)w6_packed, cur, Hp, stream);
    if (rc) return rc;
    if ((rc = snn_lif_scan(cur, T, R, Hd, Hp, p, s6, (size_t)R * Hw, spk6_count, stream))) return rc;   // :499
    rc = b3 ? snn_spike_gemm_bf16x3(s6, T * R, Hd, Hd, (const uint16_t*)w7_packed, cur, Hp, stream)
            : snn_spike_gemm(s6, T * R, Hd, Hd, (const float*)w7_packed, cur, Hp, stream);                       // :500
    if (rc) return rc;
    if ((rc = snn_lif_scan(cur, T, R, Hd, Hp, p, s7, (size_t)R * Hw, spk7_count, stream))) return rc;   // :501
    return snn_li_heads(s7, (size_t)R * Hw, T, R, Hd, w_heads_packed, K, K4, p, out_cls, out_bbox, sum_cls,
                        sum_bbox, stream);                                                               // :505-510
}

int snn_det_head_forward(const float* x, int R, int D, int Hd, int K, int K4, int T, const snn_params* p,
                         const void* w6_packed, const void* w7_packed, const float* w_heads_packed,
                         float* out_cls, float* out_bbox, uint32_t* spk6_count, uint32_t* spk7_count,
                         float* sum_cls, float* sum_bbox, void* ws, size_t ws_bytes, snn_stream_t stream) {
    if (!x || !p || !w6_packed || !w7_packed || !w_heads_packed || !out_cls || !out_bbox || !ws)
        return fail(-1, "snn_det_head_forward: null argument");
    if (R <= 0 || D <= 0 || Hd <= 0 || K <= 0 || K4 <= 0) return fail(-1, "snn_det_head_forward: bad shape");
    if (check_T(T, "snn_det_head_forward")) return -1;
    size_t o_enc, o_cur, o_s6, o_s7, need;
    det_ws_layout(R, D, Hd, T, &o_enc, &o_cur, &o_s6, &o_s7, &need);
    if (ws_bytes < need) return fail(-2, "snn_det_head_forward: workspace %zu < %zu bytes", ws_bytes, need);
    int rc = snn_encode_rows(x, R, D, T, p, (uint32_t*)((char*)ws + o_enc), (size_t)R * cdiv(D, 32), stream);
    if (rc) return rc;
    return det_head_from_planes(R, D, Hd, K, K4, T, p, w6_packed, w7_packed, w_heads_packed, out_cls, out_bbox,
                                spk6_count, spk7_count, sum_cls, sum_bbox, ws, stream);
}

int snn_det_head_forward_roialign(const snn_roi_level* levels_host, int n_levels, int C, const float* rois,
                                  const int* roi_batch, const int* roi_level, int R, int Hd, int K, int K4, int T,
                                  const snn_params* p, const void* w6_packed, const void* w7_packed,
                                  const float* w_heads_packed, float* out_cls, float* out_bbox, uint32_t* spk6_count,
                                  uint32_t* spk7_count, float* sum_cls, float* sum_bbox, void* ws, size_t ws_bytes,
                                  snn_stream_t stream) {
    if (!p || !w6_packed || !w7_packed || !w_heads_packed || !out_cls || !out_bbox || !ws)
        return fail(-1, "snn_det_head_forward_roialign: null argument");
    if (R <= 0 || C <= 0 || Hd <= 0 || K <= 0 || K4 <= 0) return fail(-1, "snn_det_head_forward_roialign: bad shape");
    const int D = C * 49;
    size_t o_enc, o_cur, o_s6, o_s7, need;
    det_ws_layout(R, D, Hd, T, &o_enc, &o_cur, &o_s6, &o_s7, &need);
    if (ws_bytes < need) return fail(-2, "snn_det_head_forward_roialign: workspace %zu < %zu bytes", ws_bytes, need);
    int rc = snn_roi_align_encode(levels_host, n_levels, C, rois, roi_batch, roi_level, R, T, p,
                                  (uint32_t*)((char*)ws + o_enc), (size_t)R * cdiv(D, 32), nullptr, stream);
    if (rc) return rc;
    return det_head_from_planes(R, D, Hd, K, K4, T, p, w6_packed, w7_packed, w_heads_packed, out_cls, out_bbox,
                                spk6_count, spk7_count, sum_cls, sum_bbox, ws, stream);
}

int snn_det_exchange_payload(const float* class_logits, const float* box_regression, int N, int rois_per_image, int K,
                             int max_det, float* payload, int* counts, snn_stream_t s) {
    if (!class_logits || !box_regression || !payload || !counts || N <= 0 || rois_per_image <= 0 || K < 2 || max_det <= 0)
        return fail(-1, "snn_det_exchange_payload: bad argument");
    if (rois_per_image > 4096 || K > 0xFFFF) return fail(-4, "snn_det_exchange_payload: %d RoIs per image (max 4096)", rois_per_image);
    int npad = 64;
    while (npad < rois_per_image) npad <<= 1;
    if (npad <= 1024)
        hipLaunchKernelGGL(k_det_payload<1>, dim3(N), dim3(npad), (size_t)npad * 8, (hipStream_t)s, class_logits,
                           box_regression, rois_per_image, K, max_det, npad, payload, counts);
    else
        hipLaunchKernelGGL(k_det_payload<4>, dim3(N), dim3(1024), (size_t)npad * 8, (hipStream_t)s, class_logits,
                           box_regression, rois_per_image, K, max_det, npad, payload, counts);
    SNN_CHECK_LAUNCH("k_det_payload");
    return 0;
}

}  // extern "C"
