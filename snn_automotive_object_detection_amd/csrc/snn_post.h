// callers on either side of the heads: batched NMS, RPN proposal selection, detection post-processing.
// Included by snn_kernels.hip.
#pragma once

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
                    // (readlane returns a signed int: without the uint32_t casts a row whose bit 31 is set sign-extends into
                    // bits 32..63 and removes the upper half of the chunk - found by the reference-pinned fixtures of round 2)
                    rc |= ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(dhi, b) << 32) |
                          (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(dlo, b);
                }
            }
            // kept boxes -> keep[] in order: lane b is kept box number base + popcount(kept_bits below b)
            if ((kept_bits >> t) & 1ull)
                keep[base + __popcll(kept_bits & ((1ull << t) - 1ull))] = c * 64 + t;
            if (t == 0) { count_s = count; kept_cur = kept_bits; }
        }
        __syncthreads();
        const unsigned long long kept = kept_cur;
        if (t < words && t > c) {             // later words: OR the rows of the kept boxes (all 64 rows read, the others masked:
            unsigned long long acc = removed; // independent pipelined LDS loads instead of one dependent load per kept box)
#pragma unroll 16
            for (int b = 0; b < 64; ++b) {
                const unsigned long long v = cur[b * words + t];
                acc |= ((kept >> b) & 1ull) ? v : 0ull;
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
// List form of the same NMS for the two post-processing pipelines (round 2).  batched_nms only ever suppresses inside a
// category (pyramid level / class), so every (image, category) pair is its own LIST: candidates already in score order,
// mask and in-order walk per list, all lists of the batch in one launch each, and a rank-by-binary-search merge of the
// kept candidates afterwards.  Against one walk over all categories of an image (round 1: 76 chunks of 64 for 4864 RPN
// candidates, 125 for 8000 detector candidates) a list is 16 chunks, and the lists run side by side.
// ------------------------------------------------------------------------------------------------
#define NMS_MAX_CAT 96
struct NmsLists {
    const float* boxes;           // candidate boxes of the batch; list (img, c) starts at (img * img_stride + off[c]) * 4
    const float* skey;            // nullable: candidate i of a list is out of the game when skey[...] < 0 (filtered before NMS)
    const int* n_dev;             // nullable: candidates per list on the device [N * L]; else n[c]
    long long img_stride;         // candidates per image (also the stride of keep[])
    long long mask_img;           // mask words (uint64) per image
    int L;                        // lists (categories) per image
    int off[NMS_MAX_CAT];         // first candidate of category c inside the image (boxes, skey, keep[])
    int cap[NMS_MAX_CAT];         // candidates of category c (capacity; the mask row of the list has ceil(cap / 64) words)
    long long moff[NMS_MAX_CAT];  // first mask word of category c inside the image
    int max_keep0, max_keep;      // walk stops after that many kept boxes (category 0 / the others)
    // torchvision's batched_nms on the CPU (the reference's path) runs ONE nms over all categories of a call when the call has
    // at most 4000 box coordinates, after shifting category c by c * (max coordinate + 1): the IoUs are then those of the SHIFTED
    // fp32 coordinates, and one within ~1e-6 of the threshold can be decided differently from the raw coordinates.  Reproduced
    // here: categories >= trick_c0 of an image form one call; trick_cnt / trick_max give, per list, its number of boxes and its
    // largest coordinate (as float bits; all coordinates are >= 0).  nullptr: raw coordinates (the > 4000 strategy).
    const int* trick_cnt;
    const uint32_t* trick_max;
    int trick_c0;
};

__device__ __forceinline__ float nms_trick_offset(const NmsLists& nl, int img, int c) {
    if (!nl.trick_cnt || c < nl.trick_c0) return 0.0f;
    int n = 0;
    uint32_t mx = 0u;
    for (int q = nl.trick_c0; q < nl.L; ++q) {
        n += nl.trick_cnt[img * nl.L + q];
        mx = max(mx, nl.trick_max[img * nl.L + q]);
    }
    if (4 * n > 4000) return 0.0f;                                // _batched_nms_vanilla: per category, raw coordinates
    return __fmul_rn((float)c, __fadd_rn(__uint_as_float(mx), 1.0f));   // idxs.to(boxes) * (max_coordinate + 1)
}

__global__ __launch_bounds__(64) void k_nms_mask_lists(const NmsLists nl, float thr, unsigned long long* __restrict__ mask) {
    const int rb = blockIdx.y, cb = blockIdx.x, z = blockIdx.z, img = z / nl.L, c = z % nl.L;
    if (cb < rb) return;                                  // only j > i matters
    const int n = nl.n_dev ? nl.n_dev[z] : nl.cap[c];
    if (rb * 64 >= n || cb * 64 >= n) return;
    const float* boxes = nl.boxes + ((size_t)img * nl.img_stride + nl.off[c]) * 4;
    const int words = (nl.cap[c] + 63) >> 6;
    mask += (size_t)img * nl.mask_img + nl.moff[c];
    __shared__ float cbx[64][4];
    const int t = threadIdx.x;
    const int j0 = cb * 64;
    const float off = nms_trick_offset(nl, img, c);      // 0, or the category's shift of torchvision's coordinate trick
    if (j0 + t < n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cbx[t][q] = __fadd_rn(boxes[(size_t)(j0 + t) * 4 + q], off);
    }
    __syncthreads();
    const int i = rb * 64 + t;
    if (i >= n) return;
    const float x1 = __fadd_rn(boxes[(size_t)i * 4], off), y1 = __fadd_rn(boxes[(size_t)i * 4 + 1], off);
    const float x2 = __fadd_rn(boxes[(size_t)i * 4 + 2], off), y2 = __fadd_rn(boxes[(size_t)i * 4 + 3], off);
    const float area_i = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    unsigned long long bits = 0;
    const int jn = min(64, n - j0);
    for (int jj = (rb == cb ? t + 1 : 0); jj < jn; ++jj) {
        const float w = fmaxf(__fsub_rn(fminf(x2, cbx[jj][2]), fmaxf(x1, cbx[jj][0])), 0.0f);
        const float h = fmaxf(__fsub_rn(fminf(y2, cbx[jj][3]), fmaxf(y1, cbx[jj][1])), 0.0f);
        const float inter = __fmul_rn(w, h);
        const float area_j = __fmul_rn(__fsub_rn(cbx[jj][2], cbx[jj][0]), __fsub_rn(cbx[jj][3], cbx[jj][1]));
        const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_i, area_j), inter));     // box_iou's formula
        if (iou > thr) bits |= 1ull << jj;
    }
    mask[(size_t)i * words + cb] = bits;
}

// the walk of k_nms_scan on one list per work-group; candidates flagged by skey < 0 start out removed
__global__ __launch_bounds__(256) void k_nms_scan_lists(const NmsLists nl, const unsigned long long* __restrict__ mask,
                                                        int* __restrict__ keep, int* __restrict__ n_keep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* rows = reinterpret_cast<unsigned long long*>(smem);     // [2][64][words]
    const int z = blockIdx.x, img = z / nl.L, c = z % nl.L;
    const int n = nl.n_dev ? nl.n_dev[z] : nl.cap[c];
    const int max_keep = c == 0 ? nl.max_keep0 : nl.max_keep;
    const int words = (nl.cap[c] + 63) >> 6;
    mask += (size_t)img * nl.mask_img + nl.moff[c];
    keep += (size_t)img * nl.img_stride + nl.off[c];
    n_keep += z;
    if (n <= 0) { if (threadIdx.x == 0) *n_keep = 0; return; }
    __shared__ unsigned long long removed_cur, kept_cur;
    __shared__ int count_s;
    const int t = threadIdx.x;
    const int n_chunks = (n + 63) / 64;
    unsigned long long removed = 0;          // thread t (< words): removed bits of boxes [64t, 64t+64)
    if (nl.skey && t < n_chunks) {
        const float* sk = nl.skey + (size_t)img * nl.img_stride + nl.off[c] + t * 64;
        const int m = min(64, n - t * 64);
        for (int b = 0; b < m; ++b) removed |= (unsigned long long)(sk[b] < 0.0f) << b;
    }
    auto copy_chunk = [&](int cc, int first, int step) {      // rows of chunk cc, words [cc, words) -> buffer cc & 1
        const int rn = min(64, n - cc * 64), wn = n_chunks - cc;
        unsigned long long* dst = rows + (size_t)(cc & 1) * 64 * words;
        const unsigned long long* src = mask + (size_t)cc * 64 * words;
        for (int idx = first; idx < rn * wn; idx += step) {
            const int r = idx / wn, w = cc + idx % wn;
            dst[r * words + w] = src[(size_t)r * words + w];
        }
    };
    copy_chunk(0, t, 256);
    if (t == 0) { count_s = 0; removed_cur = removed; }
    __syncthreads();
    for (int cc = 0; cc < n_chunks; ++cc) {
        const int rn = min(64, n - cc * 64);
        const unsigned long long* cur = rows + (size_t)(cc & 1) * 64 * words;
        if (t >= 64) {
            if (cc + 1 < n_chunks) copy_chunk(cc + 1, t - 64, 192);
        } else {                              // wave 0: the chunk's own block, resolved in registers
            const unsigned long long diag = t < rn ? cur[t * words + cc] : 0ull;
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
                    rc |= ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(dhi, b) << 32) |
                          (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(dlo, b);
                }
            }
            if ((kept_bits >> t) & 1ull)
                keep[base + __popcll(kept_bits & ((1ull << t) - 1ull))] = cc * 64 + t;
            if (t == 0) { count_s = count; kept_cur = kept_bits; }
        }
        __syncthreads();
        const unsigned long long kept = kept_cur;
        if (t < n_chunks && t > cc) {         // later words: OR the rows of the kept boxes.  All 64 rows are read (independent,
            unsigned long long acc = removed; // pipelined LDS loads; rows of boxes that were not kept are masked out) instead of
#pragma unroll 16                             // one dependent load per kept box
            for (int b = 0; b < 64; ++b) {
                const unsigned long long v = cur[b * words + t];
                acc |= ((kept >> b) & 1ull) ? v : 0ull;
            }
            removed = acc;
        }
        if (t == cc + 1) removed_cur = removed;
        const bool done = count_s >= max_keep;
        __syncthreads();
        if (done) break;
    }
    if (t == 0) *n_keep = count_s;
}

// ------------------------------------------------------------------------------------------------
// RPN proposal selection (rpn.py:420-499 + 262-296 + the box coder), one call for the batch:
//   k_topk_*      per (level, image): the pre_nms_top_n largest logits by a 3-pass radix select (11+11+10 bits) over
//                 TOPK_CH work-groups per list, then sorted by decreasing logit: candidate slots are in the reference's
//                 order (level, rank inside the level)
//   k_rpn_decode  per candidate: anchor from the level geometry, box decode, sigmoid, clip, size/score filters; the decoded
//                 un-clipped boxes + probabilities of ALL candidates are the reference's pre-NMS report (rpn.py:493-499)
//   k_nms_mask_lists / k_nms_scan_lists   one list per (image, level): its candidates already are in score order
//   k_rpn_merge   kept candidates of the level lists -> the reference's output order by rank computation, first
//                 post_nms_top_n -> [N][post_nms_top_n] padded + counts
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
    int* list_cnt;                // [N][n_levels] candidates that pass the filters (zeroed by the caller)
    uint32_t* list_max;           // [N][n_levels] their largest coordinate, as float bits (coordinates are >= 0)
};

__device__ __forceinline__ uint32_t f2key(float f) {           // monotone: larger float -> larger key
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- top-k per (level, image): 3-pass radix select (11 + 11 + 10 key bits) spread over TOPK_CH work-groups per list --------
// Round 1 ran one work-group per list: four sweeps over the 221 184 logits of the stride-4 level on ONE CU (0.3-0.45 ms, the
// largest kernel of the RPN post-processing).  Now every sweep is its own launch over TOPK_CH chunks per list:
//   k_topk_hist   chunk histogram of the key bits of this pass in LDS (one round of wave aggregation: objectness logits share
//                 their high bits, so most keys of a wave fall into one bin), non-empty bins added to the list's histogram
//   k_topk_pick   the bin in which the count of keys, from the top bin down, reaches `need` (block-wide scan) -> next prefix
//   k_topk_gather keys above the k-th key take slots by atomic counter; keys EQUAL to it are taken in element order (the
//                 chunk's first tie rank = ties of the chunks before it, from the per-chunk histograms of the last pass)
//   k_topk_sort   the k selected (key, ~index) pairs sorted in LDS: candidate slots come out in the reference's order
//                 (objectness.topk: decreasing logit; equal logits by element index here)
#define TOPK_CH 32
struct TopkState { uint32_t prefix, need, cnt_gt, pad; };

__global__ __launch_bounds__(256) void k_topk_hist(const RpnPostArgs a, int pass, const TopkState* __restrict__ state,
                                                   uint32_t* __restrict__ g_hist, uint32_t* __restrict__ g_hist3) {
    const RpnPostLevel& L = a.lv[blockIdx.y];
    const int img = blockIdx.z, list = img * a.n_levels + blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* src = L.logits + (size_t)img * L.n;
    const int chunk = (L.n + TOPK_CH - 1) / TOPK_CH, lo = ch * chunk, hi = min(L.n, lo + chunk);
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const uint32_t bm = pass == 2 ? 1023u : 2047u;
    const uint32_t pmask = pass == 0 ? 0u : (pass == 1 ? 0xFFE00000u : 0xFFFFFC00u);
    const uint32_t prefix = pass == 0 ? 0u : state[list].prefix;
    __shared__ uint32_t hist[2048];
    for (int b = tid; b < 2048; b += 256) hist[b] = 0;
    __syncthreads();
    for (int e0 = lo; e0 < hi; e0 += 1024) {
        uint32_t keys[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256 + tid;
            keys[u] = e < hi ? f2key(src[e]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256 + tid;
            bool act = e < hi && (keys[u] & pmask) == prefix;
            const uint32_t bin = (keys[u] >> shift) & bm;
            const unsigned long long m = __ballot(act);
            if (m != 0ull) {                                     // one round of wave aggregation on the leader's bin
                const int leader = __ffsll((long long)m) - 1;
                const uint32_t lb = (uint32_t)__shfl((int)bin, leader);
                const bool same = act && bin == lb;
                const unsigned long long sm = __ballot(same);
                if (lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(sm));
                act = act && !same;
            }
            if (act) atomicAdd(&hist[bin], 1u);
        }
    }
    __syncthreads();
    for (int b = tid; b < 2048; b += 256) {
        const uint32_t v = hist[b];
        if (v) atomicAdd(&g_hist[(size_t)list * 2048 + b], v);
        if (pass == 2 && b < 1024) g_hist3[((size_t)list * TOPK_CH + ch) * 1024 + b] = v;
    }
}

__global__ __launch_bounds__(1024) void k_topk_pick(const RpnPostArgs a, int pass, TopkState* __restrict__ state,
                                                    uint32_t* __restrict__ g_hist) {
    const int list = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const RpnPostLevel& L = a.lv[list % a.n_levels];
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const uint32_t prefix = pass == 0 ? 0u : state[list].prefix, need = pass == 0 ? (uint32_t)L.k : state[list].need;
    uint32_t* h = g_hist + (size_t)list * 2048;
    // thread r counts bins 2047 - 2r and 2046 - 2r: `before` = keys in the bins above them
    const int b_hi = 2047 - 2 * tid, b_lo = 2046 - 2 * tid;
    const uint32_t h_hi = h[b_hi], h_lo = h[b_lo];
    h[b_hi] = 0; h[b_lo] = 0;                                    // ready for the next pass / the next call
    uint32_t v = h_hi + h_lo, inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
    }
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t s_prefix, s_need;
    if (lane == 63) wtot[wv] = inc;
    if (tid == 0) { s_prefix = prefix; s_need = need; }          // (fewer than `need` keys match: bin 0)
    __syncthreads();
    uint32_t before = inc - v;
    for (int w = 0; w < wv; ++w) before += wtot[w];
    if (before < need && need <= before + h_hi) { s_prefix = prefix | ((uint32_t)b_hi << shift); s_need = need - before; }
    else if (before + h_hi < need && need <= before + v) { s_prefix = prefix | ((uint32_t)b_lo << shift); s_need = need - before - h_hi; }
    __syncthreads();
    if (tid == 0) { state[list].prefix = s_prefix; state[list].need = s_need; state[list].cnt_gt = 0; }
}

__global__ __launch_bounds__(256) void k_topk_gather(const RpnPostArgs a, TopkState* __restrict__ state,
                                                     const uint32_t* __restrict__ g_hist3, unsigned long long* __restrict__ sel,
                                                     int kcap) {
    const RpnPostLevel& L = a.lv[blockIdx.y];
    const int img = blockIdx.z, list = img * a.n_levels + blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* src = L.logits + (size_t)img * L.n;
    const int chunk = (L.n + TOPK_CH - 1) / TOPK_CH, lo = ch * chunk, hi = min(L.n, lo + chunk);
    const uint32_t prefix = state[list].prefix, need = state[list].need, n_gt = (uint32_t)L.k - need;
    unsigned long long* out = sel + (size_t)list * kcap;
    // ties of the chunks before this one (element order = chunk order)
    __shared__ uint32_t s_base;
    __shared__ uint32_t wsum[4];
    if (tid < 64) {
        uint32_t t = 0;
        for (int c = tid; c < ch; c += 64) t += g_hist3[((size_t)list * TOPK_CH + c) * 1024 + (prefix & 1023u)];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += (uint32_t)__shfl_xor((int)t, off);
        if (tid == 0) s_base = t;
    }
    __syncthreads();
    uint32_t taken = s_base;                                     // block-uniform: tie rank of the next tie of this chunk
    for (int e0 = lo; e0 < hi; e0 += 256) {
        const int e = e0 + tid;
        const uint32_t key = e < hi ? f2key(src[e]) : 0u;
        {                                                        // keys above the k-th key: one slot counter bump per wave
            const bool gt = e < hi && key > prefix;
            const unsigned long long gm = __ballot(gt);
            if (gm) {
                uint32_t base = 0;
                if (lane == __ffsll(gm) - 1) base = atomicAdd(&state[list].cnt_gt, (uint32_t)__popcll(gm));
                base = (uint32_t)__shfl((int)base, __ffsll(gm) - 1);
                if (gt) out[base + (uint32_t)__popcll(gm & ((1ull << lane) - 1ull))] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)e);
            }
        }
        if (taken < need) {                                      // block-uniform
            const bool tie = e < hi && key == prefix;
            const unsigned long long bal = __ballot(tie);
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int w = 0; w < 4; ++w) { if (w < wv) before += wsum[w]; total += wsum[w]; }
            const uint32_t rank = taken + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (tie && rank < need) out[n_gt + rank] = ((unsigned long long)prefix << 32) | (uint32_t)(~(uint32_t)e);
            taken += total;
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(1024) void k_topk_sort(const RpnPostArgs a, const unsigned long long* __restrict__ sel, int kcap) {
    const RpnPostLevel& L = a.lv[blockIdx.x];
    const int img = blockIdx.y, list = img * a.n_levels + blockIdx.x, tid = threadIdx.x;
    const float* src = L.logits + (size_t)img * L.n;
    __shared__ unsigned long long v[RPN_SORT_MAX];
    int np2 = 1;
    while (np2 < L.k) np2 <<= 1;
    for (int i = tid; i < np2; i += 1024) v[i] = i < L.k ? sel[(size_t)list * kcap + i] : 0ull;     // padding sorts last
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long x = v[i], y = v[p];
                    if (((i & k) == 0) ? x < y : x > y) { v[i] = y; v[p] = x; }
                }
            }
            __syncthreads();
        }
    int* out_idx = a.cand_idx + (size_t)img * a.Ktot + L.koff;
    float* out_logit = a.cand_logit + (size_t)img * a.Ktot + L.koff;
    for (int i = tid; i < L.k; i += 1024) {
        const int e = (int)(~(uint32_t)v[i]);
        out_idx[i] = e;
        out_logit[i] = src[e];
    }
}

__global__ __launch_bounds__(256) void k_rpn_decode(const RpnPostArgs a) {
    const int g_raw = blockIdx.x * 256 + threadIdx.x;
    const bool in = g_raw < a.N * a.Ktot;
    const int g = in ? g_raw : a.N * a.Ktot - 1;                // lanes past the end recompute the last candidate, store nothing
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
    const bool valid = in && __fsub_rn(bx2, bx1) >= a.min_size && __fsub_rn(by2, by1) >= a.min_size && prob >= a.score_thresh;
    if (in) {
        reinterpret_cast<float4*>(a.pre)[g] = make_float4(x1, y1, x2, y2);
        reinterpret_cast<float4*>(a.boxes)[g] = make_float4(bx1, by1, bx2, by2);
        a.prob[g] = prob;
        a.skey[g] = valid ? prob : -1.0f;
    }
    // what batched_nms's coordinate trick needs per list (image, level): boxes of the call, largest coordinate.  Consecutive
    // candidates share their list, so a wave combines its lanes first and sends one atomic pair per list it holds (one lane per
    // candidate on ten addresses serialised ~2 x 970 atomics per address: 0.13 of the kernel's 0.15 ms)
    const int list = valid ? img * a.n_levels + l : -1;
    const uint32_t mx = valid ? __float_as_uint(fmaxf(bx2, by2)) : 0u;       // coordinates are >= 0: float order = bit order
    const int lane = threadIdx.x & 63;
    unsigned long long rem = __ballot(valid);
    while (rem) {
        const int leader = __ffsll(rem) - 1;
        const int sl = __shfl(list, leader);
        const bool mine = valid && list == sl;
        const unsigned long long mm = __ballot(mine);
        uint32_t v = mine ? mx : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
        if (lane == leader) {
            atomicAdd(&a.list_cnt[sl], (int)__popcll(mm));
            atomicMax(&a.list_max[sl], v);
        }
        rem &= ~mm;
    }
}

// Kept candidates of the L level lists of an image -> the reference's output order (decreasing score; equal scores in
// candidate order = level, then rank inside the level) -> the first post_n of them.  Every list is already in that order, so
// the position of a kept candidate is its index in its own list + the number of candidates of every other list that precede
// it (one binary search each, on the kept probabilities gathered in LDS): no sort.
__global__ __launch_bounds__(1024) void k_rpn_merge(const RpnPostArgs a, const int* __restrict__ keep, const int* __restrict__ n_keep,
                                                    float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                    int* __restrict__ out_counts) {
    __shared__ float kp[RPN_SORT_MAX];
    __shared__ int kbase[SNN_MAX_LEVELS + 1];
    const int img = blockIdx.x, tid = threadIdx.x, L = a.n_levels;
    if (tid == 0) {
        int base = 0;
        for (int l = 0; l < L; ++l) { kbase[l] = base; base += min(n_keep[img * L + l], a.post_n); }
        kbase[L] = base;
    }
    __syncthreads();
    const int total = kbase[L];
    for (int i = tid; i < total; i += 1024) {
        int l = 0;
        while (i >= kbase[l + 1]) ++l;
        const int slot = a.lv[l].koff + keep[(size_t)img * a.Ktot + a.lv[l].koff + (i - kbase[l])];
        kp[i] = a.prob[(size_t)img * a.Ktot + slot];
    }
    __syncthreads();
    for (int i = tid; i < total; i += 1024) {
        int l = 0;
        while (i >= kbase[l + 1]) ++l;
        const float pr = kp[i];
        int rank = i - kbase[l];
        for (int l2 = 0; l2 < L; ++l2) {
            if (l2 == l) continue;
            int lo = 0, hi = kbase[l2 + 1] - kbase[l2];
            const float* q = kp + kbase[l2];
            while (lo < hi) {                               // first entry of list l2 that does NOT precede (pr, l)
                const int mid = (lo + hi) >> 1;
                const bool precedes = q[mid] > pr || (q[mid] == pr && l2 < l);
                if (precedes) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < a.post_n) {
            const int slot = a.lv[l].koff + keep[(size_t)img * a.Ktot + a.lv[l].koff + (i - kbase[l])];
            reinterpret_cast<float4*>(out_boxes)[(size_t)img * a.post_n + rank] = reinterpret_cast<const float4*>(a.boxes)[(size_t)img * a.Ktot + slot];
            out_scores[(size_t)img * a.post_n + rank] = pr;
        }
    }
    const int cnt = min(total, a.post_n);
    for (int r = cnt + tid; r < a.post_n; r += 1024) {
        reinterpret_cast<float4*>(out_boxes)[(size_t)img * a.post_n + r] = make_float4(0.f, 0.f, 0.f, 0.f);
        out_scores[(size_t)img * a.post_n + r] = 0.f;
    }
    if (tid == 0) out_counts[img] = cnt;
}

// ------------------------------------------------------------------------------------------------
// Detection post-processing (roi_heads.py:1075-1176, the reference's variant that also reports background boxes),
// one call for the batch.  Per image K candidate lists, one per class (class 0 = background: RoIs without any class above
// the score threshold, class-0 box), slot = RoI:
//   k_det_candidates  per (list, RoI): softmax, BoxCoder(10,10,5,5).decode, clip, score / size filters; also all_scores / all_boxes
//   k_sort_lists      per list: bitonic sort by decreasing score (ties: lower RoI, as the reference's stable sort) in LDS, gather
//   k_nms_mask_lists / k_nms_scan_lists  (a class list stops after detections_per_img kept boxes: it cannot contribute more)
//   k_det_merge       kept candidates of the foreground lists -> decreasing score (ties: reference candidate index
//                     RoI * (K-1) + class - 1) by rank computation, first detections_per_img, then the background list
// ------------------------------------------------------------------------------------------------
#define DET_SORT_MAX 16384
struct DetPostArgs {
    const float* logits;          // [R][K]
    const float* deltas;          // [R][4K]
    const float* props;           // [R][4]
    int roi_base[RPN_MAX_IMAGES + 1];
    float img_h[RPN_MAX_IMAGES], img_w[RPN_MAX_IMAGES];
    int N, K, Rmax, det_per_img, out_cap;
    float score_thresh, min_size, clip, wx, wy, ww, wh;
    float* all_scores; float* all_boxes;                       // [R][K], [R][K][4]
    float* boxes; float* skey;                                 // [N*K][Rmax] candidate lists
    float* s_boxes; float* s_score; int* s_roi; int* n_valid;   // sorted
};

__global__ __launch_bounds__(256) void k_det_candidates(const DetPostArgs a) {
    const int list = blockIdx.y, img = list / a.K, k = list % a.K, bg = k == 0;
    const int rl = blockIdx.x * 256 + threadIdx.x;
    if (rl >= a.Rmax) return;
    const int Ri = a.roi_base[img + 1] - a.roi_base[img];
    const size_t o = (size_t)list * a.Rmax + rl;
    if (rl >= Ri) { a.skey[o] = -1.0f; reinterpret_cast<float4*>(a.boxes)[o] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
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
}

// one block per list: order = decreasing (score, then lower slot); n_valid = candidates with score >= 0
__global__ __launch_bounds__(1024) void k_sort_lists(const float* __restrict__ skey, const float* __restrict__ boxes, int Kcap,
                                                     float* __restrict__ s_boxes, float* __restrict__ s_score,
                                                     int* __restrict__ s_slot, int* __restrict__ n_valid, uint32_t* __restrict__ list_max) {
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
    uint32_t mx = 0u;                                           // largest coordinate of the valid candidates (float bits, >= 0)
    for (int i = tid; i < Kcap; i += 1024) {
        const int c = (int)(~(uint32_t)v[i]);
        const size_t src = (size_t)list * Kcap + c, dst = (size_t)list * Kcap + i;
        const float4 bx = reinterpret_cast<const float4*>(boxes)[src];
        reinterpret_cast<float4*>(s_boxes)[dst] = bx;
        const float sc = skey[src];
        s_score[dst] = sc;
        s_slot[dst] = c;
        valid += sc >= 0.0f;
        if (sc >= 0.0f) mx = max(mx, __float_as_uint(fmaxf(bx.z, bx.w)));
    }
    for (int off = 32; off > 0; off >>= 1) { valid += __shfl_down(valid, off); mx = max(mx, (uint32_t)__shfl_down((int)mx, off)); }
    __syncthreads();                                            // v[] is free now
    int* part = reinterpret_cast<int*>(smem);
    if ((tid & 63) == 0) { part[tid >> 6] = valid; part[16 + (tid >> 6)] = (int)mx; }
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        uint32_t m = 0u;
        for (int w = 0; w < 16; ++w) { t += part[w]; m = max(m, (uint32_t)part[16 + w]); }
        n_valid[list] = t;
        list_max[list] = m;
    }
}

#define DET_MERGE_MAX 8192                         // kept foreground candidates of an image that can be ranked ((K-1) * detections_per_img)
__global__ __launch_bounds__(1024) void k_det_merge(const DetPostArgs a, const int* __restrict__ keep, const int* __restrict__ n_keep,
                                                    float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                    int* __restrict__ out_labels, int* __restrict__ out_counts) {
    __shared__ float ks[DET_MERGE_MAX];             // score of the kept foreground candidates, class after class
    __shared__ int ki[DET_MERGE_MAX];               // their reference candidate index: RoI * (K-1) + class - 1
    __shared__ int kbase[NMS_MAX_CAT + 1];
    const int img = blockIdx.x, tid = threadIdx.x, K = a.K;
    if (tid == 0) {
        int base = 0;
        for (int c = 1; c < K; ++c) { kbase[c] = base; base += min(n_keep[img * K + c], a.det_per_img); }
        kbase[K] = base;
    }
    __syncthreads();
    const int total = kbase[K];
    for (int i = tid; i < total; i += 1024) {
        int c = 1;
        while (i >= kbase[c + 1]) ++c;
        const size_t src = (size_t)(img * K + c) * a.Rmax + keep[(size_t)(img * K + c) * a.Rmax + (i - kbase[c])];
        ks[i] = a.s_score[src];
        ki[i] = a.s_roi[src] * (K - 1) + (c - 1);
    }
    __syncthreads();
    for (int i = tid; i < total; i += 1024) {
        int c = 1;
        while (i >= kbase[c + 1]) ++c;
        const float sc = ks[i];
        const int ci = ki[i];
        int rank = i - kbase[c];
        for (int c2 = 1; c2 < K; ++c2) {
            if (c2 == c) continue;
            int lo = 0, hi = kbase[c2 + 1] - kbase[c2];
            const float* q = ks + kbase[c2];
            const int* qi = ki + kbase[c2];
            while (lo < hi) {                               // first entry of list c2 that does NOT precede (sc, ci)
                const int mid = (lo + hi) >> 1;
                const bool precedes = q[mid] > sc || (q[mid] == sc && qi[mid] < ci);
                if (precedes) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < a.det_per_img) {
            const size_t src = (size_t)(img * K + c) * a.Rmax + keep[(size_t)(img * K + c) * a.Rmax + (i - kbase[c])];
            const size_t dst = (size_t)img * a.out_cap + rank;
            reinterpret_cast<float4*>(out_boxes)[dst] = reinterpret_cast<const float4*>(a.s_boxes)[src];
            out_scores[dst] = sc;
            out_labels[dst] = c;
        }
    }
    const int n_fg = min(total, a.det_per_img), n_bg = n_keep[img * K];
    for (int r = tid; r < n_bg; r += 1024) {                   // the surviving background boxes, in their own score order
        const size_t src = (size_t)(img * K) * a.Rmax + keep[(size_t)(img * K) * a.Rmax + r];
        const size_t dst = (size_t)img * a.out_cap + n_fg + r;
        reinterpret_cast<float4*>(out_boxes)[dst] = reinterpret_cast<const float4*>(a.s_boxes)[src];
        out_scores[dst] = a.s_score[src];
        out_labels[dst] = 0;
    }
    if (tid == 0) { out_counts[2 * img] = n_fg; out_counts[2 * img + 1] = n_bg; }
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
