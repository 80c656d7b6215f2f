/*
 * snn_hip.h — C ABI of libsnnhip.so: the spiking RPN head + spiking RoI (detector) head forward
 * path of aitor-martinez-seras/SNN-Automotive-Object-Detection as hand-written gfx950 (MI355X)
 * HIP kernels.
 *
 * The reference has no FFI of its own (it is pure Python on torch + Norse); the entry points below
 * are what a binding for this path calls instead of the reference's Python loops:
 *
 *   snn_rpn_head_forward   replaces  RPNHeadSNN.forward                    rpn.py:84-121
 *                          (+ the spike-rate variant kept as a string literal, rpn.py:126-200)
 *   snn_det_head_forward   replaces  FastRCNNPredictorSNNFull.forward      faster_rcnn.py:470-516
 *                          (+ the spike-rate variant, faster_rcnn.py:520-618)
 *   snn_pack_*             one-off re-layout of the reference's state_dict tensors
 *                          (shared_conv/conv_cls/conv_bbox, rpn.py:64-75; fc6/fc7/cls_score/bbox_pred,
 *                          faster_rcnn.py:447-467) into the MFMA fragment-major layout the kernels read
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; the caller owns every buffer; the
 *     library allocates nothing persistent and frees nothing of the caller's;
 *   - all work is enqueued on the stream passed in; no internal synchronisation, no default-stream
 *     use, no host<->device copies: every call is hipGraph-capturable;
 *   - return value 0 = success, negative = error (message via snn_last_error(), thread-local);
 *     no exceptions cross the ABI and the library never aborts;
 *   - re-entrant: no global mutable state besides the thread-local error string.  Two kinds of process-wide constants are
 *     cached at first use: the debug / A-B knobs (environment variables SNN_*, read ONCE and frozen; snn_debug_reload_knobs()
 *     re-reads them - test hook, not for concurrent use) and properties of the device (CU count, occupancy of a kernel).
 *
 * Spike tensors never exist as fp32.  A spike train is a set of BIT-PLANES
 *     plane[t][row][w]   (uint32, bit b of word w = spike of channel 32*w+b at time step t)
 * rows are spatial positions (RPN: level-major, then n, y, x) or RoIs (detector).
 */
#ifndef SNN_HIP_H
#define SNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* snn_stream_t;   /* == hipStream_t */

#define SNN_MAX_LEVELS 8
#define SNN_MAX_STEPS 32

/* Neuron constants.  The fp32 products are formed by the caller exactly as Norse forms them
 * (0-dim fp32 tensor arithmetic: dt * tau_mem_inv, -dt * tau_syn_inv) so that the element-wise
 * arithmetic is bit-identical to the reference's (norse lif.py / leaky_integrator.py; reference
 * call sites rpn.py:58,67,101,106,111,115, faster_rcnn.py:444-456,494-510). */
typedef struct snn_params {
    float dt_tau_mem;      /* fl32(dt * tau_mem_inv)   = 0.1f for the reference (dt=0.001, 1/1e-2) */
    float neg_dt_tau_syn;  /* fl32(-dt * tau_syn_inv)  = -0.2f                  (1/5e-3)           */
    float v_leak;          /* 0 */
    float v_reset;         /* 0 */
    float v_th_enc;        /* 0.25  encoder threshold (rpn.py:58, faster_rcnn.py:444) */
    float v_th_lif;        /* 0.1   hidden LIF threshold (rpn.py:67, faster_rcnn.py:449,452) */
    int32_t li_order;      /* 0 = jump-first (upstream li_feed_forward_step), 1 = voltage-first */
    int32_t precision;     /* SNN_PRECISION_*: how the two big contractions run (results are fp32-exact in both) */
} snn_params;

#define SNN_PRECISION_F32 0      /* fp32 matrix cores (v_mfma_f32_32x32x2_f32); 3x3 conv + LIF fused over T */
#define SNN_PRECISION_MXFP6 2    /* fp4 x fp6 block-scaled matrix path, weights as 6 planes of base-32 digits (snn_pack_*_mx):
                                    exact for weights within 2^5 of their block maximum, else rounded at 2^-28 of it;
                                    needs C / D / Hd multiples of 128 (else -4: use SNN_PRECISION_BF16X3) */
#define SNN_PRECISION_BF16X3 1   /* bf16 matrix cores, exact 3-way bf16 split of the fp32 weights; the conv runs
                                    time-batched (currents through HBM) followed by the LIF scan */

#define SNN_PRECISION_F32_STRICT 3 /* SNN_PRECISION_F32 with the LI heads on the fp32 VALU kernel too: no weight is ever split into bf16 planes.
                                    Where layers go whose weights snn_check_bf16x3_split reports as not exactly splittable (packed
                                    weights: the SNN_PRECISION_F32 ones) */

typedef struct snn_rpn_level {
    const float* feat;     /* [N][C][H][W] fp32, NCHW contiguous (what the FPN hands over) */
    int32_t N, H, W;
    int32_t reserved;
} snn_rpn_level;

int snn_version(void);
const char* snn_last_error(void);

/* ---- weight packing (call when the weights change; results are plain device buffers) ---------- */
/* number of floats of a packed GEMM operand with K reduction rows and N output columns */
size_t snn_packed_gemm_elems(int K_chunks32, int N);
/* shared_conv.weight [C_out][C_in][3][3] -> packed, reduction index k = tap*Cp + ci (Cp = C_in
 * rounded up to 32) */
size_t snn_packed_conv3x3_elems(int C_out, int C_in);
int snn_pack_conv3x3_weight(const float* w_oihw, int C_out, int C_in, float* packed, snn_stream_t s);
/* nn.Linear weight [N][K] -> packed */
size_t snn_packed_linear_elems(int N, int K);
int snn_pack_linear_weight(const float* w_nk, int N, int K, float* packed, snn_stream_t s);
/* the two leaky-integrator heads (cls then bbox) [NA][K] and [NB][K] -> transposed [Kp][NOp],
 * NOp = (NA+NB) rounded up to 16, Kp = K rounded up to 32 */
size_t snn_packed_heads_elems(int NA, int NB, int K);
int snn_pack_heads_weight(const float* w_a, int NA, const float* w_b, int NB, int K, float* packed,
                          snn_stream_t s);

/* Exactness check of the bf16x3 split (SNN_PRECISION_BF16X3, and the LI heads of every precision but SNN_PRECISION_F32_STRICT, carry
 * each fp32 weight as hi + mid + lo with three bf16 values).  The split is exact for every finite fp32 weight except magnitudes with
 * bits below 2^-133 and values within 2^119 of FLT_MAX; this call classifies n weights (any layout) and writes three counters:
 *   status3[0]  finite weights whose planes do NOT add up to the fp32 value
 *   status3[1]  non-finite weights (the split of a NaN / infinity is meaningless)
 *   status3[2]  weights with a plane that is a non-zero bf16 subnormal (exact; informational)
 * A binding must not run the bf16x3 kernels on a tensor with status3[0] or status3[1] != 0 (the Python modules fall back to
 * SNN_PRECISION_F32_STRICT with a RuntimeWarning).  status3 is a device buffer, written on `s`. */
int snn_check_bf16x3_split(const float* w, size_t n, uint32_t* status3, snn_stream_t s);

/* ---- RPN head ------------------------------------------------------------------------------- */
/* P = sum over levels of N*H*W.  Outputs are position-major ("NHWC"): out_logits[P][A],
 * out_bbox[P][4A]; level l starts at row sum_{j<l} N_j*H_j*W_j and is [N][H][W][A] inside.
 * Optional (nullable) spike-rate outputs of the rpn.py:126-200 variant:
 *   spike_counts[n_levels][max_N] (uint64)  number of shared-LIF spikes per level and image,
 *   sum_logits[P][A], sum_bbox[P][4A]        sum over the T steps of the LI membranes.          */
size_t snn_rpn_head_workspace_bytes(const snn_rpn_level* levels_host, int n_levels, int C, int A, int T,
                                    int precision);
int snn_rpn_head_forward(const snn_rpn_level* levels_host, int n_levels, int C, int A, int T,
                         const snn_params* p_host,
                         const void* w_shared_packed /* snn_pack_conv3x3_weight[_bf16x3], matching p_host->precision */,
                         const float* w_heads_packed,
                         float* out_logits, float* out_bbox,
                         unsigned long long* spike_counts, float* sum_logits, float* sum_bbox,
                         void* workspace, size_t workspace_bytes, snn_stream_t stream);

/* Same call restricted to a subset of its three stages (they communicate through the workspace);
 * lets a profiler bracket the dominant kernel exactly as the full call launches it.
 * CONTRACT: the workspace contents between stages are OPAQUE (which planes exist raw, which only compressed, and where, depends on
 * T, the precision and the library's knobs: since round 5 the encoder stage writes the period planes e_3 .. compressed only).  A
 * stage-by-stage caller must run the stages in order on the SAME workspace with IDENTICAL arguments (levels, C, A, T, params,
 * weights, spike_counts null or not) and an unchanged environment; a later stage after anything else has used the workspace, or
 * with other arguments, computes on stale bytes (no error is raised: the library cannot tell).  Inspect hidden spike planes through
 * snn_debug_last_rpn_planes (snn_hip_debug.h), not by reading the workspace. */
#define SNN_STAGE_ENCODE 1
#define SNN_STAGE_CONV_LIF 2
#define SNN_STAGE_LI_HEADS 4
#define SNN_STAGE_ALL 7
int snn_rpn_head_forward_stages(const snn_rpn_level* levels_host, int n_levels, int C, int A, int T,
                                const snn_params* p_host,
                                const void* w_shared_packed, const float* w_heads_packed,
                                float* out_logits, float* out_bbox,
                                unsigned long long* spike_counts, float* sum_logits, float* sum_bbox,
                                void* workspace, size_t workspace_bytes, int stage_mask,
                                snn_stream_t stream);

/* ---- detector (RoI) head -------------------------------------------------------------------- */
/* x[R][D] (the flattened [R][C][7][7] RoI features, faster_rcnn.py:473); out_cls[R][K],
 * out_bbox[R][K4].  Optional spike-rate outputs of the faster_rcnn.py:520-618 variant:
 *   spk6_count[R], spk7_count[R] (uint32)  spikes per RoI summed over T and the Hd neurons,
 *   sum_cls[R][K], sum_bbox[R][K4]         sum over T of the LI membranes.                      */
size_t snn_det_head_workspace_bytes(int R, int D, int Hd, int K, int K4, int T, int precision);
int snn_det_head_forward(const float* x, int R, int D, int Hd, int K, int K4, int T,
                         const snn_params* p_host,
                         const void* w6_packed, const void* w7_packed /* snn_pack_linear_weight[_bf16x3] */,
                         const float* w_heads_packed,
                         float* out_cls, float* out_bbox,
                         uint32_t* spk6_count, uint32_t* spk7_count, float* sum_cls, float* sum_bbox,
                         void* workspace, size_t workspace_bytes, snn_stream_t stream);

/* The same with fc6's weights packed in a PERMUTED reduction order (snn_pack_linear_weight_bf16x3_perm(w, N, K, inner)): w6_inner =
 * elements per channel of the flattened input (49 for [C][7][7] RoI features), k' = s * C + c instead of k = c * inner + s; the head
 * then transposes the encoder's planes to match.  Four consecutive k' are four channels at one bin (independent values) instead of
 * four neighbouring bins of one channel (correlated: equal firing periods), which is what lets fc6's sparse period planes run on the
 * structured-sparse matrix-core instruction (csrc/snn_sparse.h).  The contraction is the same sum in another order.  w6_inner = 0:
 * snn_det_head_forward.  bf16x3 precision, D % 32 == 0, C % 32 == 0; otherwise -4. */
int snn_pack_linear_weight_bf16x3_perm(const float* w_nk, int N, int K, int inner, uint16_t* packed, snn_stream_t s);
int snn_det_head_forward_k(const float* x, int R, int D, int Hd, int K, int K4, int T,
                           const snn_params* p_host,
                           const void* w6_packed, int w6_inner, const void* w7_packed,
                           const float* w_heads_packed,
                           float* out_cls, float* out_bbox,
                           uint32_t* spk6_count, uint32_t* spk7_count, float* sum_cls, float* sum_bbox,
                           void* workspace, size_t workspace_bytes, snn_stream_t stream);

/* ---- RoIAlign fused with the detector encoder (the step in front of the head, roi_heads.py:1217) --------
 * MultiScaleRoIAlign(7x7, sampling_ratio 2, aligned=False) + lif_current_encoder in one kernel: the [R,C,7,7]
 * fp32 RoI features are never materialised.  The caller assigns each RoI its FPN level (torchvision's
 * LevelMapper) and image index; rois are x1,y1,x2,y2 in (padded) image coordinates. */
typedef struct snn_roi_level {
    const float* feat;       /* [N][C][H][W] fp32 */
    int32_t H, W;
    float spatial_scale;     /* 1/4, 1/8, 1/16, 1/32 */
    int32_t reserved;
} snn_roi_level;
int snn_roi_align_encode(const snn_roi_level* levels_host, int n_levels /* <= 4 */, int C, const float* rois /*[R][4]*/,
                         const int* roi_batch /*[R]*/, const int* roi_level /*[R]*/, int R, int T,
                         const snn_params* p_host, uint32_t* planes /*[T][R][ceil(C*49/32)]*/, size_t plane_stride_words,
                         float* pooled_dbg /* nullable [R][C*49]: the pooled features, parity tests only */,
                         snn_stream_t stream);
/* snn_det_head_forward with the encoder fed by snn_roi_align_encode (D = C*49; same workspace size) */
int snn_det_head_forward_roialign(const snn_roi_level* levels_host, int n_levels, int C, const float* rois,
                                  const int* roi_batch, const int* roi_level, int R, int Hd, int K, int K4, int T,
                                  const snn_params* p_host, const void* w6_packed, const void* w7_packed,
                                  const float* w_heads_packed, float* out_cls, float* out_bbox,
                                  uint32_t* spk6_count, uint32_t* spk7_count, float* sum_cls, float* sum_bbox,
                                  void* workspace, size_t workspace_bytes, snn_stream_t stream);

int snn_det_head_forward_roialign_k(const snn_roi_level* levels_host, int n_levels, int C, const float* rois,
                                    const int* roi_batch, const int* roi_level, int R, int Hd, int K, int K4, int T,
                                    const snn_params* p_host, const void* w6_packed, int w6_inner, const void* w7_packed,
                                    const float* w_heads_packed, float* out_cls, float* out_bbox,
                                    uint32_t* spk6_count, uint32_t* spk7_count, float* sum_cls, float* sum_bbox,
                                    void* workspace, size_t workspace_bytes, snn_stream_t stream);

/* ---- finished spike-rate tensors of the two spike-rate variants (rpn.py:171-195, faster_rcnn.py:568-618) from the raw side
 * outputs above: rows (rate, "FLOPs") as float32, what the reference hstacks per layer.
 *   snn_rpn_rates: rates[n_levels][3][max_N][2]; [l][0] = shared-LIF spikes / (T*C*H*W) with 9*H*W*C*C, [l][1] / [l][2] = mean
 *   over (A,H,W) / (4A,H,W) of sum_t membrane / T with H*W*C*A*4 / H*W*C*A (the reference's swapped labels, kept).  Two launches.
 *   snn_det_rates: rates[4][R][2] for lif6, lif7, cls_score, bbox_pred (D*Hd, Hd*Hd, Hd*K, Hd*K*4 or Hd*K when only_one_bbox).
 * The spike counts themselves come out of the LIF epilogues of the fused kernels (ballot + popcount + integer atomics). */
size_t snn_rpn_rates_workspace_bytes(int n_levels, int max_N);
int snn_rpn_rates(const snn_rpn_level* levels_host, int n_levels, int C, int A, int T,
                  const unsigned long long* spike_counts, const float* sum_logits, const float* sum_bbox,
                  float* rates, void* workspace, size_t workspace_bytes, snn_stream_t stream);
int snn_det_rates(int R, int D, int Hd, int K, int K4, int T, int only_one_bbox, const uint32_t* spk6_count,
                  const uint32_t* spk7_count, const float* sum_cls, const float* sum_bbox, float* rates, snn_stream_t stream);

/* ---- greedy (batched) NMS for the callers either side of the heads (rpn.py:517, roi_heads.py:1160-1161) ----
 * boxes [n][4] already sorted by decreasing score; category (nullable) restricts suppression to equal values
 * (level for the RPN, class for the detector).  keep_out receives up to max_keep indices into the SORTED order,
 * in that order; *n_keep_out their number (device memory).  n <= 16384. */
size_t snn_nms_workspace_bytes(int n);
int snn_nms_sorted(const float* boxes_sorted, const int* category_sorted, int n, float iou_threshold, int max_keep,
                   int* keep_out, int* n_keep_out, void* workspace, size_t workspace_bytes, snn_stream_t stream);

/* ---- RPN proposal selection (rpn.py:420-499: per-level top-k, box decode, sigmoid, clip, size / score filters,
 * per-level NMS, post_nms_top_n) for the whole batch, no host synchronisation ---------------
 * (multi-block radix select + sort per level, decode, NMS mask and NMS walk with one list per (image, level), rank merge of the
 * kept candidates: 12 small launches)
 * Inputs are the head's own position-major outputs.  The K candidates of an image are in the reference's order (level by
 * level, inside a level by decreasing logit as objectness.topk returns them, equal logits by element index); pre_boxes /
 * pre_prob report them in that order (rpn.py:493-499).  Equal sigmoid values keep that order in the NMS walk, like the
 * reference's stable sort.  Outputs are padded to post_nms_top_n rows, out_counts[N] holds the valid rows. */
typedef struct {
    const float* logits;        /* [N*H*W][A]  objectness logits, rows n*H*W + y*W + x                  */
    const float* deltas;        /* [N*H*W][4A] box regression                                           */
    int32_t H, W;
    float stride_h, stride_w;   /* image size // feature size (AnchorGenerator)                         */
    float base_anchors[16][4];  /* the level's cell anchors (rounded, as AnchorGenerator builds them)   */
} snn_rpn_post_level;
int snn_rpn_proposals_candidates(const snn_rpn_post_level* levels, int n_levels, int A, int pre_nms_top_n);
size_t snn_rpn_proposals_workspace_bytes(int N, int K_candidates);
int snn_rpn_proposals(const snn_rpn_post_level* levels, int n_levels, int N, int A, const float* image_hw_host,
                      int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float score_thresh, float min_size,
                      float* out_boxes, float* out_scores, int* out_counts,
                      float* pre_boxes /* nullable [N][K][4]: decoded un-clipped candidates */,
                      float* pre_prob /* nullable [N][K] */, void* workspace, size_t workspace_bytes,
                      snn_stream_t stream);

/* ---- detection post-processing (roi_heads.py:1075-1176, incl. the reference's background-box report): softmax,
 * per-class box decode, clip, score / size filters, per-class NMS (one list per (image, class)), detections_per_img - five
 * launches for the batch, no host synchronisation.  Rows of image i in out_* [N][out_cap]: out_counts[2i] foreground
 * detections by decreasing score, then out_counts[2i+1] background boxes (RoIs without any class above the score
 * threshold).  all_scores [R][K] / all_boxes [R][K][4] receive the softmax scores and clipped boxes of every class.
 * out_cap >= detections_per_img + max RoIs per image; max RoIs per image <= 10240, K <= 96 and
 * (K-1) * detections_per_img <= 8192, else -4. */
size_t snn_det_postprocess_workspace_bytes(int N, int max_rois_per_image, int K);
int snn_det_postprocess(const float* class_logits, const float* box_regression, const float* proposals,
                        const int* rois_per_image_host, int N, int K, const float* image_hw_host,
                        const float* box_weights_host /* [4]: BoxCoder weights, (10, 10, 5, 5) in the reference */,
                        float score_thresh, float nms_thresh, int detections_per_img, float min_size,
                        float* all_scores, float* all_boxes, float* out_boxes, float* out_scores, int* out_labels,
                        int* out_counts, int out_cap, void* workspace, size_t workspace_bytes, snn_stream_t stream);

/* ---- exchange payload of the data-parallel path: what each rank hands to the ONE all-gather of a batch (the reference's
 * counterpart is the pickled all_gather_object of coco_eval.py:158-177, disabled under NCCL at train.py:874-880).  Per
 * image the max_det RoIs with the highest foreground score (softmax over K, best class >= 1; the selection of
 * roi_heads.py:1103-1110 without NMS) as rows (the 4 regression values of that class, score, label) by decreasing
 * score, ties by RoI index; counts[i] = min(max_det, rois_per_image).  One launch; rois_per_image <= 4096. */
int snn_det_exchange_payload(const float* class_logits /* [N*rois_per_image][K] */,
                             const float* box_regression /* [N*rois_per_image][4K] */, int N, int rois_per_image, int K,
                             int max_det, float* payload /* [N][max_det][6] */, int* counts /* [N] */, snn_stream_t stream);

/* ---- helper outside the spiking path: FrozenBatchNorm2d (+ residual) (+ ReLU) of the stock backbone in one pass -----
 * y[n][c][i] = relu?( (x[n][c][i] * scale[c] + bias[c]) (+ residual[n][c][i]) ), the operations of torchvision's
 * FrozenBatchNorm2d.forward / Bottleneck.forward (called at faster_rcnn.py:693-694 through resnet_fpn_backbone) in the same
 * order with separate roundings: bit-identical to the four torch launches it replaces.  residual nullable; y may be x. */
int snn_affine_act_nchw(const float* x, const float* scale, const float* bias, const float* residual, int N, int C,
                        int HW, int relu, float* y, snn_stream_t stream);

/* ---- stage-level entry points (parity tests drive the layers one by one, teacher-forced) ----- */
/* constant-current LIF encoder -> bit-planes.  NCHW feature map -> planes[T][N*H*W][Cw]          */
int snn_encode_nchw(const float* feat, int N, int C, int H, int W, int T, const snn_params* p_host,
                    uint32_t* planes, size_t plane_stride_words, snn_stream_t stream);
/* row-major x[R][D] -> planes[T][R][Dw]                                                          */
int snn_encode_rows(const float* x, int R, int D, int T, const snn_params* p_host,
                    uint32_t* planes, size_t plane_stride_words, snn_stream_t stream);
/* fused 3x3 spike convolution + LIF over T for ONE level: enc planes [T][N*H*W][Cw] ->
 * spk planes [T][N*H*W][Nw]; counts[N] nullable; dbg_cur (nullable, parity tests only) receives the
 * shared-LIF input currents [T][N*H*W][Nw*32] fp32                                               */
int snn_conv3x3_lif(const uint32_t* enc, size_t enc_stride_words, int N, int C_in, int C_out, int H, int W,
                    int T, const snn_params* p_host, const float* w_packed,
                    uint32_t* spk, size_t spk_stride_words, unsigned long long* counts, float* dbg_cur,
                    snn_stream_t stream);
/* time-batched spike GEMM: A planes as [M][Kw] rows -> cur[M][ldo] fp32 (ldo >= Np)              */
int snn_spike_gemm(const uint32_t* a_rows, int M, int K, int N, const float* w_packed, float* cur,
                   int ldo, snn_stream_t stream);
/* LIF scan over T of currents cur[T][R][ldc] -> spk planes [T][R][Nw]; row_counts[R] nullable     */
int snn_lif_scan(const float* cur, int T, int R, int N, int ldc, const snn_params* p_host,
                 uint32_t* spk, size_t spk_stride_words, uint32_t* row_counts, snn_stream_t stream);
/* leaky-integrator heads on spike planes [T][M][Kw]: out_a[M][NA], out_b[M][NB] = LI membranes at
 * the last step; sum_a / sum_b (nullable) = their sums over the T steps                           */
int snn_li_heads(const uint32_t* spk, size_t spk_stride_words, int T, int M, int K,
                 const float* w_heads_packed, int NA, int NB, const snn_params* p_host,
                 float* out_a, float* out_b, float* sum_a, float* sum_b, snn_stream_t stream);

/* ---- exact bf16x3 contractions (bf16 matrix cores, fp32-exact result) -------------------------------
 * Spikes are exactly {0,1} and every fp32 weight is exactly the sum of three bf16 values, so
 * A x W = A x W_hi + A x W_mid + A x W_lo with every product exact and fp32 accumulation; measured as
 * accurate as the fp32 MFMA chain (tools/bf16x3_numerics.hip).  Packed operand: uint16 [3][K/32][Np][32]. */
size_t snn_packed_bf16x3_elems(int K_chunks32, int N);
size_t snn_packed_conv3x3_bf16x3_elems(int C_out, int C_in);
int snn_pack_conv3x3_weight_bf16x3(const float* w_oihw, int C_out, int C_in, uint16_t* packed, snn_stream_t s);
size_t snn_packed_linear_bf16x3_elems(int N, int K);
int snn_pack_linear_weight_bf16x3(const float* w_nk, int N, int K, uint16_t* packed, snn_stream_t s);
/* Block-scaled fp6 digit planes (precision "mxfp6", csrc/snn_mx.h): the fp32 weights as 6 planes of signed base-32
 * digits in fp6 e2m3 with one E8M0 scale per (32 consecutive k, column).  Exact for every weight within 2^5 of the
 * largest magnitude of its block, else rounded at 2^-28 of that magnitude.  Packed operand: uint32 words. */
size_t snn_packed_linear_mx_words(int N, int K);
size_t snn_packed_conv3x3_mx_words(int C_out, int C_in);
int snn_pack_linear_weight_mx(const float* w_nk, int N, int K, uint32_t* packed, snn_stream_t s);
int snn_pack_conv3x3_weight_mx(const float* w_oihw, int C_out, int C_in, uint32_t* packed, snn_stream_t s);
/* the spike GEMMs on the fp4 x fp6 block-scaled matrix path (k_gemm_mx); K / C_in must be a multiple of 128 (-4 else),
 * same operands and results as the _bf16x3 entry points - except that the two conv entry points read encoder planes
 * with a one-position ZERO HALO around every image (row (n, y, x) of an H x W level at (n (H+2) + y+1) (W+2) + x+1,
 * levels back to back, enc_stride >= sum N (H+2) (W+2) C_in/32 words): 3x3 taps then need no border logic */
int snn_spike_gemm_mx(const uint32_t* a_rows, int M, int K, int N, const uint32_t* w_packed, float* cur, int ldo,
                      snn_stream_t stream);
int snn_spike_gemm_lif_mx(const uint32_t* a_planes, int T, int R, int K, int N, const snn_params* p,
                          const uint32_t* w_packed, uint32_t* spk, size_t spk_stride, snn_stream_t stream);
int snn_conv3x3_lif_mx(const uint32_t* enc, size_t enc_stride, const snn_rpn_level* lv, int n_levels, int C_in, int C_out,
                       int T, const snn_params* p, const uint32_t* w_packed, uint32_t* spk, size_t spk_stride,
                       snn_stream_t stream);
int snn_spike_conv3x3_mx(const uint32_t* enc, size_t enc_stride, const snn_rpn_level* lv, int n_levels, int C_in,
                         int C_out, int T, const uint32_t* w_packed, float* cur, int ldo, snn_stream_t stream);
/* cur[M][ldo] = A_bits[M][K] x W[K][N] */
int snn_spike_gemm_bf16x3(const uint32_t* a_rows, int M, int K, int N, const uint16_t* w_packed, float* cur,
                          int ldo, snn_stream_t stream);
/* Linear layer + LIF over all T steps in one launch (faster_rcnn.py:498-499 / 500-501): a_planes [T][R][K/32] spike
 * bit-planes in, spk [T][R][N/32 words] out (spk_stride = words per time plane).  A row tile of the GEMM holds all T
 * steps of its RoIs, so the input currents and the LIF state never leave the chip.  Returns -4 when T does not fit a
 * row tile (T > 64 or a poor divisor of 256/192/128): use snn_spike_gemm_bf16x3 + snn_lif_scan then. */
int snn_spike_gemm_lif_bf16x3(const uint32_t* a_planes, int T, int R, int K, int N, const snn_params* p,
                              const uint16_t* w_packed, uint32_t* spk, size_t spk_stride, snn_stream_t stream);
/* 3x3 spike convolution + LIF fused over T on the bf16 matrix cores, all levels in one launch:
 * enc planes [T][P][Cw] -> spk planes [T][P][Nw]; the membrane state never leaves the registers */
int snn_conv3x3_lif_bf16x3(const uint32_t* enc, size_t enc_stride_words, const snn_rpn_level* levels_host,
                           int n_levels, int C_in, int C_out, int T, const snn_params* p_host,
                           const uint16_t* w_packed, uint32_t* spk, size_t spk_stride_words, snn_stream_t stream);
/* un-fused time-batched 3x3 spike convolution over all levels: enc planes [T][P][Cw] -> cur[T*P][ldo]
 * (row = t*P + position; position order as in snn_rpn_head_forward); follow with snn_lif_scan */
int snn_spike_conv3x3_bf16x3(const uint32_t* enc, size_t enc_stride_words, const snn_rpn_level* levels_host,
                             int n_levels, int C_in, int C_out, int T, const uint16_t* w_packed, float* cur,
                             int ldo, snn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SNN_HIP_H */
