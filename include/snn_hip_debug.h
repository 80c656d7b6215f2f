/*
 * snn_hip_debug.h — introspection and test plumbing of libsnnhip.so (same shared object as snn_hip.h).
 *
 * NOT part of the drop-in boundary: a binding of the reference's spiking heads needs snn_hip.h only.  These entry
 * points exist for this repository's parity tests, bench.py's side legs and the tools under tools/: they re-read the SNN_* debug
 * knobs, expose the encoder's threshold table, report which launch plan a configuration takes, and tell the tests where the calling
 * thread's last head forward left its hidden spike planes inside the CALLER's workspace (so that every output row that is off
 * tolerance can be attributed to a flipped spike of the launch that actually ran).
 */
#ifndef SNN_HIP_DEBUG_H
#define SNN_HIP_DEBUG_H
#include "snn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

void snn_debug_reload_knobs(void);     /* re-read the SNN_* debug knobs from the environment (tests only) */
/* Introspection (tests): the threshold table of the period-plane encoder for these parameters - a neuron's first spike is at or before
 * step t iff its input is >= th[t] (csrc/snn_common.h: THRESHOLD FORM).  th[0..31] out; returns 1 if the table verified against the
 * recurrence on the host (the encoders then use it), 0 if not (they keep the recurrence), negative on bad arguments. */
int snn_debug_encoder_thresholds(const snn_params* p, float* th32);
/* Introspection (bench.py's t_sweep leg, tests): the launch plan of a T-in-tile layer of the bf16x3 family - the one the launcher takes
 * for these shapes with the default neuron parameters and the current knobs (host-only call).
 *   conv        1: the RPN's shared 3x3 conv (units = positions of the pyramid, k_in = C_in, n_cols = C_out); 0: a linear layer of the
 *               detector head (units = RoIs, k_in = its inputs, n_cols = its outputs)
 *   num_steps   LIF steps T of the head
 *   spike_rates (conv = 0 only) non-zero: the head runs in spike-rate mode, whose fc6 window is one step longer (the rate counts
 *               lif6's spikes of every step, faster_rcnn.py:556)
 *   layer       (conv = 0 only) 6 or 7: fc6's or fc7's window of time steps (any other value reads as 6)
 * out[0..11] = {M-tile slots per (row-)wave, short row-waves (dense tile) / 1 for the FAT shape of four waves (sparse plan), logical tile rows, positions or RoIs per tile, time steps whose currents
 * are formed (dead time steps removed), work-groups of the launch, column blocks, waves along N,
 * 1 if the structured-sparse launch runs (csrc/snn_sparse.h) / 0 for the dense k_gemm_bf16x3 tile, period planes on the dense
 * instruction, period planes on the structured-sparse instruction, M-tile slots in use per work-group (sparse plan; else 0)}.
 * Returns 0, or -4 if no tile holds that many steps (the launch then takes the un-fused path). */
int snn_debug_tile_shape(int conv, long long units, int k_in, int n_cols, int num_steps, int spike_rates, int layer, int32_t* out12);
/* Introspection (tests, bench.py): 1 if the calling thread's last bf16x3 RPN conv + LIF enqueued the structured-sparse launch pair
 * (csrc/snn_sparse.h: planes e_3 .. on v_smfmac, no dense launch), 0 if it took the dense launch (SNN_SPARSE=0, conv T outside 5 .. 16, channel counts that are not multiples of 64, ...). */
int snn_debug_last_conv_path(void);
int snn_debug_last_fc6_path(void);      /* the same for the detector head's fc6 + LIF */
/* Introspection (parity tests): where the calling thread's last detector-head forward left its hidden spike planes in the caller's
 * workspace: out3[0] / out3[1] = byte offsets of lif6's / lif7's planes ([T][R][Hd/32] words; lif6's are word-major [T][Hd/32][R] if
 * out3[2] == 1).  The tests attribute every RoI that is off tolerance to a flipped spike of the launch that actually ran. */
void snn_debug_last_det_planes(unsigned long long* out3);
/* The same for the RPN head: out3[0] = byte offset of the shared LIF's spike planes ([T][P][C/32] words over all levels; in blocks of
 * four words [T][C/128][P][4] if out3[1] == 1), out3[2] = P. */
void snn_debug_last_rpn_planes(unsigned long long* out3);

/* Measurement (bench.py `held_clock`): enqueue a ONE-wave probe on `stream` that sleeps / polls the constant 100-MHz counter for
 * `ticks_100mhz` ticks (<= 1 s) and then writes {shader-clock cycles elapsed, ticks elapsed} to out2_dev[0..1] (device memory, caller-owned):
 * clock held over that window = 0.1 GHz x cycles / ticks.  Run it on a side stream BESIDE the launches whose clock is asked for. */
int snn_debug_clock_probe(unsigned long long* out2_dev, unsigned int ticks_100mhz, snn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SNN_HIP_DEBUG_H */
