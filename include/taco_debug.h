/* taco_debug.h -- test hooks, A/B switches and timing / tracing hooks of libtaco_hip.so.
 *
 * NOT part of the drop-in boundary: a reference-side binding needs include/taco_abi.h only.  What is
 * declared here exists for this repository's own tests (tests/), benchmarks (bench.py companions and stage timing) and tools/;
 * the functions may change between builds without a change of TACO_ABI_VERSION.  The library exports them unconditionally. */
#ifndef TACO_DEBUG_H
#define TACO_DEBUG_H

#include "taco_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test hook: 0 = per-step launches for the sequential loops, 1 (default) = the persistent kernels that fit (post-net scan: k_bigru_duo);
 * 2..7 select earlier scan kernels, 8 = k_bigru_xcd (round 2: one direction per group of 16 CUs), 9 its two-workgroups-per-CU geometry
 * (tests / A-B timing) */
int taco_debug_set_persistent(taco_model* m, int on);

/* test hook: on = 1 (default) runs the feed-forward GEMMs of inference on the bf16 matrix cores with 3-term split
 * operands (fp32-grade accuracy, ~1e-5); 0 = exact-fp32 MFMA everywhere.  tile_n: 0 auto, 1 = 128x64, 2 = 128x128,
 * 3 = 64x256 (2x2 waves), 4 = 64x64, 5 = 64x64 with four wave groups splitting K inside the workgroup, 7 = 64x256 by 1x8 waves,
 * 9 = 64x128 by 1x4 waves, 10 = tile 7 with two wave groups splitting K, 11 = 128x256 by 1x8 waves (auto picks 4 / 5 / 7 / 9 / 10 / 11).
 * on bit 2 (on = 5): the point-wise tail of a CBHG ([dense ->] highway x depth -> BiGRU input projection; modules.py:72-96) runs as
 * one launch per layer instead of ONE launch with the activations resident on the CU (csrc/taco_chain.h, the default with on = 1).
 * on bit 3 (on = 9): conv bank and proj_1 of a CBHG as two launches instead of the fused front (csrc/taco_front.h).
 * on bit 4 (on = 17): with the fused front, proj_1's epilogue (k_front_combine) and proj_2 as launches of their own instead of the fused
 * entry of the point-wise chain (csrc/taco_chain.h). */
int taco_debug_set_bf3(taco_model* m, int on, int tile_n);

/* test hook: > 0 = taco_forward_infer runs the post-net feed-forward stages behind the decoder on a second stream
 * (fork/join by events, a parallel branch in the hipGraph) in chunks of max(on,16) decoder steps; 0 (default) =
 * strictly sequential, which measures faster on MI355X (profiles/README.md) */
/* debug/test: 0 = run decoder prenet layer 1 as its own launch every step (default 1: folded into the previous step's
 * frame-projection launch through composite weights; same function, rounding differs at the 1e-7 level) */
int taco_debug_set_fuse_prenet(taco_model* m, int on);
/* debug/test: 0 = run the concat projection (rnn_wrappers.py:405-415 + OutputProjectionWrapper, tacotron.py:166-170) as its own
 * launch every step (default 1: folded into the gates launch of the first decoder GRU through composite weights Wc . Wg_x; the
 * same launch emits the projection output for the residual connection; rounding differs at the 1e-7 level) */
int taco_debug_set_fuse_concat(taco_model* m, int on);
/* debug/test: attention launch shape.  -1 (default): one workgroup per batch row, or -- few rows, long inputs (B <= 16, T_in >= 256) --
 * two launches with 4 slices per row; 0: always one workgroup per row; n > 1: always n slices per row */
int taco_debug_set_att_split(taco_model* m, int slices);
int taco_debug_set_overlap(taco_model* m, int on);

/* Decoder loop engine (reference: rnn_wrappers.py:218-341,367-415; helpers.py:9-32).  mode 1 (default): the whole loop runs as ONE
 * persistent, weight-stationary launch (csrc/taco_decoder_xcd.h) whenever the configuration fits it -- reference widths (256-wide
 * cells, prenet 256/128, two decoder GRUs), model_type single or deepvoice, no manual alignments, no teacher forcing, the
 * attention memory slice of a member fits its LDS; every other call uses the launch-per-stage loop.  mode 0: always launch per
 * stage.  mode 2: persistent with write-through (placement-independent) exchanges even when the census finds one group per XCD.
 * rows_per_group: 0 = smallest of 1/2/4/8 that covers the batch with 8 groups; a larger value packs the batch onto fewer XCDs. */
int taco_debug_set_decoder_persist(taco_model* m, int mode, int rows_per_group);
/* test hook: set the sticky device error word (as a persistent kernel does when its bounded spin expires) to `value` */
int taco_debug_raise_device_error(taco_model* m, int value);
/* after a forward: out16[0] = exchange protocol the last persistent decoder launch used (0 none ran, 1 XCD-local plain stores,
 * 2 write-through), out16[1..8] = workgroups the census saw per XCD, out16[9] = protocol of the persistent BPTT launch when the
 * last decoder backward (training shadow model) used it, else 0, out16[14] = compute units of the device (the whole-chip
 * persistent kernels are used only when there are 256: an unpartitioned MI355X; a CPX / DPX partition runs the launch-per-stage
 * engine), out16[15] = 1 when the model has a persistent-decoder pack */
int taco_debug_decoder_info(taco_model* m, int* out16);
/* phase timeline of group 0 / member 0 for the first 8 decoder steps (scan: steps 8-15): enable bit 0 = launches enqueued from now
 * on write their stamps (the buffer is allocated on first use and lives as long as the model, because captured plans keep its
 * address; drop plans captured under the other setting); out (nullable) receives [8][16] shader-clock stamps of the decoder, or,
 * with enable bit 1, of the post-net scan (its own third of the buffer), or, with bit 2, of the persistent BPTT of a training
 * shadow model (k_decoder_bwd_xcd, steps 8-15 of its launch) */
int taco_debug_decoder_trace(taco_model* m, int enable, long long* out);

/* timing hook: on = 1 leaves the recurrent scan launches of both CBHGs out of every forward / stage call enqueued from now on
 * (their outputs are then meaningless), so that the feed-forward part of a stage can be timed alone (bench.py roofline.stages) */
int taco_debug_set_skip_scans(taco_model* m, int on);

/* A/B hook: on = 0 switches off the per-device ordering of whole-chip kernels across the streams of this process (taco_plan_whole_chip in
 * taco_abi.h); two persistent forwards on different streams then starve each other until their bounded spins report a device fault */
int taco_debug_set_chip_turns(int on);

/* test hook: force the k_gemm tile configuration (0: 128x64, 1: 64x64, 2: 32x64 split-K, 3: 128x128; -1 auto) */
int taco_debug_force_gemm_config(taco_model* m, int cfg);

/* tuning hook of k_cbhg_front (csrc/taco_front.h: conv bank -> max-pool -> proj_1 as one launch): the second K half of every workgroup
 * starts `delay_clocks` shader clocks late and runs at s_setprio level `prio` (0..3).  Defaults 0 / 1 (tools/time_front.py).
 * taco_debug_set_bf3 bit 3 (on = 9) switches the fused front off (bank and proj_1 as two k_gemm_bf3 launches). */
int taco_debug_set_front(taco_model* m, int delay_clocks, int prio);

/* Test hook: the post-net BiGRU scan alone -- forward with the gate tape, then backward -- on caller data, ragged lengths and
 * initial states included.  persistent = 1: the whole-chip kernels k_bigru_duo<RG, true> + k_bigru_duo_bwd; 0: k_bigru_res + k_bigru_rows_bwd.
 * d_xproj [B*T, 6H] (hoisted input projection, backward direction time-reversed per row), d_lengths [B] / NULL, d_h0 [B, 2H] / NULL,
 * d_dout [B*T, 2H] -> d_out [B*T, 2H], d_gsave / d_dg [B*T, 6H], d_rh [B*T, 2H], d_dh0 [B, 2H] / NULL; scratch >= 1 MB. */
int taco_train_debug_bigru(taco_train* t, void* hip_stream, const float* d_xproj, const int32_t* d_lengths, const float* d_h0,
                           const float* d_dout, int B, int T, int persistent, float* d_out, float* d_gsave, float* d_dg, float* d_rh,
                           float* d_dh0, void* d_scratch, size_t scratch_bytes);

#ifdef __cplusplus
}
#endif
#endif /* TACO_DEBUG_H */
