/* taco_abi.h -- C ABI of libtaco_hip.so, the MI355X (gfx950) Tacotron hot path.
 *
 * The reference (GSByeon/multi-speaker-tacotron-tensorflow) has no FFI/plugin interface: its hot
 * path is the TF1 graph built by Tacotron.initialize() (models/tacotron.py:21-271) and executed by
 * sess.run from Synthesizer.synthesize (synthesizer.py:166-167) and train() (train.py:217-219).
 * This header is the boundary a maintainer would bind instead of sess.run (see INTEGRATION.md for
 * the ctypes stub).  Each entry point cites the reference lines it replaces.
 *
 * Conventions: extern "C"; plain pointers and sizes; fp32 row-major [batch, time, channels];
 * all `d_*` pointers are DEVICE pointers owned by the caller; the library owns only the weight
 * pack inside taco_model.  Every call is asynchronous on `hip_stream` (a hipStream_t passed as
 * void*), performs no allocation and no host<->device synchronisation, and is hipGraph-capturable.
 * Return value 0 = OK; negative = error class below, text in taco_last_error().
 */
#ifndef TACO_ABI_H
#define TACO_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TACO_ABI_VERSION 1

#define TACO_OK 0
#define TACO_ERR_ARG (-1)         /* null pointer, bad size, unknown name */
#define TACO_ERR_SHAPE (-2)       /* weight/activation shape mismatch (tacotron.py:192-194) */
#define TACO_ERR_UNSUPPORTED (-3) /* unknown model_type / attention_type (tacotron.py:88,152) */
#define TACO_ERR_HIP (-4)         /* HIP runtime error */
#define TACO_ERR_STATE (-5)       /* model not finalized / missing weights / workspace too small */

typedef struct taco_model taco_model; /* opaque: hparams + device weight pack */
typedef struct taco_plan taco_plan;   /* opaque: an instantiated hipGraph of one forward */

/* Mirrors the model keys of hparams.py:33-69 and max_iters (hparams.py:141). */
typedef struct {
  int32_t num_symbols;             /* len(symbols), text/symbols.py:13 (80) */
  int32_t num_mels, num_freq;      /* hparams.py:16-17 */
  int32_t num_speakers;            /* train.py:301 / synthesizer.py:28 */
  int32_t model_type;              /* 0 single, 1 simple, 2 deepvoice      tacotron.py:51-94 */
  int32_t speaker_embedding_size;  /* hparams.py:35 */
  int32_t embedding_size;          /* hparams.py:37 */
  int32_t enc_prenet_n, enc_prenet[4];                                     /* hparams.py:40 */
  int32_t enc_bank_size, enc_bank_channels, enc_maxpool, enc_highway_depth, enc_rnn_size;
  int32_t enc_proj_n, enc_proj[4], enc_proj_width;                         /* hparams.py:41-47 */
  int32_t attention_type;          /* 0 bah, 1 bah_norm, 2 bah_mon         tacotron.py:132-146 */
  int32_t attention_size, attention_state_size;                            /* hparams.py:51-52 */
  int32_t dec_layer_num, dec_rnn_size;                                     /* hparams.py:55-56 */
  int32_t dec_prenet_n, dec_prenet[4];                                     /* hparams.py:59 */
  int32_t post_bank_size, post_bank_channels, post_maxpool, post_highway_depth, post_rnn_size;
  int32_t post_proj_n, post_proj[4], post_proj_width;                      /* hparams.py:60-66 */
  int32_t reduction_factor;        /* hparams.py:68 */
  int32_t max_iters;               /* hparams.py:141 */
} taco_hparams;

int taco_abi_version(void);
const char* taco_last_error(void); /* thread-local; valid until the next call on this thread */

/* ---- model life cycle: replaces create_model + tf.train.Saver.restore (models/__init__.py:6-7,
 *      synthesizer.py:46-67).  Weight names are the canonical names listed in DESIGN.md
 *      (one per TF variable of SURVEY App. B); tensors are HOST fp32 arrays in TF layout
 *      (dense kernel [in,out]; conv1d kernel [k,in,out]; GRU gates [in+n,2n] r|u, candidate [in+n,n]). */
int taco_model_create(const taco_hparams* hp, int device, taco_model** out);
int taco_model_set_weight(taco_model* m, const char* name, const float* host, const int64_t* shape, int ndim);
int taco_model_num_weights(const taco_model* m);                 /* how many tensors the hparams require */
int taco_model_weight_name(const taco_model* m, int i, char* buf, int buflen, int64_t* shape4, int* ndim);
int taco_model_finalize(taco_model* m); /* repack: MFMA fragment layouts, BN -> scale/shift, GRU [x;h] split */
void taco_model_destroy(taco_model* m);

/* ---- whole forward: replaces sess.run([linear_outputs, alignments]) at synthesizer.py:166-167
 *      over the graph of models/tacotron.py:29-239 (inference, is_training False). */
size_t taco_workspace_bytes(const taco_model* m, int B, int T_in, int n_steps);

int taco_forward_infer(taco_model* m, void* hip_stream,
                       const int32_t* d_inputs,          /* [B,T_in] ids, PAD=0 EOS=1   tacotron.py:38 */
                       const int32_t* d_input_lengths,   /* [B]                          synthesizer.py:120 */
                       const int32_t* d_speaker_id,      /* [B] or NULL (zeros)          synthesizer.py:43-44 */
                       int B, int T_in, int n_steps,     /* n_steps = max_iters          tacotron.py:210 */
                       const float* d_manual_alignments, /* NULL, or [B,n_steps,T_in]    rnn_wrappers.py:313-317 */
                       float* d_mel,                     /* [B,n_steps*r,num_mels]       tacotron.py:213-214 */
                       float* d_linear,                  /* [B,n_steps*r,num_freq]       tacotron.py:235 */
                       float* d_alignments,              /* [B,T_in,n_steps]             tacotron.py:238-239 */
                       int32_t* d_stop_step,             /* [1]: decoder steps the reference's stop rule
                                                            (helpers.py:29 + dynamic_decode) would have run;
                                                            == n_steps unless every row emitted an all-zero step.
                                                            NEGATIVE = -(device error word): a persistent kernel of
                                                            this forward (or of an earlier one whose error was never
                                                            acknowledged with taco_model_device_errors) gave up and
                                                            the outputs are invalid */
                       void* d_workspace, size_t workspace_bytes);

/* The same forward captured once into a hipGraph (all pointers baked in) and replayed. */
int taco_plan_create(taco_model* m, const int32_t* d_inputs, const int32_t* d_input_lengths,
                     const int32_t* d_speaker_id, int B, int T_in, int n_steps,
                     const float* d_manual_alignments, float* d_mel, float* d_linear,
                     float* d_alignments, int32_t* d_stop_step, void* d_workspace, size_t workspace_bytes,
                     taco_plan** out);
int taco_plan_launch(taco_plan* p, void* hip_stream);
int taco_plan_num_nodes(const taco_plan* p);
/* 1 if the plan contains a whole-chip persistent kernel (the decoder loop or a post-net scan on the persistent engine).  Such kernels need
 * every compute unit of the device at once; the library puts them -- eager launches and plan replays alike, from any stream or thread of
 * the process -- in a total order per device (a launch that follows one on another stream waits for an event recorded on that stream), so that
 * two of them never wait for each other's compute units.
 * Everything else of a forward still overlaps across streams.  Other processes on the same device are outside its reach. */
int taco_plan_whole_chip(const taco_plan* p);
void taco_plan_destroy(taco_plan* p);

/* ---- stage-level entry points (parity tests, profiling) ---- */
/* embedding -> prenet -> encoder CBHG (tacotron.py:34-112).  d_encoder_out [B,T_in,2*enc_rnn_size]. */
int taco_encoder_forward(taco_model* m, void* hip_stream, const int32_t* d_inputs,
                         const int32_t* d_input_lengths, const int32_t* d_speaker_id, int B, int T_in,
                         float* d_encoder_out, void* d_workspace, size_t workspace_bytes);
/* attention memory + decoder loop (tacotron.py:120-214; rnn_wrappers.py:218-341,367-415; helpers.py).
 * d_teacher_frames NULL, or [B,n_steps,num_mels]: frame fed at step t>=1 is teacher[:,t-1]
 * (TacoTrainingHelper rule, helpers.py:44,66).  d_dbg_states NULL, or
 * [n_steps,B,attention_state_size + 2*enc_rnn_size + dec_layer_num*dec_rnn_size]: (h_att, ctx, h_1..h_L) after each step. */
int taco_decoder_forward(taco_model* m, void* hip_stream, const float* d_encoder_out,
                         const int32_t* d_speaker_id, int B, int T_in, int n_steps,
                         const float* d_manual_alignments, const float* d_teacher_frames,
                         float* d_mel, float* d_alignments, int32_t* d_stop_step, float* d_dbg_states,
                         void* d_workspace, size_t workspace_bytes);
/* post-net CBHG + linear head (tacotron.py:219-235).  d_mel [B,T_mel,num_mels] -> d_linear [B,T_mel,num_freq];
 * d_post_out optional [B,T_mel,2*post_rnn_size]; d_speaker_id is read only by model_type 'simple' (:226-233). */
int taco_postnet_forward(taco_model* m, void* hip_stream, const float* d_mel, const int32_t* d_speaker_id, int B, int T_mel,
                         float* d_linear, float* d_post_out, void* d_workspace, size_t workspace_bytes);
size_t taco_stage_workspace_bytes(const taco_model* m, int B, int T);

/* ---- op-level entry points, addressed by layer name inside the model ---- */
/* modules.py:123-131 conv1d -> act -> batch_norm (inference).  act: 0 none, 1 relu.
 * maxpool_width > 1 applies max_pooling1d(width, stride 1, 'same') (modules.py:47-51) to x first. */
int taco_conv1d_bn_f32(taco_model* m, void* hip_stream, const char* layer, const float* d_x, int B, int T,
                       int act, int maxpool_width, float* d_out);
/* tf.layers.dense on [rows, in] (A.1).  act: 0 none, 1 relu, 2 sigmoid, 3 tanh. */
int taco_dense_f32(taco_model* m, void* hip_stream, const char* layer, const float* d_x, int rows, int act,
                   float* d_out);
/* modules.py:105-120 on [rows, D]. */
int taco_highway_f32(taco_model* m, void* hip_stream, const char* layer, const float* d_x, int rows,
                     float* d_out);
/* modules.py:82-96 BiGRU; scope "encoder_cbhg" or "post_cbhg".  d_lengths / d_init_state ([B,2H]) nullable. */
int taco_bigru_f32(taco_model* m, void* hip_stream, const char* scope, const float* d_x,
                   const int32_t* d_lengths, const float* d_init_state, int B, int T, float* d_out,
                   void* d_workspace, size_t workspace_bytes);
/* one attention evaluation (rnn_wrappers.py:304-341 + TF-sem score/normaliser):
 * query = d_cell_output . W_q; alignments from keys/prev; context = alignments . values. */
int taco_attention_step_f32(taco_model* m, void* hip_stream, const float* d_cell_output, const float* d_keys,
                            const float* d_values, const float* d_prev_alignments, int B, int T_in,
                            float* d_alignments, float* d_context, void* d_workspace, size_t workspace_bytes);

/* one tf.contrib.rnn.GRUCell step (A.6) of a decoder GRU by name ("decoder/attention_gru",
 * "decoder/gru_1", ...; tacotron.py:127-130,171-172).  d_h [R,H] is updated in place; d_out_res
 * (nullable) = h' + x (ResidualWrapper).  Workspace: 3*R*H floats. */
int taco_gru_cell_f32(taco_model* m, void* hip_stream, const char* name, const float* d_x, float* d_h, int R,
                      float* d_out_res, void* d_workspace, size_t workspace_bytes);

/* attention-based end-of-speech trimming (synthesizer.py:242-262, attention_trim && end_of_sentence): per batch row the number of
 * spectrogram frames to keep, reduction_factor * j + 3, from the argmax walk over d_alignments [B, T_in, n_steps];
 * d_seq_len[b] = len(sequence) of the row (tokens incl. EOS and padding, as the reference passes it).  d_spec_end [B]. */
int taco_attention_trim(void* hip_stream, const float* d_alignments, const int32_t* d_seq_len, int B, int T_in, int n_steps,
                        int reduction_factor, int32_t* d_spec_end);
/* The stop rule (helpers.py:29 + dynamic_decode: the loop ends after the first step at which every row has emitted an all-zero
 * step) evaluated on a finished mel buffer d_mel [B, n_steps, width = r*num_mels] per group of rows_per_group consecutive rows:
 * d_stop[B / rows_per_group].  For requests that were served together through one plan (PlanPool coalesce > 1), whose plan-wide
 * stop step covers all of them; with rows_per_group = B it equals the stop step the forward itself reports. */
int taco_stop_steps(void* hip_stream, const float* d_mel, int B, int n_steps, int width, int rows_per_group, int32_t* d_stop);

/* ---- spectrogram -> waveform (audio/__init__.py:54-56,76-96,118-122,149-165; synthesizer.py:264) ---- */
typedef struct {
  int32_t num_freq, sample_rate, griffin_lim_iters;
  float frame_length_ms, frame_shift_ms, preemphasis, min_level_db, ref_level_db, power;
} taco_audio_hparams;                       /* hparams.py:16-23,144-145 */
typedef struct taco_gl taco_gl;
int taco_gl_create(const taco_audio_hparams* hp, int device, taco_gl** out);
void taco_gl_destroy(taco_gl* g);
int taco_gl_num_samples(const taco_gl* g, int T);            /* hop_length * (T - 1), librosa istft with center=True */
size_t taco_gl_workspace_bytes(const taco_gl* g, int B, int T);
/* inv_spectrogram of B utterances: d_spec [B, T, num_freq] in the model's linear_outputs layout (the reference transposes to
 * [num_freq, T] first).  d_init_uniform [B, T, num_freq] in [0,1) plays np.random.rand of _griffin_lim (NULL: counter-based
 * hash of `seed`).  iters < 0: griffin_lim_iters.  d_wav [B, taco_gl_num_samples(T)].  Needs hop*(T-1) > n_fft/2. */
int taco_gl_inv_spectrogram(taco_gl* g, void* hip_stream, const float* d_spec, const float* d_init_uniform,
                            unsigned long long seed, int B, int T, int iters, float* d_wav, void* d_workspace, size_t workspace_bytes);

/* ---- training-side entry points on flat buffers (loss, schedule, clip + Adam); forward/backward: taco_train_* below ---- */
/* add_loss (tacotron.py:274-302).  d_mel_* [B,T,num_mels], d_lin_* [B,T,num_freq], d_loss_coeff [B] (nullable = 1).
 * d_losses[4] = loss, mel_loss, linear_loss, loss_without_coeff.  Workspace >= 64 KiB. */
int taco_loss_f32(void* hip_stream, const float* d_mel_out, const float* d_mel_tgt, const float* d_lin_out,
                  const float* d_lin_tgt, const float* d_loss_coeff, int B, int T, int num_mels, int num_freq,
                  int prioritize_loss, int sample_rate, float* d_losses, void* d_workspace, size_t workspace_bytes);
/* learning-rate schedule of add_optimizer (tacotron.py:313-325); global_step = completed updates. */
float taco_learning_rate(long long global_step, float initial_learning_rate, int decay_learning_rate_mode,
                         int is_randomly_initialized);
/* clip_by_global_norm + tf.train.AdamOptimizer update (tacotron.py:327-336, TF form of Adam) on flat fp32 buffers.
 * Workspace >= 8 KiB.  d_gnorm_out nullable.  A non-finite global norm (NaN / inf: a step poisoned after a device fault, or a genuine
 * overflow) leaves parameters and moments untouched -- where tf.clip_by_global_norm would turn every parameter into NaN. */
int taco_adam_step_f32(void* hip_stream, float* d_params, const float* d_grads, float* d_m, float* d_v, size_t n,
                       long long global_step, float learning_rate, float beta1, float beta2, float epsilon, float clip_norm,
                       float* d_gnorm_out, void* d_workspace, size_t workspace_bytes);

/* ---- training path (config C4): teacher-forced forward with batch-statistics BatchNorm
 * (tacotron.py:26,199-202; modules.py:131; helpers.py:35-67), add_loss and its full backward pass (tf.gradients of
 * tacotron.py:274-336).  All parameters live in ONE flat fp32 device buffer owned by the caller (layout: the tensors of
 * taco_model_weight_name() in order, TF layout, no padding) and all gradients in a second buffer of the same layout --
 * the single all-reduce bucket of the data-parallel step.  A taco_train owns only index maps and weight packs that it
 * regenerates from the flat parameters (taco_train_refresh) after every optimizer step.
 * modules.py:24 calls tf.layers.dropout without training=True, so the reference applies no prenet dropout even when
 * training; neither does this path.  Supported: single-speaker and multi-speaker ('simple', 'deepvoice') models, all three attention types. ---- */
typedef struct taco_train taco_train;
int taco_train_create(const taco_hparams* hp, int device, taco_train** out);
void taco_train_destroy(taco_train* t);
/* the model handle whose taco_model_num_weights / taco_model_weight_name give the flat parameter order and shapes */
taco_model* taco_train_model(taco_train* t);
/* Deterministic reductions: on = 1 makes every sum over rows that normally leaves its workgroup through fp32 atomics (weight
 * gradients, bias and BatchNorm sums, embedding gradients, d attention_v) a two-stage sum in a fixed order, so that a step is
 * run-to-run reproducible TO THE BIT, as the reference's single-device step is (train.py:215-219).  Costs 192 MB of workspace
 * (taco_train_workspace_bytes reflects it: query it again) and 6.6 % of the step (15.85 instead of 14.87 ms at the C4 shard, round 4: the small
 * problems stay in their group launches, their per-slice partials are added up by two more group launches).  Default 1 since round 4
 * (rounds 1-3: 0); on = 0 selects the atomics. */
int taco_train_set_deterministic(taco_train* t, int on);
/* Weight gradients: on = 0 (default) computes dW = X^T . dY on the bf16 matrix cores with operands split three ways (24 bits) and
 * six products per tile, fp32 accumulation (k_wgrad_bf3: fp32-grade, ~2^-24 per product); on = 1 keeps them on the
 * exact-fp32 MFMA (k_wgrad, round 1).  Process-wide A/B and test hook. */
int taco_train_set_exact_wgrad(taco_train* t, int on);
/* Split-bf16 weight gradients from PRE-SPLIT operands (csrc/taco_wgrad_planes.h): mode 1 (default) converts the operands of the large
 * problems -- a whole conv bank, proj_1, the linear head -- once into bf16 planes and multiplies them with a kernel that converts
 * nothing; mode 2 sends every eligible problem that way (test hook); mode 0 keeps all of them on k_wgrad_bf3.  Same six products, same
 * fixed summation order per slice.  Changing 0 <-> non-zero changes taco_train_workspace_bytes. */
int taco_train_set_wgrad_planes(taco_train* t, int mode);
/* How many weight gradients of the last taco_train_forward_backward were computed from pre-split planes (a conv bank counts once). */
int taco_train_planes_problems(const taco_train* t);
/* Feed-forward GEMMs of the training step (both CBHGs' conv banks / projections / highways / GRU input projections, encoder prenet,
 * linear head) and their data gradients.  k_gemm = exact-fp32 MFMA; k_gemm_bf3 = the inference kernels: bf16 matrix cores, both
 * operands split in two, three products per tile (~2^-17 per product, fp32 accumulation), weight planes re-split on the device from
 * the live parameters by taco_train_refresh (k_bf3_gather).
 *   on = 3 (the default of rounds 2-3): forward on k_gemm, data gradients on k_gemm_bf3.  The forward -- and with it every ReLU / max-pool decision and
 *          every BatchNorm statistic -- is exact; a data gradient is linear in dY, and the split costs 4e-6 of the gradient norm
 *          against the all-exact step (measured; tensor by tensor <= 4e-5 of the tensor's scale).  9 % off the C4-shard step.
 *   on = 1: everything on k_gemm (rounds 1-2).
 *   on = 0: everything on k_gemm_bf3: 16 % off the step; the forward's ~1e-5 relative product error is amplified by the BatchNorm
 *          backward's cancellations to ~1e-3 of the gradient norm and resolves near-ties differently.
 *   on = 2: forward k_gemm_bf3, data gradients k_gemm (A/B hook).
 *   on = 4 (default since round 4): forward on the SIX-product instantiation of k_gemm_bf3 (every operand split three ways, hi + lo + l3 = 24 mantissa bits;
 *          l3*hi, hi*l3, lo*lo, lo*hi, hi*lo, hi*hi accumulated in fp32: 2^-24 per product, fp32-grade, as k_wgrad_bf3), data
 *          gradients as in mode 3: the forward leaves the fp32-input MFMA (1/16 of the bf16 rate) without touching its decisions.
 * Call taco_train_refresh after switching (the planes are generated only while an engine that needs them is selected). */
int taco_train_set_exact_gemm(taco_train* t, int on);
/* Back-propagation through the decoder loop (tf.gradients of rnn_wrappers.py:218-341 under train.py:215-219): persistent = 1 (default)
 * runs all T_out/r steps as ONE whole-chip launch (k_decoder_bwd_xcd, csrc/taco_decoder_bwd_xcd.h) when the forward ran on the persistent
 * decoder (reference widths, every model type, <= 64 rows, a whole MI355X); 0 keeps the chain of per-stage launches. */
int taco_train_set_bptt_engine(taco_train* t, int persistent);
size_t taco_train_num_params(const taco_train* t);
int taco_train_param_offset(const taco_train* t, const char* name, size_t* offset);
/* regenerate every weight pack from the flat parameter buffer (call after loading parameters and after every update) */
int taco_train_refresh(taco_train* t, void* hip_stream, const float* d_params);
size_t taco_train_workspace_bytes(const taco_train* t, int B, int T_in, int T_out);
/* Synchronised BatchNorm for the data-parallel step: with a callback set, every BatchNorm layer's batch
 * statistics (forward: sum, centred sum of squares; backward: sum dy, sum dy*xhat) are handed to `fn` as a device vector that
 * the host must replace, in place and ordered on the stream of the running taco_train_forward_backward call, by its sum over
 * all `world_size` ranks (RCCL all-reduce).  Mean, variance, moving averages and the input gradient are then those of the
 * global batch, i.e. of the reference's single-device step over the whole batch (modules.py:131, train.py:145-166); the
 * gamma/beta gradients stay per-rank sums and are averaged by the flat gradient all-reduce like every other parameter.
 * fn == NULL (default): statistics of this rank's rows only.  40 calls per step at the reference architecture (12 forward,
 * 28 backward), 80 to 4096 floats each.  Not capturable into a hipGraph. */
typedef void (*taco_sync_sum_fn)(void* user, float* d_vec, int n);
int taco_train_set_sync_bn(taco_train* t, taco_sync_sum_fn fn, void* user, int world_size);
/* One training forward (+ backward when d_grads != NULL).  d_params is in/out: a forward+backward pass updates the BatchNorm
 * moving averages in place (UPDATE_OPS run only as a dependency of `optimize`, tacotron.py:334); a forward-only pass
 * (d_grads == NULL: loss fetches, the test model) leaves them untouched, as the reference does.  d_mel_targets [B,T_out,num_mels], d_linear_targets [B,T_out,num_freq],
 * T_out a multiple of r, T_out/r <= max_iters (helpers.py:44-48).  d_losses[4] = loss, mel_loss, linear_loss,
 * loss_without_coeff (nullable).  d_mel_out / d_linear_out / d_alignments ([B,T_in,T_out/r]) nullable.
 * d_grads (flat, overwritten) = d loss / d parameter; moving statistics get zero.
 * rnn_decoder_test_mode bit 0 (helpers.py:63-64, the test model of train.py:158-166): the decoder is fed its own previous
 * output instead of the target frame; forward/loss only (d_grads must be NULL).  Bit 1: freeze the moving averages even though
 * d_grads is given (a warm-up pass whose update is discarded, e.g. before capturing the step into a graph).
 * Device faults: if a persistent whole-chip kernel of the step gave up (its bounded spin expired; taco_model_device_errors on
 * taco_train_model(t) reports and clears the sticky word), d_losses[0..3] and d_grads[0] are set to NaN on the stream: the fault is
 * visible in the step's own outputs, survives the data-parallel all-reduce, and makes taco_adam_step_f32 skip the update (the
 * BatchNorm moving averages, which the pass writes itself, may have absorbed that pass's statistics with weight 0.01). */
int taco_train_forward_backward(taco_train* t, void* hip_stream, float* d_params, float* d_grads, const int32_t* d_inputs,
                                const int32_t* d_input_lengths, const int32_t* d_speaker_id /* nullable: single speaker */,
                                const float* d_mel_targets, const float* d_linear_targets,
                                const float* d_loss_coeff, int B, int T_in, int T_out, int prioritize_loss, int sample_rate,
                                float* d_losses, float* d_mel_out, float* d_linear_out, float* d_alignments,
                                int rnn_decoder_test_mode, void* d_workspace, size_t workspace_bytes);

/* Persistent (multi-workgroup, in-kernel synchronised) kernels bound every spin; if one ever expires it sets a
 * device-side error word and all workgroups leave.  This call synchronises, returns the word in *out and clears it;
 * non-zero means the outputs of the affected forward are invalid. */
int taco_model_device_errors(taco_model* m, int* out);
/* Which engine a forward of this shape WOULD run on -- and, when it is not the persistent whole-chip one, why not (widths, rows, LDS,
 * compute units of the device, debug switches).  Nothing is launched.  `out` receives a NUL-terminated line (out_len >= 64; truncated
 * if shorter than the text).  The run-time facts (exchange protocol the census chose) are in taco_debug_decoder_info afterwards. */
int taco_model_engine_plan(taco_model* m, int B, int T_in, int T_mel, int manual, char* out, int out_len);
/* Batch-position bit-invariance for serving setups that need it (a request's outputs must not depend on which rows it was batched or
 * sharded with).  on = 1: every row tile of the fused point-wise kernel walks a layer's contraction from step 0, so a row's fp32
 * accumulation order is the same wherever the row lands: outputs are bitwise equal under batch permutation / re-sharding at any
 * size.  on = 0 (default): the workgroups of an XCD start their K loops at different steps (faster weight stream out of the L2);
 * results are reproducible run to run and equal under permutation to fp32 rounding (bitwise only while a layer has fewer than 8
 * row tiles of 64 rows).  Takes effect for calls and plans made afterwards. */
int taco_model_set_batch_invariant(taco_model* m, int on);

#ifdef __cplusplus
}
#endif
#endif /* TACO_ABI_H */
