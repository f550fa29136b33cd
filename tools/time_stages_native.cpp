// time_stages_native.cpp -- the stages of one C2-shaped forward timed straight through the C ABI (include/taco_abi.h), no Python:
// a fresh GPU box spends 1-2 minutes importing torch before bench.py's first kernel runs, this starts in a second -- for the
// iterate-on-a-kernel loop under a GPU-minute budget.  Random-init weights of the reference architecture (every tensor the model asks
// for, by name and shape, through taco_model_num_weights / taco_model_weight_name), synthetic inputs; HIP events on the stream used.
//   hipcc -O2 -I include tools/time_stages_native.cpp -L multi-speaker-tacotron-tensorflow_amd/csrc -ltaco_hip \
//         -Wl,-rpath,'$ORIGIN/../multi-speaker-tacotron-tensorflow_amd/csrc' -o tools/time_stages_native
//   ./tools/time_stages_native [B=32] [T_in=128] [n_steps=128] [reps=10]
// Prints ms per call: whole forward (eager launches), encoder, decoder loop, post-net, and the feed-forward part of the two CBHG stages
// alone (taco_debug_set_skip_scans: recurrent scans skipped, timing only).  Numbers are eager launches; bench.py replays a hipGraph plan.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "taco_abi.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define TK(x) do { int r_ = (x); if (r_ != 0) { printf("taco error %d at line %d: %s\n", r_, __LINE__, taco_last_error()); return 1; } } while (0)

static unsigned g_seed = 20240927u;
static float urand() { g_seed = g_seed * 1664525u + 1013904223u; return (float)((g_seed >> 8) & 0xFFFFFF) / 16777216.f; }
static bool ends_with(const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, T_in = argc > 2 ? atoi(argv[2]) : 128, n = argc > 3 ? atoi(argv[3]) : 128, reps = argc > 4 ? atoi(argv[4]) : 10;
  taco_hparams hp;
  memset(&hp, 0, sizeof hp);
  hp.num_symbols = 80; hp.num_mels = 80; hp.num_freq = 1025; hp.num_speakers = 1; hp.model_type = 0; hp.speaker_embedding_size = 16;
  hp.embedding_size = 256; hp.enc_prenet_n = 2; hp.enc_prenet[0] = 256; hp.enc_prenet[1] = 128;
  hp.enc_bank_size = 16; hp.enc_bank_channels = 128; hp.enc_maxpool = 2; hp.enc_highway_depth = 4; hp.enc_rnn_size = 128;
  hp.enc_proj_n = 2; hp.enc_proj[0] = 128; hp.enc_proj[1] = 128; hp.enc_proj_width = 3;
  hp.attention_type = 2; hp.attention_size = 256; hp.attention_state_size = 256; hp.dec_layer_num = 2; hp.dec_rnn_size = 256;
  hp.dec_prenet_n = 2; hp.dec_prenet[0] = 256; hp.dec_prenet[1] = 128;
  hp.post_bank_size = 8; hp.post_bank_channels = 256; hp.post_maxpool = 2; hp.post_highway_depth = 4; hp.post_rnn_size = 256;
  hp.post_proj_n = 2; hp.post_proj[0] = 256; hp.post_proj[1] = 80; hp.post_proj_width = 3;
  hp.reduction_factor = 4; hp.max_iters = n;
  CK(hipSetDevice(0));
  taco_model* m = nullptr;
  TK(taco_model_create(&hp, 0, &m));
  const int nw = taco_model_num_weights(m);
  size_t nparam = 0;
  for (int i = 0; i < nw; ++i) {
    char name[256]; int64_t shp[4] = {0, 0, 0, 0}; int nd = 0;
    TK(taco_model_weight_name(m, i, name, sizeof name, shp, &nd));
    size_t cnt = 1;
    for (int d = 0; d < nd; ++d) cnt *= (size_t)shp[d];
    nparam += cnt;
    std::vector<float> v(cnt);
    const std::string s(name);
    if (ends_with(s, "moving_variance") || ends_with(s, "gamma")) for (auto& x : v) x = 0.5f + urand();           // randomised BatchNorm
    else if (ends_with(s, "moving_mean") || ends_with(s, "beta")) for (auto& x : v) x = 0.2f * (urand() - 0.5f);
    else if (ends_with(s, "gates/bias")) for (auto& x : v) x = 1.0f;                                                  // TF-sem GRUCell
    else if (ends_with(s, "/T/bias")) for (auto& x : v) x = -1.0f;                                                    // modules.py:119
    else if (ends_with(s, "bias") || nd == 0) for (auto& x : v) x = 0.f;
    else {                                                                                                             // Glorot-uniform kernels / embeddings
      double fan_in = 1;
      for (int d = 0; d + 1 < nd; ++d) fan_in *= (double)shp[d];
      const float lim = nd >= 2 ? (float)std::sqrt(6.0 / (fan_in + (double)shp[nd - 1])) : 0.5f;
      for (auto& x : v) x = (2.f * urand() - 1.f) * (lim > 0.5f ? 0.5f : lim);
    }
    TK(taco_model_set_weight(m, name, v.data(), shp, nd));
  }
  TK(taco_model_finalize(m));
  const int r = hp.reduction_factor, T_mel = n * r;
  std::vector<int32_t> ids((size_t)B * T_in), lens(B, T_in - 1);
  for (auto& x : ids) x = 2 + (int)(urand() * 77.99f);
  for (int b = 0; b < B; ++b) ids[(size_t)b * T_in + T_in - 1] = 1;
  int32_t *d_ids, *d_len, *d_stop;
  float *d_enc, *d_mel, *d_lin, *d_ali;
  CK(hipMalloc(&d_ids, ids.size() * 4)); CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_len, B * 4)); CK(hipMemcpy(d_len, lens.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_stop, 64)); CK(hipMemset(d_stop, 0, 64));
  CK(hipMalloc(&d_enc, (size_t)B * T_in * 2 * hp.enc_rnn_size * 4));
  CK(hipMalloc(&d_mel, (size_t)B * T_mel * hp.num_mels * 4));
  CK(hipMalloc(&d_lin, (size_t)B * T_mel * hp.num_freq * 4));
  CK(hipMalloc(&d_ali, (size_t)B * T_in * n * 4));
  const size_t wsb = std::max(taco_workspace_bytes(m, B, T_in, n), taco_stage_workspace_bytes(m, B, std::max(T_in, T_mel)));
  void* d_ws; CK(hipMalloc(&d_ws, wsb));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("C ABI stage timer: B=%d T_in=%d n_steps=%d (T_mel=%d), %d tensors / %.3f M parameters, workspace %.1f MB, %d timed calls each\n",
         B, T_in, n, T_mel, nw, nparam * 1e-6, wsb / 1048576.0, reps);
  auto timed = [&](const char* what, auto&& fn) -> int {
    for (int i = 0; i < 3; ++i) TK(fn());
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) TK(fn());
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    int err = 0; TK(taco_model_device_errors(m, &err));
    printf("  %-46s %8.3f ms%s\n", what, ms / reps, err ? "   (DEVICE ERROR WORD SET)" : "");
    return 0;
  };
  auto fwd = [&]() { return taco_forward_infer(m, st, d_ids, d_len, nullptr, B, T_in, n, nullptr, d_mel, d_lin, d_ali, d_stop, d_ws, wsb); };
  auto enc = [&]() { return taco_encoder_forward(m, st, d_ids, d_len, nullptr, B, T_in, d_enc, d_ws, wsb); };
  auto dec = [&]() { return taco_decoder_forward(m, st, d_enc, nullptr, B, T_in, n, nullptr, nullptr, d_mel, d_ali, d_stop, nullptr, d_ws, wsb); };
  auto post = [&]() { return taco_postnet_forward(m, st, d_mel, nullptr, B, T_mel, d_lin, nullptr, d_ws, wsb); };
  if (timed("whole forward (eager launches)", fwd)) return 1;
  if (timed("encoder", enc)) return 1;
  if (timed("decoder loop", dec)) return 1;
  if (timed("post-net + linear head", post)) return 1;
  TK(taco_debug_set_skip_scans(m, 1));
  if (timed("encoder, feed-forward part only", enc)) return 1;
  if (timed("post-net + linear head, feed-forward part only", post)) return 1;
  TK(taco_debug_set_skip_scans(m, 0));
  TK(taco_debug_set_bf3(m, 0, 0));
  if (timed("whole forward, exact-fp32 GEMMs", fwd)) return 1;
  int info[16]; TK(taco_debug_decoder_info(m, info));
  printf("  decoder protocol %d (1 = XCD-local), compute units %d\n", info[0], info[14]);
  taco_model_destroy(m);
  return 0;
}
