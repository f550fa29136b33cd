// time_stages_native.cpp -- the stages of one C2-shaped forward timed straight through the C ABI (include/taco_abi.h), no Python:
// a fresh GPU box spends 1-2 minutes importing torch before bench.py's first kernel runs, this starts in a second -- for the
// iterate-on-a-kernel loop under a GPU-minute budget.  Random-init weights of the reference architecture (every tensor the model asks
// for, by name and shape, through taco_model_num_weights / taco_model_weight_name), synthetic inputs; HIP events on the stream used.
//   hipcc -O2 -I include tools/time_stages_native.cpp -L multi-speaker-tacotron-tensorflow_amd/csrc -ltaco_hip \
//         -Wl,-rpath,'$ORIGIN/../multi-speaker-tacotron-tensorflow_amd/csrc' -o tools/time_stages_native
//   ./tools/time_stages_native [B=32] [T_in=128] [n_steps=128] [reps=10]
// Prints ms per call: whole forward (eager launches), encoder, decoder loop, post-net, and the feed-forward part of the two CBHG stages
// alone (taco_debug_set_skip_scans: recurrent scans skipped, timing only).  Numbers are eager launches; bench.py replays a hipGraph plan.
#include "native_model.h"

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, T_in = argc > 2 ? atoi(argv[2]) : 128, n = argc > 3 ? atoi(argv[3]) : 128, reps = argc > 4 ? atoi(argv[4]) : 10;
  taco_hparams hp; taco_model* m = nullptr; int nw = 0; size_t nparam = 0;
  if (native_model(n, hp, m, nw, nparam)) return 1;
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
