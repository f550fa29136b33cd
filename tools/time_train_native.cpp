// time_train_native.cpp -- one training step at the C4 shard (B = 32, T_in = 128, T_out = 512) timed straight through the C ABI
// (taco_train_forward_backward -> taco_adam_step_f32 -> taco_train_refresh, train.py:215-219), no Python: tools/bench_train.py for the
// iterate-on-a-kernel loop.  Random-init parameters in the flat buffer, targets ~U[0,1], HIP events on the stream used.
//   hipcc -O2 -I include tools/time_train_native.cpp -L multi-speaker-tacotron-tensorflow_amd/csrc -ltaco_hip \
//         -Wl,-rpath,'$ORIGIN/../multi-speaker-tacotron-tensorflow_amd/csrc' -o tools/time_train_native
//   ./tools/time_train_native [B=32] [T_in=128] [T_out=512] [reps=8] [exact_gemm=4] [bptt_persistent=1]
// Prints ms per step and per part (forward only; forward + backward; clip + Adam; refresh) and the losses of the last step.
// First run: round 4 (profiles/r04_v5_time_train_native.txt).
#include "native_model.h"

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, T_in = argc > 2 ? atoi(argv[2]) : 128, T_out = argc > 3 ? atoi(argv[3]) : 512;
  const int reps = argc > 4 ? atoi(argv[4]) : 8, gemm = argc > 5 ? atoi(argv[5]) : 4, bptt = argc > 6 ? atoi(argv[6]) : 1;
  taco_hparams hp;
  native_hparams(200, hp);
  CK(hipSetDevice(0));
  taco_train* t = nullptr;
  TK(taco_train_create(&hp, 0, &t));
  taco_model* mh = taco_train_model(t);
  const size_t np_ = taco_train_num_params(t);
  std::vector<float> hparams_flat(np_, 0.f);
  const int nw = taco_model_num_weights(mh);
  for (int i = 0; i < nw; ++i) {
    char name[256]; int64_t shp[4] = {0, 0, 0, 0}; int nd = 0;
    TK(taco_model_weight_name(mh, i, name, sizeof name, shp, &nd));
    size_t cnt = 1, off = 0;
    for (int d = 0; d < nd; ++d) cnt *= (size_t)shp[d];
    TK(taco_train_param_offset(t, name, &off));
    if (off + cnt > np_) { printf("parameter %s outside the flat buffer\n", name); return 1; }
    native_fill(name, shp, nd, hparams_flat.data() + off, cnt);
  }
  TK(taco_train_set_exact_gemm(t, gemm));
  TK(taco_train_set_bptt_engine(t, bptt));
  hipStream_t st; CK(hipStreamCreate(&st));
  float *d_p, *d_g, *d_m, *d_v, *d_mt, *d_lt, *d_loss, *d_gn;
  int32_t *d_ids, *d_len;
  CK(hipMalloc(&d_p, np_ * 4)); CK(hipMalloc(&d_g, np_ * 4)); CK(hipMalloc(&d_m, np_ * 4)); CK(hipMalloc(&d_v, np_ * 4));
  CK(hipMemcpy(d_p, hparams_flat.data(), np_ * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_g, 0, np_ * 4)); CK(hipMemset(d_m, 0, np_ * 4)); CK(hipMemset(d_v, 0, np_ * 4));
  TK(taco_train_refresh(t, st, d_p));
  std::vector<int32_t> ids((size_t)B * T_in), lens(B, T_in);            // training lengths include the EOS (datafeeder.py:294)
  for (auto& x : ids) x = 2 + (int)(urand() * 77.99f);
  for (int b = 0; b < B; ++b) ids[(size_t)b * T_in + T_in - 1] = 1;
  CK(hipMalloc(&d_ids, ids.size() * 4)); CK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_len, B * 4)); CK(hipMemcpy(d_len, lens.data(), B * 4, hipMemcpyHostToDevice));
  const size_t nm = (size_t)B * T_out * hp.num_mels, nl = (size_t)B * T_out * hp.num_freq;
  { std::vector<float> h(nl); for (auto& x : h) x = urand();
    CK(hipMalloc(&d_mt, nm * 4)); CK(hipMemcpy(d_mt, h.data(), nm * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_lt, nl * 4)); CK(hipMemcpy(d_lt, h.data(), nl * 4, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&d_loss, 64)); CK(hipMalloc(&d_gn, 64));
  const size_t wsb = taco_train_workspace_bytes(t, B, T_in, T_out);
  void* d_ws; CK(hipMalloc(&d_ws, wsb));
  void* d_aws; CK(hipMalloc(&d_aws, 1 << 16));
  hipEvent_t e[5]; for (auto& x : e) CK(hipEventCreate(&x));
  printf("C ABI training-step timer: B=%d T_in=%d T_out=%d, %.3f M parameters, workspace %.1f MB, GEMM engines %d, persistent BPTT %d, %d timed steps\n",
         B, T_in, T_out, np_ * 1e-6, wsb / 1048576.0, gemm, bptt, reps);
  auto fb = [&](float* grads) {
    return taco_train_forward_backward(t, st, d_p, grads, d_ids, d_len, nullptr, d_mt, d_lt, nullptr, B, T_in, T_out, 0, 24000, d_loss,
                                       nullptr, nullptr, nullptr, 0, d_ws, wsb);
  };
  long long step = 0;
  double acc[4] = {0, 0, 0, 0};
  for (int i = 0; i < 2 + reps; ++i) {
    CK(hipEventRecord(e[0], st));
    TK(fb(nullptr));                                                     // forward only (a loss fetch)
    CK(hipEventRecord(e[1], st));
    TK(fb(d_g));                                                         // the step's forward + backward
    CK(hipEventRecord(e[2], st));
    TK(taco_adam_step_f32(st, d_p, d_g, d_m, d_v, np_, step, taco_learning_rate(step, 0.002f, 0, 1), 0.9f, 0.999f, 1e-8f, 1.0f, d_gn, d_aws, 1 << 16));
    CK(hipEventRecord(e[3], st));
    TK(taco_train_refresh(t, st, d_p));
    CK(hipEventRecord(e[4], st));
    CK(hipStreamSynchronize(st));
    ++step;
    if (i >= 2)
      for (int k = 0; k < 4; ++k) { float ms; CK(hipEventElapsedTime(&ms, e[k], e[k + 1])); acc[k] += ms; }
  }
  float losses[4], gn;
  CK(hipMemcpy(losses, d_loss, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&gn, d_gn, 4, hipMemcpyDeviceToHost));
  int err = 0; TK(taco_model_device_errors(mh, &err));
  printf("  forward only %.3f ms | forward + backward %.3f ms | clip + Adam %.3f ms | refresh %.3f ms  =>  step (fwd+bwd, Adam, refresh) %.3f ms\n",
         acc[0] / reps, acc[1] / reps, acc[2] / reps, acc[3] / reps, (acc[1] + acc[2] + acc[3]) / reps);
  printf("  last step: loss %.5f mel %.5f linear %.5f without_coeff %.5f, gradient norm %.4f%s\n", losses[0], losses[1], losses[2], losses[3], gn,
         err ? "   (DEVICE ERROR WORD SET)" : "");
  int info[16]; TK(taco_debug_decoder_info(mh, info));
  printf("  decoder protocol %d, BPTT protocol %d\n", info[0], info[9]);
  taco_train_destroy(t);
  return 0;
}
