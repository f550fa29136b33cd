// time_layers_native.cpp -- the large feed-forward layers of a C2 forward one by one through the op-level C ABI (taco_conv1d_bn_f32 /
// taco_dense_f32 / taco_highway_f32), under the exact-fp32 kernel, the split-bf16 kernel as the library picks its tile, and every forced
// tile of taco_debug_set_bf3 -- tools/time_gemm_layers.py without Python (seconds on a fresh box).  TF-eq = 2 M N K / time (the
// split-bf16 kernel issues three bf16 MFMAs per product, so its share of the 2.5 PF bf16 pipe is 3 x TF-eq / 2500).
//   hipcc -O2 -I include tools/time_layers_native.cpp -L multi-speaker-tacotron-tensorflow_amd/csrc -ltaco_hip \
//         -Wl,-rpath,'$ORIGIN/../multi-speaker-tacotron-tensorflow_amd/csrc' -o tools/time_layers_native
//   ./tools/time_layers_native [reps=20]
#include "native_model.h"

struct Case { const char* kind; const char* layer; int B, T, Cin, Cout, kw, mpw, act; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  taco_hparams hp; taco_model* m = nullptr; int nw = 0; size_t nparam = 0;
  if (native_model(128, hp, m, nw, nparam)) return 1;
  const Case cases[] = {
      {"conv", "post_cbhg/proj_1", 32, 512, 2048, 256, 3, 2, 1},
      {"conv", "post_cbhg/conv_bank/conv1d_8", 32, 512, 80, 256, 8, 1, 1},
      {"conv", "post_cbhg/proj_2", 32, 512, 256, 80, 3, 1, 0},
      {"dense", "post_cbhg/dense", 32, 512, 80, 256, 1, 1, 0},
      {"hw", "post_cbhg/highway_1", 32, 512, 256, 256, 1, 1, 0},
      {"dense", "linear", 32, 512, 512, 1025, 1, 1, 0},
      {"conv", "encoder_cbhg/proj_1", 32, 128, 2048, 128, 3, 2, 1},
      {"conv", "encoder_cbhg/proj_2", 32, 128, 128, 128, 3, 1, 0},
      {"conv", "encoder_cbhg/conv_bank/conv1d_16", 32, 128, 128, 128, 16, 1, 1},
      {"hw", "encoder_cbhg/highway_1", 32, 128, 128, 128, 1, 1, 0},
  };
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *d_x, *d_o;
  const size_t nx = (size_t)32 * 512 * 2048, no = (size_t)32 * 512 * 1025;
  CK(hipMalloc(&d_x, nx * 4)); CK(hipMalloc(&d_o, no * 4));
  { std::vector<float> h(nx); for (auto& v : h) v = urand() - 0.5f; CK(hipMemcpy(d_x, h.data(), nx * 4, hipMemcpyHostToDevice)); }
  printf("feed-forward layers of C2 through the op-level C ABI, %d timed calls each (us | TF-eq)\n", reps);
  const int tiles[] = {1, 3, 4, 5, 7, 9, 10, 11};
  printf("%-36s %8s | %-15s | %-15s |", "layer", "GFLOP", "exact fp32", "split-bf16 auto");
  for (int tn : tiles) printf(" tile %-2d        |", tn);
  printf("\n");
  for (const Case& c : cases) {
    const double gf = 2.0 * c.B * c.T * c.Cin * c.Cout * c.kw * (strcmp(c.kind, "hw") == 0 ? 2 : 1) / 1e9;
    auto call = [&]() {
      if (strcmp(c.kind, "conv") == 0) return taco_conv1d_bn_f32(m, st, c.layer, d_x, c.B, c.T, c.act, c.mpw, d_o);
      if (strcmp(c.kind, "dense") == 0) return taco_dense_f32(m, st, c.layer, d_x, c.B * c.T, c.act, d_o);
      return taco_highway_f32(m, st, c.layer, d_x, c.B * c.T, d_o);
    };
    auto timed = [&](double& us) -> int {
      for (int i = 0; i < 3; ++i) TK(call());
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) TK(call());
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      us = ms * 1e3 / reps;
      return 0;
    };
    printf("%-36s %8.2f |", c.layer, gf);
    double us;
    TK(taco_debug_set_bf3(m, 0, 0));
    if (timed(us)) return 1;
    printf(" %7.1f %6.1f  |", us, gf / us * 1e3);      // GFLOP / us = PFLOP/s; printed as TFLOP/s
    TK(taco_debug_set_bf3(m, 1, 0));
    if (timed(us)) return 1;
    printf(" %7.1f %6.1f  |", us, gf / us * 1e3);
    for (int tn : tiles) {
      TK(taco_debug_set_bf3(m, 1, tn));
      if (timed(us)) return 1;
      printf(" %7.1f %6.1f|", us, gf / us * 1e3);
    }
    TK(taco_debug_set_bf3(m, 1, 0));
    printf("\n");
  }
  taco_model_destroy(m);
  return 0;
}
