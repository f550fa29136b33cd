// native_model.h -- shared by the C ABI tools (time_stages_native.cpp, time_layers_native.cpp): a model of the reference architecture with
// random-init weights, created through include/taco_abi.h alone (every tensor the model asks for, by name and shape).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "taco_abi.h"
#include "taco_debug.h"      // the timing tools use the A/B switches

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define TK(x) do { int r_ = (x); if (r_ != 0) { printf("taco error %d at line %d: %s\n", r_, __LINE__, taco_last_error()); return 1; } } while (0)

static unsigned g_seed = 20240927u;
static float urand() { g_seed = g_seed * 1664525u + 1013904223u; return (float)((g_seed >> 8) & 0xFFFFFF) / 16777216.f; }
static bool ends_with(const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

// random-init values of one tensor by its name (BatchNorm statistics randomised as SURVEY 8d asks, TF's GRU / highway bias conventions,
// Glorot-uniform kernels)
static void native_fill(const char* name, const int64_t* shp, int nd, float* v, size_t cnt) {
  const std::string s(name);
  auto all = [&](auto&& f) { for (size_t i = 0; i < cnt; ++i) v[i] = f(); };
  if (ends_with(s, "moving_variance") || ends_with(s, "gamma")) all([] { return 0.5f + urand(); });
  else if (ends_with(s, "moving_mean") || ends_with(s, "beta")) all([] { return 0.2f * (urand() - 0.5f); });
  else if (ends_with(s, "gates/bias")) all([] { return 1.0f; });                       // TF-sem GRUCell
  else if (ends_with(s, "/T/bias")) all([] { return -1.0f; });                        // modules.py:119
  else if (ends_with(s, "bias") || nd == 0) all([] { return 0.f; });
  else {
    double fan_in = 1;
    for (int d = 0; d + 1 < nd; ++d) fan_in *= (double)shp[d];
    float lim = nd >= 2 ? (float)std::sqrt(6.0 / (fan_in + (double)shp[nd - 1])) : 0.5f;
    if (lim > 0.5f) lim = 0.5f;
    all([lim] { return (2.f * urand() - 1.f) * lim; });
  }
}

// hparams.py's effective defaults (single speaker, bah_mon, r = 4) with max_iters = n; returns 0 and the finalized model
static void native_hparams(int n, taco_hparams& hp) {
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
}

static int native_model(int n, taco_hparams& hp, taco_model*& m, int& nw, size_t& nparam) {
  native_hparams(n, hp);
  CK(hipSetDevice(0));
  m = nullptr;
  TK(taco_model_create(&hp, 0, &m));
  nw = taco_model_num_weights(m);
  nparam = 0;
  for (int i = 0; i < nw; ++i) {
    char name[256]; int64_t shp[4] = {0, 0, 0, 0}; int nd = 0;
    TK(taco_model_weight_name(m, i, name, sizeof name, shp, &nd));
    size_t cnt = 1;
    for (int d = 0; d < nd; ++d) cnt *= (size_t)shp[d];
    nparam += cnt;
    std::vector<float> v(cnt);
    native_fill(name, shp, nd, v.data(), cnt);
    TK(taco_model_set_weight(m, name, v.data(), shp, nd));
  }
  TK(taco_model_finalize(m));
  return 0;
}
