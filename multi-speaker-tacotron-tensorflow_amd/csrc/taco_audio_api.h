// taco_audio_api.h -- C ABI of the spectrogram -> waveform step; included inside extern "C".

int taco_gl_create(const taco_audio_hparams* hp, int device, taco_gl** out) {
  if (!hp || !out) return fail(TACO_ERR_ARG, "null argument");
  taco_gl* g = new taco_gl();
  g->hp = *hp;
  g->F = hp->num_freq; g->n_fft = (hp->num_freq - 1) * 2;                       // audio/__init__.py:118-122
  g->hop = (int)(hp->frame_shift_ms / 1000.0 * hp->sample_rate);
  g->win = (int)(hp->frame_length_ms / 1000.0 * hp->sample_rate);
  if (g->F < 2 || g->hop < 1 || g->win < 2 || g->win > g->n_fft) { delete g; return fail(TACO_ERR_ARG, "bad STFT parameters"); }
  g->lpad = (g->n_fft - g->win) / 2;
  const int N = g->n_fft, F = g->F, W = g->win;
  const double PI2 = 6.283185307179586476925286766559;
  std::vector<double> w(W);
  for (int n = 0; n < W; ++n) w[n] = 0.5 - 0.5 * std::cos(PI2 * n / W);          // periodic Hann (fftbins=True)
  std::vector<float> fw((size_t)W * 2 * F), iv((size_t)2 * F * W), w2(N, 0.f);
  for (int n = 0; n < W; ++n) {
    w2[g->lpad + n] = (float)(w[n] * w[n]);
    for (int k = 0; k < F; ++k) {
      const long q = ((long)k * (n + g->lpad)) % N;                               // exact angle reduction
      const double c = std::cos(PI2 * q / N), s = std::sin(PI2 * q / N);
      fw[(size_t)n * 2 * F + k] = (float)(w[n] * c);                              // Re X_k =  sum f w cos
      fw[(size_t)n * 2 * F + F + k] = (float)(-w[n] * s);                         // Im X_k = -sum f w sin
      const double ck = (k == 0 || k == N / 2) ? 1.0 : 2.0;                       // irfft: Hermitian half counted twice
      iv[(size_t)k * W + n] = (float)(ck / N * c * w[n]);
      iv[(size_t)(F + k) * W + n] = (float)((k == 0 || k == N / 2) ? 0.0 : -ck / N * s * w[n]);
    }
  }
  taco_model* gm = new taco_model();
  gm->device = device; gm->bf3 = 1;
  g->gm = gm;
  int Kq, NT;
  g->fwd.kw = 1; g->fwd.cin = W; g->fwd.N = 2 * F;
  g->fwd.wp = pack_w32(gm, fw.data(), 1, W, 2 * F, &g->fwd.cin_pad, &Kq, &NT);
  pack_bf3(gm, fw.data(), 1, W, 2 * F, &g->fwd.bh, &g->fwd.bl, &g->fwd.K16, &g->fwd.cin_pad16);
  g->inv.kw = 1; g->inv.cin = 2 * F; g->inv.N = W;
  g->inv.wp = pack_w32(gm, iv.data(), 1, 2 * F, W, &g->inv.cin_pad, &Kq, &NT);
  pack_bf3(gm, iv.data(), 1, 2 * F, W, &g->inv.bh, &g->inv.bl, &g->inv.K16, &g->inv.cin_pad16);
  g->w2 = arena_put(gm, w2.data(), w2.size());
  add_var(gm, g->fwd, 0); add_var(gm, g->inv, 0);
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&gm->darena, gm->harena.size() * sizeof(float)) != hipSuccess ||
      hipMemcpy(gm->darena, gm->harena.data(), gm->harena.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    delete gm; delete g; return fail(TACO_ERR_HIP, "could not upload the DFT packs");
  }
  for (auto& v : gm->hvars) {
    v.wp = AP(gm, (size_t)v.wp); v.wp2 = nullptr; v.bias = nullptr; v.bias2 = nullptr; v.bn_scale = nullptr; v.bn_shift = nullptr;
    v.bh = (const unsigned short*)AP(gm, (size_t)v.bh); v.bl = (const unsigned short*)AP(gm, (size_t)v.bl); v.bh2 = nullptr; v.bl2 = nullptr;
  }
  gm->harena.clear(); gm->harena.shrink_to_fit();
  gm->finalized = true;
  *out = g;
  return 0;
}

void taco_gl_destroy(taco_gl* g) {
  if (!g) return;
  if (g->gm) { if (g->gm->darena) (void)hipFree(g->gm->darena); delete g->gm; }
  delete g;
}

int taco_gl_num_samples(const taco_gl* g, int T) { return (g && T > 0) ? g->hop * (T - 1) : 0; }

size_t taco_gl_workspace_bytes(const taco_gl* g, int B, int T) {
  if (!g || B <= 0 || T <= 1) return 0;
  Carver cv(nullptr, 0);
  GlWs w; carve_gl(cv, g, B, T, w);
  return cv.off;
}

int taco_gl_inv_spectrogram(taco_gl* g, void* hip_stream, const float* d_spec, const float* d_init_uniform, unsigned long long seed,
                            int B, int T, int iters, float* d_wav, void* d_workspace, size_t workspace_bytes) {
  if (!g || !d_spec || !d_wav || !d_workspace || B <= 0 || T <= 1) return fail(TACO_ERR_ARG, "bad argument");
  const int L = g->hop * (T - 1), half = g->n_fft / 2;
  if (L <= half) return fail(TACO_ERR_SHAPE, "utterance too short for reflect padding: hop*(T-1) = %d <= n_fft/2 = %d", L, half);
  HIPCHK(hipSetDevice(g->gm->device));
  hipStream_t st = (hipStream_t)hip_stream;
  Carver cv(d_workspace, workspace_bytes);
  GlWs w; carve_gl(cv, g, B, T, w);
  if (!cv.ok()) return fail(TACO_ERR_STATE, "workspace too small: need %zu bytes, have %zu", cv.off, workspace_bytes);
  const int Tr = gl_rows(g, T), F = g->F;
  const size_t R = (size_t)B * Tr, slot = gl_slot(g, T);
  if (iters < 0) iters = g->hp.griffin_lim_iters;
  HIPCHK(zero_async(w.ypad, ((size_t)B * slot + 2 * g->n_fft + g->win) * sizeof(float), st));
  hipLaunchKernelGGL(k_gl_wss, EWGRID((size_t)L + g->n_fft), 0, st, AP(g->gm, g->w2), w.wss, T, g->n_fft, g->hop);
  hipLaunchKernelGGL(k_gl_magnitude, EWGRID(R * F), 0, st, d_spec, w.S, B, T, Tr, F, g->hp.min_level_db, g->hp.ref_level_db, g->hp.power);
  hipLaunchKernelGGL(k_gl_init_phase, EWGRID(R * F), 0, st, w.S, d_init_uniform, seed, w.X, B, T, Tr, F);
  HIPCHK(hipGetLastError());
  auto synth = [&]() -> int {     // y = istft(X): frames = X . IDFT_w ; overlap-add / window sum-square ; reflect pad for the next stft
    GemmCall c; c.x = w.X; c.ldx = 2 * F; c.M = (int)R; c.out = w.Y; c.ldo = g->win;
    TRY(run_gemm(g->gm, st, &g->inv, 1, false, c));
    hipLaunchKernelGGL(k_gl_overlap_add, EWGRID((size_t)B * L), 0, st, w.Y, w.wss, w.ypad, B, T, Tr, g->win, g->hop, g->lpad, g->n_fft, slot);
    hipLaunchKernelGGL(k_gl_reflect, EWGRID((size_t)B * half), 0, st, w.ypad, B, T, g->hop, g->n_fft, slot);
    HIPCHK(hipGetLastError());
    return 0;
  };
  TRY(synth());
  for (int it = 0; it < iters; ++it) {
    // est = stft(y): row (b, t) of the frame matrix is the hop-strided window ypad[b*slot + t*hop + lpad ...][0 .. win)
    GemmCall c; c.x = w.ypad + g->lpad; c.ldx = g->hop; c.M = (int)R; c.out = w.est; c.ldo = 2 * F;
    TRY(run_gemm(g->gm, st, &g->fwd, 1, false, c));
    hipLaunchKernelGGL(k_gl_project, EWGRID(R * F), 0, st, w.est, w.S, w.X, R, F);
    HIPCHK(hipGetLastError());
    TRY(synth());
  }
  hipLaunchKernelGGL(k_inv_preemphasis, dim3(B), dim3(1024), 0, st, w.ypad, d_wav, L, half, slot, g->hp.preemphasis);
  HIPCHK(hipGetLastError());
  return 0;
}
