// taco_train.h -- host side of the training path (included at the end of taco_lib.hip):
//   * taco_train: a "shadow" taco_model whose weight packs are regenerated on the device from ONE flat fp32
//     parameter buffer (index maps built by running the normal packers over index-valued tensors), plus the
//     transposed packs the backward pass needs;
//   * training-mode forward (tacotron.py:26 is_training graph: batch-statistics BatchNorm with moving-average
//     updates, TacoTrainingHelper teacher forcing helpers.py:35-67) that records a tape in the caller's workspace;
//   * the backward pass (tf.gradients of tacotron.py:274-336's loss), gradients accumulated into one flat buffer
//     laid out like the parameters (the single RCCL all-reduce bucket of the data-parallel step).
// modules.py:24 calls tf.layers.dropout without training=True, so the prenet dropout is the identity in the
// reference even when training; nothing is dropped here either.
// Supported for training: single-speaker and multi-speaker ('simple', 'deepvoice') models, attention bah / bah_norm / bah_mon.
#pragma once

struct CbhgT {
  std::vector<ConvL> bank_f, bank_d;   // training-forward (no folded BN) and data-gradient layers, order of Cbhg.bank
  std::vector<ConvL> proj_f, proj_d;
  ConvL dense_d;
  std::vector<ConvL> hw_d;             // [2D -> D]: rows = [W_H^T ; W_T^T]
  ConvL xproj_d;                       // [6H -> I]
  size_t ghT[2] = {0, 0}, chT[2] = {0, 0};   // transposed h-rows of the GRU kernels (raw, arena offsets)
};
struct GruT { SkW gT, cT; int I = 0, H = 0; };   // gates/kernel^T (K = 2H, N = I+H), candidate/kernel^T (K = H, N = I+H)
struct TrainPacks {
  std::vector<ConvL> encpre_d;
  CbhgT enc, post;
  ConvL mem_d, lin_d, frame_d;         // frame_d: [r*num_mels -> Hd], the frame projection's data gradient for all steps at once (persistent BPTT)
  std::vector<SkW> decpre_T;
  SkW decpre0_ctxT;                     // transposed decoder prenet layer 1, context rows only: [P0 -> D] (the teacher frame gets no gradient)
  GruT att, dec[4];
  SkW concat_T, frame_T;
  size_t wqT = 0;
  std::vector<SkW> spk_T;              // deepvoice speaker layers transposed: [dim_i -> S]
  SkW lin_spk_T;                       // 'simple': speaker rows of the linear head transposed [F -> S]
};

struct taco_train {
  taco_model* sm = nullptr;            // shadow model (arena = gather of the flat parameters)
  TrainPacks tp;
  std::map<std::string, size_t> poff;  // flat offset of every spec tensor
  size_t NP = 0, arena_n = 0;
  float* d_map = nullptr;              // index map of the arena
  size_t n_bf3 = 0; int n_bf3_segs = 0; bool want_bf3_planes = false; bool bf3_current = false;   // planes regenerated from the current parameters by the last refresh?
  int wgrad_bf3 = 1, dgrad_bf3 = 0, dgrad_exact = 0;   // engine switches of THIS trainer (taco_train_set_exact_wgrad / _gemm); a step installs them in the thread-local g_* the helpers read
  unsigned* d_bf3_idx = nullptr; Bf3Seg* d_bf3_segs = nullptr;   // index list and segment table of the split-bf16 packs (k_bf3_gather)
  float* d_fold = nullptr;             // [Z + 1, 3H] concat projection folded into decoder GRU 1 (k_dx_fold), the index map's second source
  // synchronised BatchNorm over the data-parallel group: the host sums a device vector in place over all ranks
  // (ordered on the step's stream); null = statistics of this rank's rows only
  void (*sync_fn)(void* user, float* d_vec, int n) = nullptr;
  void* sync_user = nullptr;
  int sync_world = 1;
  int resident_bwd_scan = 1;           // the encoder's backward scan with the recurrent kernels in registers (k_bigru_resb); 0 (with taco_train_set_bptt_engine(t, 0)): k_bigru_rows_bwd
  int bptt_persistent = 1;             // taco_train_set_bptt_engine: the decoder's BPTT as one persistent launch (k_decoder_bwd_xcd) where it fits
  mutable int planes_problems = 0;     // weight gradients of the last backward pass that were computed from pre-split planes (a conv bank counts once)
  int wgrad_planes = 1;                // taco_train_set_wgrad_planes: 1 = large weight gradients from pre-split bf16 planes (taco_wgrad_planes.h), 2 = every eligible one (tests), 0 = off
  int deterministic = 1;               // taco_train_set_deterministic: ordered two-stage sums (default since round 4) or fp32 atomics; the former needs DET_SCRATCH_FLOATS of workspace
};
#define WP_SCRATCH_COLS 2048           // plane scratch: room for (widest conv bank + WP_SCRATCH_COLS) columns of the longest row set, three bf16 planes each
#define DET_SCRATCH_FLOATS ((size_t)48 << 20)      // 192 MB: e.g. 30 M-slices of the largest weight gradient (post-net proj_1, 3 x 2048 x 256)

// ---------------------------------------------------------------------------------------------------------------
// backward packs (run inside taco_model_finalize of the shadow model, before the arena upload)
// ---------------------------------------------------------------------------------------------------------------
static ConvL make_conv_T(taco_model* m, const float* W, int kw, int cin, int N) {
  // data gradient of y[t] = sum_j W[j] x[t + j - padl] is the SAME-padded correlation of dy with
  // WT[j'][n][c] = W[kw-1-j'][c][n] and left padding kw-1-padl
  std::vector<float> WT((size_t)kw * N * cin);
  for (int j = 0; j < kw; ++j)
    for (int c = 0; c < cin; ++c)
      for (int n = 0; n < N; ++n) WT[((size_t)(kw - 1 - j) * N + n) * cin + c] = W[((size_t)j * cin + c) * N + n];
  ConvL L; L.kw = kw; L.cin = N; L.N = cin;
  int Kq, NT;
  L.wp = pack_w32(m, WT.data(), kw, N, cin, &L.cin_pad, &Kq, &NT);
  pack_bf3(m, WT.data(), kw, N, cin, &L.bh, &L.bl, &L.K16, &L.cin_pad16);
  add_var(m, L, 0);
  m->hvars[L.var_index].padl = kw - 1 - (kw - 1) / 2;
  return L;
}
static ConvL make_conv_T_named(taco_model* m, const std::string& name) {
  const HostTensor& k = T_(m, name + "/kernel");
  if (k.shape.size() == 3) return make_conv_T(m, k.data.data(), (int)k.shape[0], (int)k.shape[1], (int)k.shape[2]);
  return make_conv_T(m, k.data.data(), 1, (int)k.shape[0], (int)k.shape[1]);
}
static std::vector<float> transpose2d(const float* W, int rows, int cols, int r0 = 0, int nr = -1) {
  if (nr < 0) nr = rows - r0;
  std::vector<float> Tt((size_t)cols * nr);
  for (int r = 0; r < nr; ++r)
    for (int c = 0; c < cols; ++c) Tt[(size_t)c * nr + r] = W[(size_t)(r0 + r) * cols + c];
  return Tt;
}
static SkW pack_w16_T(taco_model* m, const float* W, int rows, int cols) {   // pack of W^T: K = cols, N = rows
  std::vector<float> Tt = transpose2d(W, rows, cols);
  return pack_w16(m, Tt.data(), rows, 0, cols, 0, rows, nullptr);
}
static ConvL conv_noBN(taco_model* m, const ConvL& L, int coff) {
  ConvL F = L; F.bns = F.bnb = 0; F.var_index = -1;      // (the split-bf16 planes are the same weights: kept)
  add_var(m, F, coff);
  return F;
}
static void build_cbhg_T(taco_model* m, const Cbhg& c, const std::string& sc, CbhgT& t) {
  for (size_t i = 0; i < c.bank.size(); ++i) {
    const int k = c.bank[i].kw;
    t.bank_f.push_back(conv_noBN(m, c.bank[i], (k - 1) * c.C));
    t.bank_d.push_back(make_conv_T_named(m, sc + "/conv_bank/conv1d_" + std::to_string(k)));
  }
  for (size_t i = 0; i < c.proj.size(); ++i) {
    t.proj_f.push_back(conv_noBN(m, c.proj[i], 0));
    t.proj_d.push_back(make_conv_T_named(m, sc + "/proj_" + std::to_string(i + 1)));
  }
  if (c.has_dense) t.dense_d = make_conv_T_named(m, sc + "/dense");
  const int D = c.rnn;
  for (int i = 0; i < c.depth; ++i) {
    const std::string n = sc + "/highway_" + std::to_string(i + 1);
    const auto& WH = T_(m, n + "/H/kernel").data; const auto& WT = T_(m, n + "/T/kernel").data;
    std::vector<float> cat((size_t)2 * D * D);          // [2D, D] = [W_H^T ; W_T^T]
    for (int r = 0; r < D; ++r)
      for (int q = 0; q < D; ++q) { cat[(size_t)q * D + r] = WH[(size_t)r * D + q]; cat[(size_t)(D + q) * D + r] = WT[(size_t)r * D + q]; }
    ConvL L; L.kw = 1; L.cin = 2 * D; L.N = D;
    int Kq, NT;
    L.wp = pack_w32(m, cat.data(), 1, 2 * D, D, &L.cin_pad, &Kq, &NT);
    pack_bf3(m, cat.data(), 1, 2 * D, D, &L.bh, &L.bl, &L.K16, &L.cin_pad16);
    add_var(m, L, 0);
    t.hw_d.push_back(L);
  }
  const int H = c.rnn, I = c.rnn;
  std::vector<float> WxT((size_t)6 * H * I);              // [6H, I]: transposed hoisted input projection
  for (int dir = 0; dir < 2; ++dir) {
    const std::string n = sc + "/bigru/" + (dir ? "bw" : "fw");
    const auto& gk = T_(m, n + "/gates/kernel").data; const auto& ck = T_(m, n + "/candidate/kernel").data;
    for (int i = 0; i < I; ++i) {
      for (int j = 0; j < 2 * H; ++j) WxT[(size_t)(dir * 3 * H + j) * I + i] = gk[(size_t)i * 2 * H + j];
      for (int j = 0; j < H; ++j) WxT[(size_t)(dir * 3 * H + 2 * H + j) * I + i] = ck[(size_t)i * H + j];
    }
    std::vector<float> gT = transpose2d(gk.data(), I + H, 2 * H, I, H);   // [2H, H]
    std::vector<float> cT = transpose2d(ck.data(), I + H, H, I, H);       // [H, H]
    t.ghT[dir] = arena_put(m, gT.data(), gT.size());
    t.chT[dir] = arena_put(m, cT.data(), cT.size());
  }
  ConvL X; X.kw = 1; X.cin = 6 * H; X.N = I;
  int Kq, NT;
  X.wp = pack_w32(m, WxT.data(), 1, 6 * H, I, &X.cin_pad, &Kq, &NT);
  pack_bf3(m, WxT.data(), 1, 6 * H, I, &X.bh, &X.bl, &X.K16, &X.cin_pad16);
  add_var(m, X, 0);
  t.xproj_d = X;
}
static GruT make_gru_T(taco_model* m, const std::string& name, int I, int H) {
  GruT g; g.I = I; g.H = H;
  g.gT = pack_w16_T(m, T_(m, name + "/gates/kernel").data.data(), I + H, 2 * H);
  g.cT = pack_w16_T(m, T_(m, name + "/candidate/kernel").data.data(), I + H, H);
  return g;
}
static int build_train_packs(taco_model* m) {
  TrainPacks& tp = *m->tp;
  const taco_hparams& hp = m->hp;
  for (int i = 0; i < hp.enc_prenet_n; ++i) tp.encpre_d.push_back(make_conv_T_named(m, "prenet/dense_" + std::to_string(i + 1)));
  build_cbhg_T(m, m->enc, "encoder_cbhg", tp.enc);
  build_cbhg_T(m, m->post, "post_cbhg", tp.post);
  tp.mem_d = make_conv_T_named(m, "attention/memory_layer");
  tp.lin_d = make_conv_T_named(m, "linear");
  const int D = 2 * hp.enc_rnn_size, As = hp.attention_state_size, Hd = hp.dec_rnn_size, A = hp.attention_size;
  int d = hp.num_mels + D;
  for (int i = 0; i < hp.dec_prenet_n; ++i) {
    tp.decpre_T.push_back(pack_w16_T(m, T_(m, "decoder/prenet/dense_" + std::to_string(i + 1) + "/kernel").data.data(), d, hp.dec_prenet[i]));
    d = hp.dec_prenet[i];
  }
  { const HostTensor& k = T_(m, "decoder/prenet/dense_1/kernel");      // [Mm + D, P0]: rows Mm.. transposed -> [P0, D]
    const int Mm = hp.num_mels, P0 = hp.dec_prenet[0];
    std::vector<float> Tt = transpose2d(k.data.data(), Mm + D, P0, Mm, D);
    tp.decpre0_ctxT = pack_w16(m, Tt.data(), D, 0, P0, 0, D, nullptr); }
  tp.att = make_gru_T(m, "decoder/attention_gru", d + simple_S(m), As);
  for (int i = 0; i < hp.dec_layer_num; ++i) tp.dec[i] = make_gru_T(m, "decoder/gru_" + std::to_string(i + 1), Hd, Hd);
  tp.concat_T = pack_w16_T(m, T_(m, "decoder/concat_projection/kernel").data.data(), As + D + simple_S(m), Hd);
  if (is_simple(m)) tp.lin_spk_T = pack_w16_T(m, T_(m, "linear/kernel_spk").data.data(), hp.speaker_embedding_size, hp.num_freq);
  tp.frame_d = make_conv_T_named(m, "decoder/frame_projection");
  tp.frame_T = pack_w16_T(m, T_(m, "decoder/frame_projection/kernel").data.data(), Hd, hp.num_mels * hp.reduction_factor);
  { std::vector<float> wqT = transpose2d(T_(m, "attention/query_layer/kernel").data.data(), As, A);
    tp.wqT = arena_put(m, wqT.data(), wqT.size()); }
  if (is_deepvoice(m) && hp.speaker_embedding_size != 1) {
    std::vector<std::string> names = {kSpkNames[0], kSpkNames[1], kSpkNames[2]};
    for (int i = 0; i < hp.dec_layer_num; ++i) names.push_back("decoder_rnn_init_" + std::to_string(i + 1));
    for (auto& n : names) {
      const HostTensor& k = T_(m, "spk/" + n + "/kernel");
      tp.spk_T.push_back(pack_w16_T(m, k.data.data(), (int)k.shape[0], (int)k.shape[1]));
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------------------------
#define EWGRID(n) dim3((unsigned)(((size_t)(n) + 255) / 256)), dim3(256)
static inline bool al16h(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }      // may a matrix be read four columns at a time (the _v4 kernels)
// Deterministic reductions (taco_train_set_deterministic): the sums over rows that normally leave their workgroups through fp32
// atomics (weight gradients, bias / BatchNorm sums, embedding gradients, d attention_v) are written as per-slice partials into this
// scratch and added up in a fixed order by a second launch -- the step becomes run-to-run reproducible, as the reference's
// single-device step is.  Scratch of the step in flight; thread-local because the launch helpers below have no context argument
// (one step per host thread at a time: the library's threading contract).
struct DetScratch { float* p = nullptr; size_t cap = 0; };
static thread_local DetScratch g_det;
// Batching region for small weight gradients (k_wgrad_bf3_group): between wg_begin and wg_end, run_wgrad calls that take the one-wave
// tile are collected and launched together -- at wg_flush / wg_end, or when the table is full.  The caller flushes before anything
// overwrites an operand of a collected problem or reads one of their outputs.
// Deterministic mode: the collected problems write per-slice partials into consecutive regions of the deterministic scratch (det_used: the
// cursor; a problem that does not fit flushes the region first) and two more group launches add the slices up in a fixed order.  (Problems
// launched on their own in between use the scratch from its start: everything is stream ordered, the group's kernels run at the flush.)
struct WgBatch { WgGroup g; ColGroup c; WgRedGroup r; ColRedGroup cr; size_t det_used = 0; bool active = false; hipStream_t st = nullptr; };
static thread_local WgBatch g_wgb;
static int wg_flush() {      // (column sums of the region ride along: same hazards, same flush points)
  WgGroup& G = g_wgb.g; ColGroup& Cg = g_wgb.c;
  if (Cg.n > 0) hipLaunchKernelGGL(k_colsum_group, dim3(Cg.start[Cg.n]), dim3(256), 0, g_wgb.st, Cg);
  if (G.n > 0) hipLaunchKernelGGL(k_wgrad_bf3_group, dim3(G.start[G.n]), dim3(64), 0, g_wgb.st, G);
  if (g_wgb.cr.n > 0) hipLaunchKernelGGL(k_colsum_reduce_group, dim3(g_wgb.cr.start[g_wgb.cr.n]), dim3(256), 0, g_wgb.st, g_wgb.cr);
  if (g_wgb.r.n > 0) hipLaunchKernelGGL(k_wgrad_reduce_group, dim3(g_wgb.r.start[g_wgb.r.n]), dim3(256), 0, g_wgb.st, g_wgb.r);
  if (Cg.n > 0 || G.n > 0) HIPCHK(hipGetLastError());
  G.n = 0; G.start[0] = 0; Cg.n = 0; Cg.start[0] = 0;
  g_wgb.r.n = 0; g_wgb.r.start[0] = 0; g_wgb.cr.n = 0; g_wgb.cr.start[0] = 0; g_wgb.det_used = 0;
  return 0;
}
static void wg_begin(hipStream_t st) {
  g_wgb.active = true; g_wgb.st = st; g_wgb.g.n = 0; g_wgb.g.start[0] = 0; g_wgb.c.n = 0; g_wgb.c.start[0] = 0;
  g_wgb.r.n = 0; g_wgb.r.start[0] = 0; g_wgb.cr.n = 0; g_wgb.cr.start[0] = 0; g_wgb.det_used = 0;
}
static int wg_end() { const int rc = wg_flush(); g_wgb.active = false; return rc; }
struct WgRegion {      // RAII: a region ends (and flushes) on every return path
  explicit WgRegion(hipStream_t st) { wg_begin(st); }
  ~WgRegion() { if (g_wgb.active) (void)wg_end(); }
};
static int run_colsum(hipStream_t st, const float* a, int lda, const float* b, int ldb, const float* mu, const float* rstd,
                      float* out1, float* out2, int M, int C, int mode, bool assign = false) {
  // assign (deterministic mode only -- the caller zero-fills otherwise): the ordered sums replace out1 / out2 instead of being added to them
  ColArgs g; g.a = a; g.b = b; g.mu = mu; g.rstd = rstd; g.out1 = out1; g.out2 = out2; g.lda = lda; g.ldb = ldb; g.M = M; g.C = C;
  g.mode = mode; g.rpb = 256; g.part = nullptr;
  if (g_det.p) {
    while ((size_t)cdiv(M, g.rpb) * 2 * C > g_det.cap) g.rpb *= 2;
    g.part = g_det.p;
  }
  const int nchunks = cdiv(M, g.rpb);
  if (g_wgb.active) {      // inside a batching region: joins the group launch (the caller flushes before the sums are read)
    if (g.part) {          // deterministic: its own region of the scratch, summed by the group's reduce launch
      const size_t need = (size_t)nchunks * 2 * C;
      // (two problems of one reduce launch must not add into the same vector: their read-modify-writes would race, where the atomics
      // of the other mode simply accumulate -- e.g. the five speaker projections of model type deepvoice sum into one embedding gradient)
      bool clash = false;
      for (int i = 0; i < g_wgb.cr.n; ++i)
        clash = clash || (out1 && (g_wgb.cr.p[i].out1 == out1 || g_wgb.cr.p[i].out2 == out1)) || (out2 && (g_wgb.cr.p[i].out1 == out2 || g_wgb.cr.p[i].out2 == out2));
      if (clash || g_wgb.det_used + need > g_det.cap) TRY(wg_flush());
      g.part = g_det.p + g_wgb.det_used; g_wgb.det_used += need;
      ColRedGroup& R = g_wgb.cr;
      R.p[R.n].part = g.part; R.p[R.n].out1 = out1; R.p[R.n].out2 = out2; R.p[R.n].nchunks = nchunks; R.p[R.n].C = C; R.p[R.n].mode = mode | (assign ? 8 : 0);
      R.start[R.n + 1] = R.start[R.n] + cdiv(C, 256); ++R.n;
    }
    ColGroup& G = g_wgb.c;
    G.p[G.n] = g;
    G.start[G.n + 1] = G.start[G.n] + cdiv(C, 64) * nchunks;
    if (++G.n == COL_MAXP) return wg_flush();
    return 0;
  }
  hipLaunchKernelGGL(k_colsum, dim3(cdiv(C, 64), nchunks), dim3(256), 0, st, g);
  if (g.part) hipLaunchKernelGGL(k_colsum_reduce, EWGRID(C), 0, st, (const float*)g.part, nchunks, C, mode | (assign ? 8 : 0), out1, out2);
  HIPCHK(hipGetLastError());
  return 0;
}
// ---- weight gradients from pre-split planes (taco_wgrad_planes.h) ----
// The operands of a LARGE weight gradient are converted once into bf16 planes (k_wp_split: fragment-major, the tap shifts and their
// batch-row masks applied to copies of the narrower operand) and multiplied by a kernel that converts nothing (k_wp_gemm).  Measured per
// problem against k_wgrad_bf3 (tools/ubench_wgrad_planes.hip, profiles/r06_ubench_wgrad_planes.txt): the conversion pass is bound by
// its 10 bytes per element, so it pays where an element meets many products -- post-net proj_1 (542 -> 382 us), the linear head
// (365 -> 208), a whole conv bank as ONE product launch over one conversion of its dz (post-net: ~1240 -> ~330) -- and loses on
// 256 x 256-sized problems (highway, GRU kernels, the decoder's hoisted gradients), which stay on k_wgrad_bf3.
static thread_local int g_wgrad_planes = 1;            // installed from the trainer for the duration of a step (EngineGuard)
struct WpScratch { uint4* p = nullptr; size_t cap = 0; };
static thread_local WpScratch g_wps;
static thread_local int g_wp_count = 0;               // problems of the step in flight that took this path
#define WP_MIN_MACS 3.0e9                               // mode 1: problems below this many multiply-adds stay on k_wgrad_bf3
constexpr int WP_SM = 1, WP_NB = 3;                    // 16-row stages, ring of three
template <int WK, int WN>
static int wp_gemm_launch_t(hipStream_t st, const WpGemmArgs& g, int z) {
  constexpr size_t lds = (size_t)WP_NB * 3 * 2 * (WK + WN) * WP_SM * 1024;
  static const hipError_t attr = hipFuncSetAttribute((const void*)k_wp_gemm<WK, WN, WP_SM, WP_NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  HIPCHK(attr);
  hipLaunchKernelGGL((k_wp_gemm<WK, WN, WP_SM, WP_NB>), dim3(cdiv(g.K, 64 * WK), cdiv(g.N, 64 * WN), z), dim3(64 * WK * WN), lds, st, g);
  HIPCHK(hipGetLastError());
  return 0;
}
// workgroup tile (k x n): 128 x 128 (four waves, 72 KB of LDS, two workgroups per CU)
static void wp_tile(int K, int N, int& tk, int& tn) { (void)K; (void)N; tk = 128; tn = 128; }
static int wp_gemm_launch(hipStream_t st, const WpGemmArgs& g, int tk, int tn, int z) {
  if (tk == 128 && tn == 128) return wp_gemm_launch_t<2, 2>(st, g, z);
  return fail(TACO_ERR_STATE, "no k_wp_gemm instantiation for a %d x %d tile", tk, tn);
}
static int wp_rows_per_slice(int Mp, long tiles, size_t per) {      // ~768 workgroups; in deterministic mode the slices' partial tiles must fit the scratch
  int rpb = Mp;
  while (rpb > 256 && tiles * cdiv(Mp, rpb) < 768) rpb >>= 1;
  rpb = cdiv(rpb, 16 * WP_SM) * 16 * WP_SM;
  if (g_det.p) while ((size_t)cdiv(Mp, rpb) * per > g_det.cap && rpb < Mp) rpb *= 2;
  return rpb;
}
static void wp_split_launch(hipStream_t st, const float* src, const int* gather, int ld, int M, int T, int C, int Mp, int ncopy, int sigma0, int dsigma, uint4* out) {
  WpSplitArgs a; a.src = src; a.gather = gather; a.out = out; a.ld = ld; a.M = M; a.T = T; a.C = C; a.Mp = Mp; a.ncopy = ncopy; a.sigma0 = sigma0; a.dsigma = dsigma;
  hipLaunchKernelGGL(k_wp_split, dim3(cdiv(cdiv(C, 32), 4), Mp / 64, ncopy), dim3(256), 0, st, a);
}
// one weight gradient (the arguments of run_wgrad); handled = false: not eligible, nothing was launched
static int run_wgrad_planes(hipStream_t st, const float* x, const int* gather, int ldx, const float* dy, int ldy, float* dw, int lddw,
                            int M, int T, int K, int N, int kw, int padl, bool& handled) {
  handled = false;
  if (!g_wgrad_planes || !g_wps.p) return 0;
  if (g_wgrad_planes == 1 && (double)M * K * N * kw < WP_MIN_MACS) return 0;
  const bool shifted = kw > 1 || padl != 0;
  if (shifted && (gather || T <= 0)) return 0;
  const int Mp = cdiv(M, 64) * 64;
  const bool a_per_tap = K <= N;            // the narrower operand carries the tap copies
  const size_t na = wp_plane_uint4(K, Mp, a_per_tap ? kw : 1), nb = wp_plane_uint4(N, Mp, a_per_tap ? 1 : kw);
  const size_t per = (size_t)kw * K * N;
  if (na + nb > g_wps.cap || (g_det.p && per > g_det.cap)) return 0;
  uint4* pa = g_wps.p; uint4* pb = g_wps.p + na;
  // dW[tap] = sum_m X[m + s] dY[m], s = tap - padl: either X carries the shift s, or dY carries -s (m' = m + s)
  wp_split_launch(st, x, gather, ldx, M, T, K, Mp, a_per_tap ? kw : 1, (shifted && a_per_tap) ? -padl : 0, a_per_tap ? 1 : 0, pa);
  wp_split_launch(st, dy, nullptr, ldy, M, T, N, Mp, a_per_tap ? 1 : kw, (shifted && !a_per_tap) ? padl : 0, a_per_tap ? 0 : -1, pb);
  WpGemmArgs g; memset(&g, 0, sizeof g);
  g.a = pa; g.b = pb; g.K = K; g.N = N; g.Mp = Mp; g.kw = kw; g.a_per_tap = a_per_tap ? 1 : 0; g.dw = dw; g.lddw = lddw; g.nw = 0;
  int tk, tn; wp_tile(K, N, tk, tn);
  const long tiles = (long)cdiv(K, tk) * cdiv(N, tn) * kw;
  g.rpb = wp_rows_per_slice(Mp, tiles, per);
  g.part = g_det.p;
  const int nsplit = cdiv(Mp, g.rpb);
  TRY(wp_gemm_launch(st, g, tk, tn, kw * nsplit));
  if (g.part) hipLaunchKernelGGL(k_wgrad_reduce, EWGRID(per), 0, st, (const float*)g.part, nsplit, kw, K, N, dw, lddw);
  HIPCHK(hipGetLastError());
  handled = true; ++g_wp_count;
  return 0;
}
// ALL widths 1 .. nw of a conv bank (dz of every width complete in dY [M, nw * Cw]): one conversion of dz, nw shifted copies of the
// bank's input, one product launch, one ordered sum per width (a group launch).  dwk[k - 1] = the kernel gradient of width k [k][K][Cw].
static int run_wgrad_bank_planes(hipStream_t st, const float* x, int ldx, const float* dY, int ldy, float* const* dwk, int M, int T, int K, int Cw, int nw, bool& handled) {
  handled = false;
  if (!g_wgrad_planes || !g_wps.p || nw < 2 || nw > WP_MAXW || (Cw & 31) || T <= 0) return 0;
  const int ntap = nw * (nw + 1) / 2;
  if (g_wgrad_planes == 1 && (double)M * K * Cw * ntap < WP_MIN_MACS) return 0;
  const int Mp = cdiv(M, 64) * 64;
  const size_t na = wp_plane_uint4(K, Mp, nw), nb = wp_plane_uint4(nw * Cw, Mp, 1);
  const size_t per = (size_t)ntap * K * Cw;
  if (na + nb > g_wps.cap || (g_det.p && per > g_det.cap) || nw > WG_MAXP) return 0;
  uint4* pa = g_wps.p; uint4* pb = g_wps.p + na;
  wp_split_launch(st, x, nullptr, ldx, M, T, K, Mp, nw, -((nw - 1) / 2), 1, pa);       // copy j: shift j - (nw - 1) / 2; width k, tap: shift tap - (k - 1) / 2
  wp_split_launch(st, dY, nullptr, ldy, M, T, nw * Cw, Mp, 1, 0, 0, pb);
  WpGemmArgs g; memset(&g, 0, sizeof g);
  g.a = pa; g.b = pb; g.K = K; g.N = Cw; g.Mp = Mp; g.kw = 1; g.a_per_tap = 1; g.lddw = Cw; g.nw = nw;
  int tk, tn; wp_tile(K, Cw, tk, tn);
  const long tiles = (long)cdiv(K, tk) * cdiv(Cw, tn) * ntap;
  g.rpb = wp_rows_per_slice(Mp, tiles, per);
  g.part = g_det.p;
  const int nsplit = cdiv(Mp, g.rpb);
  size_t off = 0;
  for (int k = 1; k <= nw; ++k) { g.dwk[k - 1] = dwk[k - 1]; g.part_off[k - 1] = (unsigned)off; off += (size_t)nsplit * k * K * Cw; }
  TRY(wp_gemm_launch(st, g, tk, tn, ntap * nsplit));
  if (g.part) {
    WgRedGroup R; R.n = 0; R.start[0] = 0;
    for (int k = 1; k <= nw; ++k) {
      R.p[R.n].part = g.part + g.part_off[k - 1]; R.p[R.n].dw = dwk[k - 1]; R.p[R.n].nsplit = nsplit; R.p[R.n].kw = k; R.p[R.n].K = K; R.p[R.n].N = Cw; R.p[R.n].lddw = Cw;
      R.start[R.n + 1] = R.start[R.n] + cdiv(k * K * Cw, 256); ++R.n;
    }
    hipLaunchKernelGGL(k_wgrad_reduce_group, dim3(R.start[R.n]), dim3(256), 0, st, R);
  }
  HIPCHK(hipGetLastError());
  handled = true; ++g_wp_count;
  return 0;
}
// 1 (default): weight gradients on the bf16 matrix cores with operands split three ways and six products per tile (k_wgrad_bf3:
// fp32-grade); 0: exact-fp32 MFMA (k_wgrad).  taco_train_set_exact_wgrad.
static thread_local int g_wgrad_bf3 = 1;     // installed from the trainer for the duration of a step (EngineGuard)
static int run_wgrad(hipStream_t st, const float* x, const int* gather, int ldx, const float* dy, int ldy, float* dw, int lddw,
                     int M, int T, int K, int N, int kw = 1, int padl = 0, const int* ygather = nullptr) {
  WgArgs g; g.ygather = ygather; g.x = x; g.gather = gather; g.dy = dy; g.dw = dw; g.ldx = ldx; g.ldy = ldy; g.lddw = lddw; g.M = M; g.T = T; g.K = K; g.N = N;
  g.kw = kw; g.padl = padl;
  const bool bf3 = g_wgrad_bf3 != 0;
  if (bf3 && !ygather) {                     // large problems: from pre-split planes
    bool handled = false;
    TRY(run_wgrad_planes(st, x, gather, ldx, dy, ldy, dw, lddw, M, T, K, N, kw, padl, handled));
    if (handled) return 0;
  }
  // split-bf16 kernel: 128 x 128 tiles (4 waves) when those alone give a few hundred workgroups, else 64 x 64 tiles (1 wave): every
  // M-slice a workgroup takes ends in one atomic per output element, so slices are kept LONG (>= 256 rows where the grid allows)
  const long t128 = (long)cdiv(K, 128) * cdiv(N, 128) * kw, t64 = (long)cdiv(K, 64) * cdiv(N, 64) * kw;
  const bool big = bf3 && t128 >= 64;
  { const long tiles = big ? t128 : t64; int rpb = 1024;
    const long want = bf3 ? (big ? 768 : 1024) : 2048;      // workgroups (one-wave workgroups: four times as many fit a CU).  (Ordered sums: halving or doubling the
                                                             // slices of the small problems changes nothing, 15.87 / 15.88 ms; a quarter of them: 16.14 -- the partials' traffic is not what the mode costs)
    while (rpb > (bf3 ? 128 : 64) && tiles * cdiv(M, rpb) < want) rpb >>= 1;
    g.rpb = rpb; }
  g.part = nullptr;
  if (g_det.p) {   // per-slice partial tiles + an ordered sum instead of atomics; fewer, longer slices if the scratch is short
    const size_t per = (size_t)kw * K * N;
    if (per > g_det.cap) return fail(TACO_ERR_STATE, "deterministic-reduction scratch too small for a %d x %d x %d weight gradient", kw, K, N);
    while ((size_t)cdiv(M, g.rpb) * per > g_det.cap) g.rpb *= 2;
    g.part = g_det.p;
  }
  const int nsplit = cdiv(M, g.rpb);
  if (g_wgb.active && bf3 && !big) {       // small problem inside a batching region: joins the group launch
    if (g.part) {                          // deterministic: its own region of the scratch, summed by the group's reduce launch
      const size_t need = (size_t)nsplit * kw * K * N;
      bool clash = false;                  // (same rule as for the column sums: one writer per output per reduce launch)
      for (int i = 0; i < g_wgb.r.n; ++i) clash = clash || g_wgb.r.p[i].dw == dw;
      if (clash || g_wgb.det_used + need > g_det.cap) TRY(wg_flush());
      g.part = g_det.p + g_wgb.det_used; g_wgb.det_used += need;
      WgRedGroup& R = g_wgb.r;
      R.p[R.n].part = g.part; R.p[R.n].dw = dw; R.p[R.n].nsplit = nsplit; R.p[R.n].kw = kw; R.p[R.n].K = K; R.p[R.n].N = N; R.p[R.n].lddw = lddw;
      R.start[R.n + 1] = R.start[R.n] + cdiv(kw * K * N, 256); ++R.n;
    }
    WgGroup& G = g_wgb.g;
    G.p[G.n] = g;
    G.start[G.n + 1] = G.start[G.n] + cdiv(K, 64) * cdiv(N, 64) * kw * nsplit;
    if (++G.n == WG_MAXP) return wg_flush();
    return 0;
  }
  if (bf3 && big) hipLaunchKernelGGL((k_wgrad_bf3<4>), dim3(cdiv(K, 128), cdiv(N, 128), kw * nsplit), dim3(256), 0, st, g);
  else if (bf3) hipLaunchKernelGGL((k_wgrad_bf3<1>), dim3(cdiv(K, 64), cdiv(N, 64), kw * nsplit), dim3(64), 0, st, g);
  else hipLaunchKernelGGL(k_wgrad, dim3(cdiv(K, 64), cdiv(N, 64), kw * nsplit), dim3(256), 0, st, g);
  if (g.part) hipLaunchKernelGGL(k_wgrad_reduce, EWGRID((size_t)kw * K * N), 0, st, (const float*)g.part, nsplit, kw, K, N, dw, lddw);
  HIPCHK(hipGetLastError());
  return 0;
}
static void run_embed_bwd(hipStream_t st, const float* dx, const int* ids, float* dE, int M, int E, int V) {
  if (g_det.p) hipLaunchKernelGGL(k_embed_bwd_det, EWGRID((size_t)V * ((E + 63) / 64) * 64), 0, st, dx, ids, dE, M, E, V);      // one wave per (table row, 64 columns)
  else hipLaunchKernelGGL(k_embed_bwd, EWGRID((size_t)M * E), 0, st, dx, ids, dE, M, E);
}
static thread_local int g_dgrad_bf3 = 0;       // taco_train_set_exact_gemm(t, 3): forward GEMMs exact fp32, data gradients split-bf16
static thread_local int g_dgrad_exact = 0;     // taco_train_set_exact_gemm(t, 2): data gradients on the exact-fp32 MFMA, forward GEMMs split-bf16 (A/B hook)
// y = x . W^T style data gradient through k_gemm: out = conv_T(dy) (+ res)
static int run_dgrad(const taco_model* m, hipStream_t st, const ConvL& Ld, const float* dy, int lddy, int M, int T, float* out, int ldo,
                     const float* res = nullptr, int ldres = 0) {
  GemmCall g; g.x = dy; g.ldx = lddy; g.M = M; g.T = T; g.out = out; g.ldo = ldo; g.res = res; g.ldres = ldres;
  if (g_dgrad_exact) { ConvL E = Ld; E.bh = E.bl = 0; return run_gemm(m, st, &E, 1, false, g); }
  if (g_dgrad_bf3) { g_gemm_force_bf3 = 1; const int rc = run_gemm(m, st, &Ld, 1, false, g); g_gemm_force_bf3 = 0; return rc; }
  return run_gemm(m, st, &Ld, 1, false, g);
}

// ---------------------------------------------------------------------------------------------------------------
// tape layout
// ---------------------------------------------------------------------------------------------------------------
// BatchNorm + max-pool of a conv bank as one pass (k_bn_pool_bank_v4): the widths fit the pointer table and the columns go four at a time
static inline bool bank_pool_fused(const Cbhg& c) { return c.K <= BNB_MAXK && (c.C & 3) == 0; }
struct CbhgTape {
  float *bank_a, *bank_y, *pool, *bank_mu, *bank_rs;
  float *pa[4], *py[4], *pmu[4], *prs[4];
  float *res, *hx[10], *hH[8], *hT[8];
  float *xproj, *out, *gsave;
  float *dg, *rh, *d0, *d1, *dcat, *dbig0, *dbig1, *stat;
  unsigned long long* gxbuf = nullptr; unsigned* gxctl = nullptr;      // k_bigru_duo: exchange granules, census words
};
static void carve_cbhg_tape(Carver& cv, const Cbhg& c, int B, int T, CbhgTape& w) {
  const size_t M = (size_t)B * T, KC = (size_t)c.K * c.C;
  w.bank_a = cv.f(M * KC); w.bank_y = cv.f(bank_pool_fused(c) ? 1 : M * KC); w.pool = cv.f(M * KC); w.bank_mu = cv.f(KC); w.bank_rs = cv.f(KC);
  size_t wide = std::max<size_t>(c.in_dim, c.rnn);
  for (int i = 0; i < c.nproj; ++i) {
    w.pa[i] = cv.f(M * c.proj_dim[i]); w.py[i] = cv.f(M * c.proj_dim[i]); w.pmu[i] = cv.f(c.proj_dim[i]); w.prs[i] = cv.f(c.proj_dim[i]);
    wide = std::max<size_t>(wide, c.proj_dim[i]);
  }
  w.res = cv.f(M * c.in_dim);
  for (int i = 0; i <= c.depth; ++i) w.hx[i] = cv.f(M * c.rnn);
  for (int i = 0; i < c.depth; ++i) { w.hH[i] = cv.f(M * c.rnn); w.hT[i] = cv.f(M * c.rnn); }
  w.xproj = cv.f(M * 6 * c.rnn); w.out = cv.f(M * 2 * c.rnn); w.gsave = cv.f(M * 6 * c.rnn);
  w.dg = cv.f(M * 6 * c.rnn); w.rh = cv.f(M * 2 * c.rnn);
  w.d0 = cv.f(M * wide); w.d1 = cv.f(M * wide); w.dcat = cv.f(M * 2 * c.rnn);
  w.dbig0 = cv.f(M * KC); w.dbig1 = cv.f(M * KC); w.stat = cv.f(5 * std::max<size_t>(KC, wide));   // sums, centred sums + the 3-vector SyncBN exchange pack
  w.gxbuf = (unsigned long long*)cv.raw(std::max(gd_xbuf_granules(8), gb_xbuf_granules(8)) * sizeof(unsigned long long)); w.gxctl = (unsigned*)cv.raw(256);
}
struct DecTape {     // every per-step tensor is [B, n, W]: step t of row b at (b*n + t)*W
  float *keys, *zero, *ctx, *pz[4], *hA, *rA, *uA, *cA, *rhA, *xcA, *alpha, *alpha0;
  float *o[5], *h[4], *r[4], *u[4], *c[4], *rh[4], *xc[4];
  int* nz;
  float* tape256 = nullptr; size_t tstride = 0;          // the 256-wide per-step arrays in DXT_* order, back to back (persistent decoder, TAPE)
  unsigned long long* xbuf = nullptr; unsigned* dxctl = nullptr; float* rowbias = nullptr;   // its exchange granules, census words, 'simple' row biases
  // backward
  float *dkeys, *dvalues, *dv_acc, *dsb_acc, *dalpha, *dctx, *dctx_t, *dhA, *dh[4], *dht, *dhp, *tmp1, *tmp2, *do_[5];
  float *g_dgp[4], *g_dcp[4], *g_dgpA, *g_dcpA, *g_do0, *g_do2, *g_dq, *g_dz[4], *dpz, *dIn;
  float *g_q, *g_e, *g_de, *g_dctx;   // attention tape: processed query, raw scores, their gradients' inputs
};
static void carve_dec_tape(Carver& cv, const taco_model* m, int B, int T_in, int n, DecTape& w) {
  const taco_hparams& hp = m->hp;
  const int D = 2 * hp.enc_rnn_size, As = hp.attention_state_size, Hd = hp.dec_rnn_size, A = hp.attention_size, L = hp.dec_layer_num;
  const size_t R = (size_t)B * n;
  w.keys = cv.f((size_t)B * T_in * A); w.zero = cv.f((size_t)B * std::max(std::max(hp.num_mels, As), std::max(Hd, D)));
  const int S = simple_S(m);      // 'simple': the speaker embedding rides behind the context and behind the last prenet output of every step
  w.ctx = cv.f(R * (D + S));
  // The persistent decoder (k_decoder_xcd<RG, true>) addresses its 256-wide per-step arrays as ONE block, slot s at tape256 + s * R * 256
  // (DXT_* order: P1, HA, RA, UA, CA, RHA, Q, O0, R1, U1, C1, RH1, H1, O1, R2, U2, C2, RH2, H2, O2): carve them back to back in that order.
  const bool dxl = dx_widths_ok(m);     // reference widths: every one of them is [R, 256]
  if (dxl) {
    float** slot[DXT_N] = {&w.pz[0], &w.hA, &w.rA, &w.uA, &w.cA, &w.rhA, &w.g_q, &w.o[0], &w.r[0], &w.u[0], &w.c[0], &w.rh[0], &w.h[0], &w.o[1],
                           &w.r[1], &w.u[1], &w.c[1], &w.rh[1], &w.h[1], &w.o[2]};
    for (int i = 0; i < DXT_N; ++i) *slot[i] = cv.f(R * DX_W);
    w.tape256 = w.pz[0]; w.tstride = R * DX_W;
    w.pz[1] = cv.f(R * (hp.dec_prenet[1] + S));
    w.xcA = cv.f(R * As);
    for (int i = 0; i < L; ++i) w.xc[i] = cv.f(R * Hd);
  } else {
    for (int i = 0; i < hp.dec_prenet_n; ++i) w.pz[i] = cv.f(R * (hp.dec_prenet[i] + (i == hp.dec_prenet_n - 1 ? S : 0)));
    w.hA = cv.f(R * As); w.rA = cv.f(R * As); w.uA = cv.f(R * As); w.cA = cv.f(R * As); w.rhA = cv.f(R * As); w.xcA = cv.f(R * As);
    for (int i = 0; i <= L; ++i) w.o[i] = cv.f(R * Hd);
    for (int i = 0; i < L; ++i) { w.h[i] = cv.f(R * Hd); w.r[i] = cv.f(R * Hd); w.u[i] = cv.f(R * Hd); w.c[i] = cv.f(R * Hd); w.rh[i] = cv.f(R * Hd); w.xc[i] = cv.f(R * Hd); }
  }
  w.alpha = cv.f((size_t)B * (n + 1) * T_in); w.alpha0 = cv.f((size_t)B * T_in);   // slot 0 = initial alignments, slot t+1 = step t
  { const size_t xb = (size_t)DX_NGROUP * std::max((size_t)dx_xlayout(8, T_in).total, dbx_xbuf_granules(T_in)) * sizeof(unsigned long long);
    w.xbuf = (unsigned long long*)cv.raw(xb); w.dxctl = (unsigned*)cv.raw(256);
    w.rowbias = cv.f(is_simple(m) ? (size_t)B * DXRB_N * DX_W : 1); }
  w.nz = cv.i((size_t)n * B);
  w.dkeys = cv.f((size_t)B * T_in * A); w.dvalues = cv.f((size_t)B * T_in * D); w.dv_acc = cv.f((size_t)B * A); w.dsb_acc = cv.f(B);
  w.dalpha = cv.f((size_t)B * T_in); w.dctx = cv.f((size_t)B * D); w.dctx_t = cv.f((size_t)B * D); w.dhA = cv.f((size_t)B * As);
  for (int i = 0; i < L; ++i) w.dh[i] = cv.f((size_t)B * Hd);
  const int Wmax = std::max(std::max(As, Hd), D);
  w.dht = cv.f((size_t)B * Wmax); w.dhp = cv.f((size_t)B * Wmax);
  const int W2 = std::max(std::max(2 * Hd, As + D), std::max(hp.dec_prenet[hp.dec_prenet_n - 1] + As, hp.num_mels + D)) + Wmax + S;
  w.tmp1 = cv.f((size_t)B * W2); w.tmp2 = cv.f((size_t)B * W2);
  for (int i = 0; i <= L; ++i) w.do_[i] = cv.f((size_t)B * Hd);
  for (int i = 0; i < L; ++i) { w.g_dgp[i] = cv.f(R * 2 * Hd); w.g_dcp[i] = cv.f(R * Hd); }
  w.g_dgpA = cv.f(R * 2 * As); w.g_dcpA = cv.f(R * As); w.g_do0 = cv.f(R * Hd); w.g_dq = cv.f(R * A); w.g_do2 = cv.f(R * Hd);
  for (int i = 0; i < hp.dec_prenet_n; ++i) w.g_dz[i] = cv.f(R * hp.dec_prenet[i]);
  w.dpz = cv.f((size_t)B * W2); w.dIn = cv.f((size_t)B * W2);
  if (!dxl) w.g_q = cv.f(R * A);
  w.g_e = cv.f(R * T_in); w.g_de = cv.f(R * T_in); w.g_dctx = cv.f(R * D);
}
struct TrainWs {
  float* pre[4]; float* dpre[4];
  CbhgTape enc, post;
  DecTape dec;
  float *teach, *mel, *linear, *dmel, *dlin, *denc, *dpost, *dmel_post, *demb, *losspart;
  SpkWs spk; float* dvec[8]; float* dspk_emb; float* dzs; int* rowidx;   // deepvoice: speaker vectors, their gradients, scratch
  float *linrv, *dlin_sum;                                               // simple: speaker term of the linear head [B, F], time-summed dlin
  float* detscr;                                                         // deterministic reductions: per-slice partial sums
  uint4* wpscr; size_t wps_uint4;                                        // bf16 planes of the weight gradients' operands (taco_wgrad_planes.h)
};
static void carve_train(Carver& cv, const taco_train* t, int B, int T_in, int n, TrainWs& w) {
  const taco_model* m = t->sm;
  const taco_hparams& hp = m->hp;
  const size_t Me = (size_t)B * T_in, Mp = (size_t)B * n * hp.reduction_factor;
  for (int i = 0; i < hp.enc_prenet_n; ++i) { w.pre[i] = cv.f(Me * hp.enc_prenet[i]); w.dpre[i] = cv.f(Me * std::max(hp.enc_prenet[i], hp.embedding_size)); }
  carve_cbhg_tape(cv, m->enc, B, T_in, w.enc);
  carve_cbhg_tape(cv, m->post, B, n * hp.reduction_factor, w.post);
  carve_dec_tape(cv, m, B, T_in, n, w.dec);
  w.teach = cv.f((size_t)B * n * hp.num_mels);
  w.mel = cv.f(Mp * hp.num_mels); w.linear = cv.f(Mp * hp.num_freq);
  w.dmel = cv.f(Mp * hp.num_mels); w.dlin = cv.f(Mp * hp.num_freq);
  w.denc = cv.f(Me * 2 * hp.enc_rnn_size);
  w.dpost = cv.f(Mp * 2 * hp.post_rnn_size); w.dmel_post = cv.f(Mp * hp.num_mels); w.demb = cv.f(Me * hp.embedding_size);
  w.losspart = (float*)cv.raw((size_t)TR_MAXBLK * 8 * sizeof(double));
  carve_spk(cv, m, B, w.spk);
  { const int dims[3] = {hp.enc_prenet[hp.enc_prenet_n - 1], hp.enc_rnn_size * 2, hp.attention_state_size};
    int dmax = 1;
    for (int i = 0; i < 3 + hp.dec_layer_num; ++i) { const int dd = i < 3 ? dims[i] : hp.dec_rnn_size; w.dvec[i] = cv.f((size_t)B * dd); dmax = std::max(dmax, dd); }
    w.dspk_emb = cv.f((size_t)B * std::max(hp.speaker_embedding_size, 1)); w.dzs = cv.f((size_t)B * dmax); w.rowidx = cv.i(B); }
  const size_t nrv = is_simple(m) ? (size_t)B * hp.num_freq : 1;
  w.linrv = cv.f(nrv); w.dlin_sum = cv.f(nrv);
  w.detscr = cv.f(t->deterministic ? DET_SCRATCH_FLOATS : 1);
  { const size_t rows = (size_t)cdiv((int)std::max(Me, Mp), 64) * 64;
    const size_t cols = (size_t)std::max(hp.enc_bank_size * hp.enc_bank_channels, hp.post_bank_size * hp.post_bank_channels) + WP_SCRATCH_COLS;
    w.wps_uint4 = t->wgrad_planes ? rows / 16 * (cols / 32) * 3 * 64 : 1;
    w.wpscr = (uint4*)cv.raw(w.wps_uint4 * sizeof(uint4)); }
}

// ---------------------------------------------------------------------------------------------------------------
// CBHG: training forward with tape, and backward
// ---------------------------------------------------------------------------------------------------------------
struct TrainCtx {
  const taco_train* t; hipStream_t st; float* P; float* G;    // flat parameters (moving statistics are updated in place) and gradients
  bool update_moving;   // BatchNorm moving averages follow this pass (UPDATE_OPS run only as a dependency of `optimize`, tacotron.py:334)
  float* p(const std::string& n) const { return P + t->poff.at(n); }
  float* g(const std::string& n) const { return G + t->poff.at(n); }
};
// batch statistics of a [M, C] activation -> mu, rstd (and the moving averages of layer `name` in the parameter buffer)
static int bn_stats(const TrainCtx& x, const float* a, int lda, int M, int C, float* mu, float* rstd, float* scratch,
                    const std::string* names, const int* cols, int nnames) {
  hipStream_t st = x.st;
  // data-parallel SyncBN: mean and centred variance are those of the global batch -- what the reference's single-device step over
  // the whole batch computes (modules.py:131) -- merged from the ranks' own (mean, centred sum) pairs
  const bool sync = x.t->sync_fn && x.t->sync_world > 1;
  const float invM = 1.0f / ((float)M * (sync ? x.t->sync_world : 1));
  const bool det = g_det.p != nullptr;      // the ordered sums replace the scratch; the atomics of the other mode need it cleared
  if (!det) HIPCHK(zero_async(scratch, (size_t)2 * C * sizeof(float), st));
  TRY(run_colsum(st, a, lda, nullptr, 0, nullptr, nullptr, scratch, nullptr, M, C, 0, det));
  hipLaunchKernelGGL(k_bn_mean, EWGRID(C), 0, st, scratch, mu, C, 1.0f / (float)M);
  TRY(run_colsum(st, a, lda, nullptr, 0, mu, nullptr, nullptr, scratch + C, M, C, 1, det));
  if (sync) {
    // ONE exchange per layer (the layer's columns, or all widths of a conv bank at once): rank means, centred sums, squared means
    // (means are exchanged as offsets from the layer's moving mean, which every rank holds identically: k_bn_sync_pack)
    float* pack = scratch + 2 * C;
    for (int i = 0, c0 = 0; i < nnames; c0 += cols[i], ++i)
      hipLaunchKernelGGL(k_bn_sync_pack, EWGRID(cols[i]), 0, st, mu, scratch + C, (const float*)x.p(names[i] + "/moving_mean"), pack, C, c0, cols[i]);
    x.t->sync_fn(x.t->sync_user, pack, 3 * C);
    for (int i = 0, c0 = 0; i < nnames; c0 += cols[i], ++i)
      hipLaunchKernelGGL(k_bn_sync_combine, EWGRID(cols[i]), 0, st, pack, (const float*)x.p(names[i] + "/moving_mean"), mu, scratch + C, C, c0, cols[i],
                         (float)M, (float)x.t->sync_world);
  }
  bool even = nnames > 1 && nnames <= BNB_MAXK;
  for (int i = 1; i < nnames; ++i) even = even && cols[i] == cols[0];
  if (even) {                          // the layers of a conv bank: one launch for all of them
    BnBank nb; memset(&nb, 0, sizeof nb); nb.Cw = cols[0];
    for (int i = 0; i < nnames; ++i) {
      nb.mov_mean[i] = x.update_moving ? x.p(names[i] + "/moving_mean") : (float*)nullptr;
      nb.mov_var[i] = x.update_moving ? x.p(names[i] + "/moving_variance") : (float*)nullptr;
    }
    hipLaunchKernelGGL(k_bn_finalize_bank, EWGRID(C), 0, st, mu, scratch + C, rstd, nb, C, invM, 1e-3f, 0.99f);
    HIPCHK(hipGetLastError());
    return 0;
  }
  int c0 = 0;
  for (int i = 0; i < nnames; ++i) {   // one BatchNorm layer per column block (conv bank) or the whole matrix
    // forward-only passes (loss fetches, the test model, a capture warm-up) leave the moving statistics alone, as the reference does
    hipLaunchKernelGGL(k_bn_finalize, EWGRID(cols[i]), 0, st, mu + c0, scratch + C + c0, rstd + c0,
                       x.update_moving ? x.p(names[i] + "/moving_mean") : (float*)nullptr,
                       x.update_moving ? x.p(names[i] + "/moving_variance") : (float*)nullptr, cols[i], invM, 1e-3f, 0.99f);
    c0 += cols[i];
  }
  HIPCHK(hipGetLastError());
  return 0;
}
static int cbhg_forward_train(const TrainCtx& x, const Cbhg& c, const CbhgT& ct, const std::string& sc, const float* in, int B, int T,
                              const int* lengths, const CbhgTape& w, const float* before_highway = nullptr, const float* init_state = nullptr) {
  const taco_model* m = x.t->sm; hipStream_t st = x.st;
  const int M = B * T, KC = c.K * c.C;
  { GemmCall g; g.x = in; g.ldx = c.in_dim; g.M = M; g.T = T; g.act = ACT_RELU; g.out = w.bank_a; g.ldo = KC;
    TRY(run_gemm(m, st, ct.bank_f.data(), c.K, false, g)); }
  { std::vector<std::string> names; std::vector<int> cols;
    // column block k-1 of the concatenation belongs to conv1d_k (modules.py:35-44)
    for (int k = 1; k <= c.K; ++k) { names.push_back(sc + "/conv_bank/conv1d_" + std::to_string(k)); cols.push_back(c.C); }
    TRY(bn_stats(x, w.bank_a, KC, M, KC, w.bank_mu, w.bank_rs, w.stat, names.data(), cols.data(), c.K));
    if (bank_pool_fused(c) && al16h(w.bank_a) && al16h(w.pool)) {      // BatchNorm of all widths + the max-pool in ONE pass: the BatchNorm output is never stored (k_bn_pool_bank_v4)
      BnBank nb; memset(&nb, 0, sizeof nb); nb.Cw = c.C;
      for (int k = 1; k <= c.K; ++k) { nb.gamma[k - 1] = x.p(names[k - 1] + "/gamma"); nb.beta[k - 1] = x.p(names[k - 1] + "/beta"); }
      hipLaunchKernelGGL(k_bn_pool_bank_v4, EWGRID((size_t)M * KC / 4), 0, st, (const float*)w.bank_a, KC, (const float*)w.bank_mu, (const float*)w.bank_rs, nb, w.pool, KC, M, T, KC, c.maxpool);
    } else {
    for (int k = 1; k <= c.K; ++k) {
      const int c0 = (k - 1) * c.C;
      hipLaunchKernelGGL(k_bn_apply, EWGRID((size_t)M * c.C), 0, st, w.bank_a + c0, KC, w.bank_mu + c0, w.bank_rs + c0,
                         x.p(names[k - 1] + "/gamma"), x.p(names[k - 1] + "/beta"), w.bank_y + c0, KC, M, c.C);
    }
    hipLaunchKernelGGL(k_maxpool_fwd, EWGRID((size_t)M * KC), 0, st, w.bank_y, w.pool, M, T, KC, c.maxpool);
    } }
  HIPCHK(hipGetLastError());
  const float* cur = w.pool; int curd = KC;
  for (int i = 0; i < c.nproj; ++i) {
    const std::string n = sc + "/proj_" + std::to_string(i + 1);
    const int N = c.proj_dim[i];
    GemmCall p; p.x = cur; p.ldx = curd; p.M = M; p.T = T; p.act = (i + 1 == c.nproj) ? ACT_NONE : ACT_RELU; p.out = w.pa[i]; p.ldo = N;
    TRY(run_gemm(m, st, &ct.proj_f[i], 1, false, p));
    TRY(bn_stats(x, w.pa[i], N, M, N, w.pmu[i], w.prs[i], w.stat, &n, &N, 1));
    if ((N & 3) == 0 && al16h(w.pa[i]) && al16h(w.py[i]))
      hipLaunchKernelGGL(k_bn_apply_v4, EWGRID((size_t)M * N / 4), 0, st, (const float*)w.pa[i], N, (const float*)w.pmu[i], (const float*)w.prs[i], (const float*)x.p(n + "/gamma"), (const float*)x.p(n + "/beta"), w.py[i], N, M, N);
    else hipLaunchKernelGGL(k_bn_apply, EWGRID((size_t)M * N), 0, st, w.pa[i], N, w.pmu[i], w.prs[i], x.p(n + "/gamma"), x.p(n + "/beta"), w.py[i], N, M, N);
    cur = w.py[i]; curd = N;
  }
  // residual (modules.py:62-69)
  hipLaunchKernelGGL(k_copy2d, EWGRID((size_t)M * c.in_dim), 0, st, cur, c.in_dim, w.res, c.in_dim, M, c.in_dim);
  hipLaunchKernelGGL(k_add2d, EWGRID((size_t)M * c.in_dim), 0, st, w.res, c.in_dim, in, c.in_dim, M, c.in_dim);
  if (before_highway) hipLaunchKernelGGL(k_add_rowvec, EWGRID((size_t)M * c.in_dim), 0, st, w.res, before_highway, M, T, c.in_dim);   // modules.py:66-69
  HIPCHK(hipGetLastError());
  if (c.has_dense) { GemmCall d; d.x = w.res; d.ldx = c.in_dim; d.M = M; d.out = w.hx[0]; d.ldo = c.rnn; TRY(run_gemm(m, st, &c.dense, 1, false, d)); }
  else hipLaunchKernelGGL(k_copy2d, EWGRID((size_t)M * c.rnn), 0, st, w.res, c.rnn, w.hx[0], c.rnn, M, c.rnn);
  for (int i = 0; i < c.depth; ++i) {
    GemmCall h; h.x = w.hx[i]; h.ldx = c.rnn; h.M = M; h.out = w.hx[i + 1]; h.ldo = c.rnn; h.aux0 = w.hH[i]; h.aux1 = w.hT[i];
    TRY(run_gemm(m, st, &c.hw[i], 1, true, h));
  }
  const int H = c.rnn;
  { GemmCall xp; xp.x = w.hx[c.depth]; xp.ldx = H; xp.M = M; xp.T = T; xp.out = w.xproj; xp.ldo = 6 * H; xp.rev_len = lengths; xp.rev_col0 = 3 * H;
    TRY(run_gemm(m, st, &c.xproj, 1, false, xp)); }
  if (lengths) HIPCHK(zero_async(w.gsave, (size_t)M * 6 * H * sizeof(float), st));      // (without lengths -- the post-net -- every step of every row is active and written)
  if (const int upw = oct_upw(m, c, B, T))     // the whole-chip scans of inference with the gate tape (k_bigru_oct<UPW, true> / k_bigru_duo<RG, true>)
    return oct_launch(m, st, c, upw, B, T, w.xproj, lengths, init_state, w.out, w.gsave, w.gxbuf, w.gxctl);
  if (duo_usable(m, c, B, T))
    return duo_launch(m, st, c, B, T, w.xproj, lengths, init_state, w.out, w.gsave, w.gxbuf, w.gxctl);
  if (H == 256 || H == 128) {     // recurrent weights resident on the CU (k_bigru_res; H = 128: k_bigru_quad, as inference runs it), gates saved for the backward scan
    BigruSArgs a; memset(&a, 0, sizeof a);
    a.xproj = w.xproj; a.g2_0 = (const float2*)AP(m, c.res_g2[0]); a.g2_1 = (const float2*)AP(m, c.res_g2[1]);
    a.c1_0 = AP(m, c.raw_ch[0]); a.c1_1 = AP(m, c.raw_ch[1]); a.lengths = lengths; a.out = w.out; a.gsave = w.gsave; a.B = B; a.T = T;
    a.h0 = init_state;
    if (H == 256) hipLaunchKernelGGL((k_bigru_res<256, 64, 24, 1, true>), dim3(2 * B), dim3(512), bigru_res_lds(256, 24, 1), st, a);
    else if (m->persist == 1 && x.t->resident_bwd_scan) hipLaunchKernelGGL(k_bigru_quad<true>, dim3(2 * B), dim3(512), 0, st, a);      // (the A/B engine keeps k_bigru_res)
    else hipLaunchKernelGGL((k_bigru_res<128, 32, 0, 1, true>), dim3(2 * B), dim3(512), bigru_res_lds(128, 0, 1), st, a);
  } else {
    int R = 0; size_t lds = 0;
    if (!bigru_rows_cfg(B, H, &R, &lds)) return fail(TACO_ERR_UNSUPPORTED, "rnn size %d does not fit the row-parallel BiGRU kernel", H);
    BigruRArgs a; memset(&a, 0, sizeof a);
    a.xproj = w.xproj; a.wg0 = AP(m, c.raw_gh[0]); a.wg1 = AP(m, c.raw_gh[1]); a.wc0 = AP(m, c.raw_ch[0]); a.wc1 = AP(m, c.raw_ch[1]);
    a.lengths = lengths; a.out = w.out; a.gsave = w.gsave; a.B = B; a.T = T; a.H = H; a.h0 = init_state;
    if (R == 2) hipLaunchKernelGGL((k_bigru_rows<2, true>), dim3(2 * cdiv(B, R)), dim3(RP_NT), lds, st, a);
    else hipLaunchKernelGGL((k_bigru_rows<1, true>), dim3(2 * cdiv(B, R)), dim3(RP_NT), lds, st, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// one conv1d+act+BN(train) layer backward: dy (grad of BN output) -> weight/bias/gamma/beta grads and dz (pre-activation grad).
// Two halves so that layers whose dy exist at the same time (the widths of a conv bank) share ONE SyncBN exchange:
// _sums: per-channel sum dy, sum dy*xhat into the beta / gamma gradients (+ a copy at sync_scratch[c0], sync_scratch[Ctot + c0]);
// _apply: dz and the bias gradient from those sums (the global ones when synchronised).
static int conv_bn_backward_sums(const TrainCtx& x, const std::string& name, const float* a, int lda, const float* dy, int lddy, const float* mu,
                                 const float* rstd, int M, int C, float* sync_scratch, int c0, int Ctot) {
  hipStream_t st = x.st;
  TRY(run_colsum(st, a, lda, dy, lddy, mu, rstd, x.g(name + "/beta"), x.g(name + "/gamma"), M, C, 2));
  if (x.t->sync_fn && x.t->sync_world > 1) {
    TRY(wg_flush());                  // (the sums are copied right here)
    // the gradient buffers keep this rank's own sums (the flat all-reduce after backward averages them like every other gradient)
    HIPCHK(hipMemcpyAsync(sync_scratch + c0, x.g(name + "/beta"), (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(sync_scratch + Ctot + c0, x.g(name + "/gamma"), (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  return 0;
}
static void conv_bn_backward_exchange(const TrainCtx& x, float* sync_scratch, int Ctot) {
  if (x.t->sync_fn && x.t->sync_world > 1) x.t->sync_fn(x.t->sync_user, sync_scratch, 2 * Ctot);
}
static int conv_bn_backward_apply(const TrainCtx& x, const std::string& name, const float* a, int lda, const float* dy, int lddy, const float* mu,
                                  const float* rstd, bool relu, float* dz, int lddz, int M, int C, const float* sync_scratch, int c0, int Ctot,
                                  bool dz_done = false) {      // dz_done: the caller formed dz for all layers of a bank in one launch (bank_bn_bwd)
  hipStream_t st = x.st;
  const bool sync = x.t->sync_fn && x.t->sync_world > 1;
  const float* sdy = sync ? sync_scratch + c0 : x.g(name + "/beta");
  const float* sdyxh = sync ? sync_scratch + Ctot + c0 : x.g(name + "/gamma");
  const float invM = 1.0f / ((float)M * (sync ? x.t->sync_world : 1));
  if (dz_done) {}
  else if ((C & 3) == 0 && (lda & 3) == 0 && (lddy & 3) == 0 && (lddz & 3) == 0 && al16h(a) && al16h(dy) && al16h(dz))
    hipLaunchKernelGGL(k_bn_bwd_v4, EWGRID((size_t)M * C / 4), 0, st, a, lda, dy, lddy, mu, rstd, (const float*)x.p(name + "/gamma"), sdy,
                       sdyxh, relu ? 1 : 0, dz, lddz, M, C, invM);
  else
  hipLaunchKernelGGL(k_bn_bwd, EWGRID((size_t)M * C), 0, st, a, lda, dy, lddy, mu, rstd, x.p(name + "/gamma"), sdy,
                     sdyxh, relu ? 1 : 0, dz, lddz, M, C, invM);
  TRY(run_colsum(st, dz, lddz, nullptr, 0, nullptr, nullptr, x.g(name + "/bias"), nullptr, M, C, 0));
  HIPCHK(hipGetLastError());
  return 0;
}
static int conv_bn_backward(const TrainCtx& x, const std::string& name, const float* a, int lda, const float* dy, int lddy, const float* mu,
                            const float* rstd, bool relu, float* dz, int lddz, int M, int C, float* sync_scratch) {
  TRY(conv_bn_backward_sums(x, name, a, lda, dy, lddy, mu, rstd, M, C, sync_scratch, 0, C));
  TRY(wg_flush());                    // (inside a batching region the sums above are still queued)
  conv_bn_backward_exchange(x, sync_scratch, C);
  return conv_bn_backward_apply(x, name, a, lda, dy, lddy, mu, rstd, relu, dz, lddz, M, C, sync_scratch, 0, C);
}
// dout [M, 2*rnn] -> din [M, in_dim]
static int cbhg_backward(const TrainCtx& x, const Cbhg& c, const CbhgT& ct, const std::string& sc, const float* in, const int* in_gather,
                         int B, int T, const int* lengths, const float* dout, float* din, const CbhgTape& w,
                         const float* h0 = nullptr, float* dh0 = nullptr, float* d_before = nullptr, int* rowidx = nullptr) {
  (void)in_gather;
  const taco_model* m = x.t->sm; hipStream_t st = x.st;
  const int M = B * T, KC = c.K * c.C, H = c.rnn, I = c.rnn;
  // ---- BiGRU ----
  // (the backward scans write the steps inside a row's length only; k_bigru_duo_bwd without lengths -- the post-net -- writes every element)
  const bool duo_bwd = duo_usable(m, c, B, T) && c.gb_pack;
  ZeroBatch zb(st);
  if (!(duo_bwd && !lengths)) {
    HIPCHK(zb.add(w.dg, (size_t)M * 6 * H * sizeof(float)));
    HIPCHK(zb.add(w.rh, (size_t)M * 2 * H * sizeof(float)));
  }
  if (duo_bwd) HIPCHK(zb.add(w.gxbuf, (size_t)((char*)w.gxctl - (char*)w.gxbuf) + 256));
  HIPCHK(zb.run());
  if (duo_bwd && oct_bwd_usable(m, c, B, T)) {
    // one row per cluster of 8 CUs, as the forward scan of these shapes (k_bigru_oct_bwd)
    TRY(oct_bwd_launch(m, st, c, B, T, dout, w.out, w.gsave, h0, lengths, w.dg, w.rh, dh0, w.gxbuf, w.gxctl));
  } else if (duo_bwd) {
    // both directions of RG rows per group of 32 CUs, the directions software-pipelined against each other (k_bigru_duo_bwd)
    ChipTurn turn(m->device, st);
    GbArgs a; memset(&a, 0, sizeof a);
    a.wpack = AP(m, c.gb_pack); a.dout = dout; a.out = w.out; a.gsave = w.gsave; a.h0 = h0; a.lengths = lengths; a.dg = w.dg; a.rh = w.rh; a.dh0 = dh0;
    a.xbuf = w.gxbuf; a.ctl = w.gxctl; a.err = m->d_err; a.B = B; a.T = T; a.force_wt = m->dx_mode == 2 ? 1 : 0;
    int RG = 1;
    while (RG * DX_NGROUP < B) RG *= 2;
    const size_t lds = std::max(gb_lds_floats(RG) * sizeof(float), (size_t)96 * 1024);      // one workgroup per CU
    const dim3 grid(DX_NGROUP * GD_MEMBERS), blk(512);
    switch (RG) {
      case 1: hipLaunchKernelGGL((k_bigru_duo_bwd<1>), grid, blk, lds, st, a); break;
      case 2: hipLaunchKernelGGL((k_bigru_duo_bwd<2>), grid, blk, lds, st, a); break;
      case 4: hipLaunchKernelGGL((k_bigru_duo_bwd<4>), grid, blk, lds, st, a); break;
      default: hipLaunchKernelGGL((k_bigru_duo_bwd<8>), grid, blk, lds, st, a); break;
    }
    HIPCHK(hipGetLastError());
  } else if (H == 128 && x.t->resident_bwd_scan) {
    // recurrent kernels resident in registers, one workgroup per (direction, row) (k_bigru_resb): the encoder at the reference width
    BigruQArgs a; memset(&a, 0, sizeof a);
    a.dout = dout; a.out = w.out; a.gsave = w.gsave; a.gh0 = AP(m, c.raw_gh[0]); a.gh1 = AP(m, c.raw_gh[1]); a.ch0 = AP(m, c.raw_ch[0]); a.ch1 = AP(m, c.raw_ch[1]);
    a.lengths = lengths; a.dg = w.dg; a.rh = w.rh; a.h0 = h0; a.dh0 = dh0; a.B = B; a.T = T;
    hipLaunchKernelGGL(k_bigru_resb<128>, dim3(2 * B), dim3(512), 0, st, a);
    HIPCHK(hipGetLastError());
  } else {
    int R = (B >= 2 && 2 * H <= RP_NT) ? 2 : 1;
    if ((size_t)R * H > RP_NT || (H % 4)) return fail(TACO_ERR_UNSUPPORTED, "rnn size %d does not fit the BiGRU backward kernel", H);
    const size_t lds = ((size_t)4 * R * H + (size_t)RP_NT * R * 4 + 64) * sizeof(float);
    BigruBArgs a; memset(&a, 0, sizeof a);
    a.dout = dout; a.out = w.out; a.gsave = w.gsave; a.wgT0 = AP(m, ct.ghT[0]); a.wgT1 = AP(m, ct.ghT[1]); a.wcT0 = AP(m, ct.chT[0]); a.wcT1 = AP(m, ct.chT[1]);
    a.lengths = lengths; a.dg = w.dg; a.rh = w.rh; a.B = B; a.T = T; a.H = H; a.h0 = h0; a.dh0 = dh0;
    if (R == 2) hipLaunchKernelGGL(k_bigru_rows_bwd<2>, dim3(2 * cdiv(B, R)), dim3(RP_NT), lds, st, a);
    else hipLaunchKernelGGL(k_bigru_rows_bwd<1>, dim3(2 * cdiv(B, R)), dim3(RP_NT), lds, st, a);
    HIPCHK(hipGetLastError());
  }
  const float* hlast = w.hx[c.depth];
  WgRegion wgr(st);          // small weight gradients of this CBHG join group launches; flushed wherever an operand is about to be reused
  for (int dir = 0; dir < 2; ++dir) {
    const std::string n = sc + "/bigru/" + (dir ? "bw" : "fw");
    const float* dgg = w.dg + dir * 3 * H;          // gates columns (r|u), candidate at +2H
    float* Gg = x.g(n + "/gates/kernel"); float* Gc = x.g(n + "/candidate/kernel");
    TRY(run_wgrad(st, hlast, nullptr, I, dgg, 6 * H, Gg, 2 * H, M, T, I, 2 * H));                                   // x rows of gates
    TRY(run_wgrad(st, hlast, nullptr, I, dgg + 2 * H, 6 * H, Gc, H, M, T, I, H));                                    // x rows of candidate
    // h rows: state before the step = output one step earlier in the direction's own time (zero at the sequence start / past the length)
    TRY(run_wgrad(st, w.out + dir * H, nullptr, 2 * H, dgg, 6 * H, Gg + (size_t)I * 2 * H, 2 * H, M, T, H, 2 * H, 1, dir ? -1 : 1));
    TRY(run_wgrad(st, w.rh + dir * H, nullptr, 2 * H, dgg + 2 * H, 6 * H, Gc + (size_t)I * H, H, M, T, H, H));
    if (h0) {   // first step of the direction: the state before it is the initial state, not a row of the output tape
      if (dir == 0) TRY(run_wgrad(st, h0, nullptr, 2 * H, dgg, T * 6 * H, Gg + (size_t)I * 2 * H, 2 * H, B, 0, H, 2 * H));
      else {
        hipLaunchKernelGGL(k_last_row_index, EWGRID(B), 0, st, lengths, rowidx, B, T);
        TRY(run_wgrad(st, h0 + H, nullptr, 2 * H, dgg, 6 * H, Gg + (size_t)I * 2 * H, 2 * H, B, 0, H, 2 * H, 1, 0, rowidx));
      }
    }
    TRY(run_colsum(st, dgg, 6 * H, nullptr, 0, nullptr, nullptr, x.g(n + "/gates/bias"), nullptr, M, 2 * H, 0));
    TRY(run_colsum(st, dgg + 2 * H, 6 * H, nullptr, 0, nullptr, nullptr, x.g(n + "/candidate/bias"), nullptr, M, H, 0));
  }
  TRY(wg_flush());
  float* dcur = w.d0; float* dalt = w.d1;
  TRY(run_dgrad(m, st, ct.xproj_d, w.dg, 6 * H, M, T, dcur, I));
  // ---- highways ----
  for (int i = c.depth - 1; i >= 0; --i) {
    const std::string n = sc + "/highway_" + std::to_string(i + 1);
    if ((H & 3) == 0 && al16h(dcur) && al16h(w.hx[i]) && al16h(w.hH[i]) && al16h(w.hT[i]) && al16h(w.dcat) && al16h(dalt))
      hipLaunchKernelGGL(k_highway_bwd_v4, EWGRID((size_t)M * H / 4), 0, st, (const float*)dcur, (const float*)w.hx[i], (const float*)w.hH[i], (const float*)w.hT[i], w.dcat, dalt, M, H);
    else hipLaunchKernelGGL(k_highway_bwd, EWGRID((size_t)M * H), 0, st, dcur, w.hx[i], w.hH[i], w.hT[i], w.dcat, dalt, M, H);
    HIPCHK(hipGetLastError());
    TRY(run_wgrad(st, w.hx[i], nullptr, H, w.dcat, 2 * H, x.g(n + "/H/kernel"), H, M, 0, H, H));
    TRY(run_wgrad(st, w.hx[i], nullptr, H, w.dcat + H, 2 * H, x.g(n + "/T/kernel"), H, M, 0, H, H));
    TRY(run_colsum(st, w.dcat, 2 * H, nullptr, 0, nullptr, nullptr, x.g(n + "/H/bias"), nullptr, M, H, 0));
    TRY(run_colsum(st, w.dcat + H, 2 * H, nullptr, 0, nullptr, nullptr, x.g(n + "/T/bias"), nullptr, M, H, 0));
    TRY(run_dgrad(m, st, ct.hw_d[i], w.dcat, 2 * H, M, 0, dalt, H, dalt, H));   // += direct path
    TRY(wg_flush());                                                            // (dcat is the next layer's scratch)
    std::swap(dcur, dalt);
  }
  // ---- dense (post-net) ----
  if (c.has_dense) {
    TRY(run_wgrad(st, w.res, nullptr, c.in_dim, dcur, H, x.g(sc + "/dense/kernel"), H, M, 0, c.in_dim, H));
    TRY(run_colsum(st, dcur, H, nullptr, 0, nullptr, nullptr, x.g(sc + "/dense/bias"), nullptr, M, H, 0));
    TRY(run_dgrad(m, st, ct.dense_d, dcur, H, M, 0, dalt, c.in_dim));
    TRY(wg_flush());
    std::swap(dcur, dalt);
  }
  // dcur = gradient of (proj_last + x (+ before_highway)): keep a copy for the residual path
  if (d_before) { hipLaunchKernelGGL(k_time_sum, EWGRID((size_t)B * c.in_dim), 0, st, dcur, d_before, B, T, c.in_dim); HIPCHK(hipGetLastError()); }
  float* dres = din;
  hipLaunchKernelGGL(k_copy2d, EWGRID((size_t)M * c.in_dim), 0, st, dcur, c.in_dim, dres, c.in_dim, M, c.in_dim);
  HIPCHK(hipGetLastError());
  // ---- projections (last first) ----
  for (int i = c.nproj - 1; i >= 0; --i) {
    const std::string n = sc + "/proj_" + std::to_string(i + 1);
    const int N = c.proj_dim[i];
    const float* xin = (i == 0) ? w.pool : w.py[i - 1]; const int xd = (i == 0) ? KC : c.proj_dim[i - 1];
    TRY(conv_bn_backward(x, n, w.pa[i], N, dcur, N, w.pmu[i], w.prs[i], i + 1 != c.nproj, dalt, N, M, N, w.stat));
    TRY(run_wgrad(st, xin, nullptr, xd, dalt, N, x.g(n + "/kernel"), N, M, T, xd, N, c.pw, (c.pw - 1) / 2));
    float* dnext = (i == 0) ? w.dbig0 : dcur;
    TRY(run_dgrad(m, st, ct.proj_d[i], dalt, N, M, T, dnext, xd));
    TRY(wg_flush());                                                            // (dalt is rewritten by the next projection)
    if (i > 0) { /* dnext == dcur already holds the gradient of py[i-1] */ }
  }
  // ---- maxpool + conv bank ----
  if (bank_pool_fused(c) && al16h(w.bank_a) && al16h(w.pool)) {      // (the forward's test: the pool's input is recomputed from the bank's activations)
    BnBank nb; memset(&nb, 0, sizeof nb); nb.Cw = c.C;
    for (int k = 1; k <= c.K; ++k) { const std::string n = sc + "/conv_bank/conv1d_" + std::to_string(k); nb.gamma[k - 1] = x.p(n + "/gamma"); nb.beta[k - 1] = x.p(n + "/beta"); }
    hipLaunchKernelGGL(k_maxpool_bwd_bn_v4, EWGRID((size_t)M * KC / 4), 0, st, (const float*)w.bank_a, KC, (const float*)w.bank_mu, (const float*)w.bank_rs, nb, (const float*)w.dbig0, w.dbig1, M, T, KC, c.maxpool);
  } else hipLaunchKernelGGL(k_maxpool_bwd, EWGRID((size_t)M * KC), 0, st, w.bank_y, w.dbig0, w.dbig1, M, T, KC, c.maxpool);
  HIPCHK(hipGetLastError());
  for (size_t bi = 0; bi < c.bank.size(); ++bi) {     // the bank's BatchNorm sums of all widths, then one exchange for all of them
    const int k = c.bank[bi].kw, c0 = (k - 1) * c.C;
    const std::string n = sc + "/conv_bank/conv1d_" + std::to_string(k);
    TRY(conv_bn_backward_sums(x, n, w.bank_a + c0, KC, w.dbig1 + c0, KC, w.bank_mu + c0, w.bank_rs + c0, M, c.C, w.stat, c0, KC));
  }
  TRY(wg_flush());                    // the sums of all widths: one group launch
  conv_bn_backward_exchange(x, w.stat, KC);
  // dz of ALL widths in one launch (their BatchNorm sums are complete: the flush above); the layers' own vectors through a pointer table
  bool bank_dz = false;
  if ((int)c.bank.size() == c.K && c.K <= BNB_MAXK && (c.C & 3) == 0 && al16h(w.bank_a) && al16h(w.dbig0) && al16h(w.dbig1)) {
    const bool sync = x.t->sync_fn && x.t->sync_world > 1;
    BnBank nb; memset(&nb, 0, sizeof nb); nb.Cw = c.C;
    bool all = true;
    for (size_t bi = 0; bi < c.bank.size(); ++bi) {
      const int k = c.bank[bi].kw, c0 = (k - 1) * c.C;
      if (k < 1 || k > c.K || nb.gamma[k - 1]) { all = false; break; }
      const std::string n = sc + "/conv_bank/conv1d_" + std::to_string(k);
      nb.gamma[k - 1] = x.p(n + "/gamma");
      nb.sdy[k - 1] = sync ? w.stat + c0 : x.g(n + "/beta");
      nb.sdyxh[k - 1] = sync ? w.stat + KC + c0 : x.g(n + "/gamma");
    }
    if (all) {
      const float invM = 1.0f / ((float)M * (sync ? x.t->sync_world : 1));
      hipLaunchKernelGGL(k_bn_bwd_bank_v4, EWGRID((size_t)M * KC / 4), 0, st, (const float*)w.bank_a, KC, (const float*)w.dbig1, KC, (const float*)w.bank_mu, (const float*)w.bank_rs, nb, 1,
                         w.dbig0, KC, M, KC, invM);
      HIPCHK(hipGetLastError());
      bank_dz = true;
    }
  }
  bool bank_wg = false;                 // every width's kernel gradient from ONE conversion of dz and one product launch (taco_wgrad_planes.h)
  if (bank_dz && g_wgrad_bf3 && !in_gather && (int)c.bank.size() <= WP_MAXW) {
    float* dwk[WP_MAXW];
    for (size_t bi = 0; bi < c.bank.size(); ++bi) dwk[c.bank[bi].kw - 1] = x.g(sc + "/conv_bank/conv1d_" + std::to_string(c.bank[bi].kw) + "/kernel");
    TRY(run_wgrad_bank_planes(st, in, c.in_dim, w.dbig0, KC, dwk, M, T, c.in_dim, c.C, c.K, bank_wg));
  }
  for (size_t bi = 0; bi < c.bank.size(); ++bi) {
    const int k = c.bank[bi].kw, c0 = (k - 1) * c.C;
    const std::string n = sc + "/conv_bank/conv1d_" + std::to_string(k);
    TRY(conv_bn_backward_apply(x, n, w.bank_a + c0, KC, w.dbig1 + c0, KC, w.bank_mu + c0, w.bank_rs + c0, true, w.dbig0 + c0, KC, M, c.C, w.stat, c0, KC, bank_dz));
    if (!bank_wg) TRY(run_wgrad(st, in, in_gather, c.in_dim, w.dbig0 + c0, KC, x.g(n + "/kernel"), c.C, M, T, c.in_dim, c.C, k, (k - 1) / 2));
    TRY(run_dgrad(m, st, ct.bank_d[bi], w.dbig0 + c0, KC, M, T, din, c.in_dim, din, c.in_dim));   // accumulates onto the residual path
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// decoder: teacher-forced training forward with tape (TacoTrainingHelper, helpers.py:35-67), and BPTT
// ---------------------------------------------------------------------------------------------------------------
static int gru_cell_train(const taco_model* m, hipStream_t st, const GruDec& g, int B, const float* xin, int ldx, const float* hprev, int ldh,
                          float* hnew, float* rh, float* u, float* xc, float* r, float* c, int ld, float* out_res) {
  SkJob ja = sk_base(m, g.gx, xin, ldx, g.I, hprev, ldh);
  ja.H = g.H; ja.e0 = hprev; ja.lde0 = ldh; ja.o0 = rh; ja.ldo0 = ld; ja.o1 = u; ja.ldo1 = ld; ja.o2 = xc; ja.ldo2 = ld; ja.o3 = r; ja.ldo3 = ld;
  TRY(run_skinny(st, B, &ja, 1, EPI_GRU_GATES));
  SkJob jb = sk_base(m, g.ch, rh, ld, g.H, nullptr, 0);
  jb.H = g.H; jb.e0 = hprev; jb.lde0 = ldh; jb.e1 = xc; jb.lde1 = ld; jb.e2 = u; jb.lde2 = ld; jb.o0 = hnew; jb.ldo0 = ld; jb.o3 = c; jb.ldo3 = ld;
  if (out_res) { jb.e3 = xin; jb.lde3 = ldx; jb.o1 = out_res; jb.ldo1 = ld; }
  TRY(run_skinny(st, B, &jb, 1, EPI_GRU_CAND));
  return 0;
}
static int decoder_forward_train(const TrainCtx& x, const float* enc_out, int B, int T_in, int n, const float* teach, float* mel,
                                 float* align_hist, const DecTape& w, bool feed_back, const float* att_init = nullptr,
                                 const float* const* dec_init = nullptr, const float* spk_emb = nullptr) {
  const taco_model* m = x.t->sm; hipStream_t st = x.st;
  const taco_hparams& hp = m->hp;
  const int D = 2 * hp.enc_rnn_size, As = hp.attention_state_size, Hd = hp.dec_rnn_size, A = hp.attention_size;
  const int Mm = hp.num_mels, rM = Mm * hp.reduction_factor, L = hp.dec_layer_num, np = hp.dec_prenet_n;
  if (T_in > ATT_MAXT) return fail(TACO_ERR_UNSUPPORTED, "T_in %d > %d", T_in, ATT_MAXT);
  if (!((A % 4 == 0) && A <= ATT_MAXT && A / 4 <= 64 * ATT_NW) || (D % 4) || A > 1024 || D > 1024 || As > 1024)
    return fail(TACO_ERR_UNSUPPORTED, "attention sizes not supported by the training kernels");
  { GemmCall g; g.x = enc_out; g.ldx = D; g.M = B * T_in; g.out = w.keys; g.ldo = A; TRY(run_gemm(m, st, &m->memory_layer, 1, false, g)); }
  const int Wz = std::max(std::max(Mm, As), std::max(Hd, D));
  hipLaunchKernelGGL(k_copy2d, EWGRID((size_t)B * Wz), 0, st, (const float*)nullptr, 0, w.zero, Wz, B, Wz);
  hipLaunchKernelGGL(k_init_align, EWGRID((size_t)B * T_in), 0, st, w.alpha0, B, T_in, hp.attention_type == 2 ? 1 : 0);
  const int ldal = (n + 1) * T_in;
  hipLaunchKernelGGL(k_copy2d, EWGRID((size_t)B * T_in), 0, st, w.alpha0, T_in, w.alpha, ldal, B, T_in);
  HIPCHK(hipGetLastError());
  HIPCHK(zero_async(w.nz, (size_t)n * B * sizeof(int), st));
  const int Pl = hp.dec_prenet[np - 1];
  const int S = simple_S(m), Dc = D + S, Pz = Pl + S;      // 'simple': speaker embedding parked behind ctx and behind the last prenet output
  if (S) {
    if (!spk_emb) return fail(TACO_ERR_ARG, "speaker embedding missing");
    hipLaunchKernelGGL(k_tile_rows, EWGRID((size_t)B * n * S), 0, st, spk_emb, w.ctx, Dc, D, B, n, S);
    hipLaunchKernelGGL(k_tile_rows, EWGRID((size_t)B * n * S), 0, st, spk_emb, w.pz[np - 1], Pz, Pl, B, n, S);
    HIPCHK(hipGetLastError());
  }
  if (w.tape256 && hp.dec_layer_num == 2 && hp.num_mels <= DX_P2 && (size_t)DXT_N * w.tstride < (1u << 31) && dx_usable(m, B, T_in, nullptr, teach)) {
    // the whole teacher-forced loop as ONE persistent launch that also writes the tape (k_decoder_xcd<RG, true>, taco_decoder_xcd.h);
    // rnn_decoder_test_mode (feed_back): the same launch, the step's own last frame exchanged in place of the teacher's
    DxArgs ta; memset(&ta, 0, sizeof ta);
    ta.teacher = feed_back ? nullptr : teach; ta.own_fb = feed_back ? 1 : 0; ta.tape = w.tape256; ta.tstride = w.tstride; ta.tp_p2 = w.pz[np - 1]; ta.ld_p2 = Pz; ta.tp_ctx = w.ctx; ta.ld_ctx = Dc;
    ta.tp_e = w.g_e; ta.tp_alpha = w.alpha;
    return dx_launch(m, st, enc_out, nullptr, spk_emb, B, T_in, n, nullptr, mel, align_hist, nullptr, 0, w.keys, w.nz, w.xbuf, w.dxctl, w.rowbias,
                     att_init, dec_init ? dec_init[0] : nullptr, dec_init ? dec_init[1] : nullptr, &ta);
  }
  for (int t = 0; t < n; ++t) {
    // helpers.py:44,66,70-72: previous teacher frame; rnn_decoder_test_mode (:63-64): last of the r frames the decoder just emitted
    const float* frame = (t == 0) ? w.zero : (feed_back ? mel + (size_t)(t - 1) * rM + (rM - Mm) : teach + (size_t)(t - 1) * Mm);
    const int ldf = (t == 0) ? Mm : (feed_back ? n * rM : n * Mm);
    const float* cprev = (t == 0) ? w.zero : w.ctx + (size_t)(t - 1) * Dc; const int ldcp = (t == 0) ? D : n * Dc;
    for (int i = 0; i < np; ++i) {
      const int P = hp.dec_prenet[i], Pw = (i == np - 1) ? Pz : P;      // row width of this layer's tape
      SkJob j = (i == 0) ? sk_linear(m, m->dec_prenet[0], frame, ldf, Mm, cprev, ldcp, ACT_RELU, w.pz[0] + (size_t)t * Pw, n * Pw)
                         : sk_linear(m, m->dec_prenet[i], w.pz[i - 1] + (size_t)t * hp.dec_prenet[i - 1], n * hp.dec_prenet[i - 1],
                                     hp.dec_prenet[i - 1], nullptr, 0, ACT_RELU, w.pz[i] + (size_t)t * Pw, n * Pw);
      TRY(run_skinny(st, B, &j, 1));
    }
    const float* hAp = (t == 0) ? (att_init ? att_init : w.zero) : w.hA + (size_t)(t - 1) * As; const int ldhA = (t == 0) ? As : n * As;   // tacotron.py:183-197
    const size_t oa = (size_t)t * As;
    TRY(gru_cell_train(m, st, m->att_gru, B, w.pz[np - 1] + (size_t)t * Pz, n * Pz, hAp, ldhA, w.hA + oa, w.rhA + oa, w.uA + oa, w.xcA + oa,
                       w.rA + oa, w.cA + oa, n * As, nullptr));
    { AttnArgs a; memset(&a, 0, sizeof a);
      a.hq = w.hA + oa; a.ldhq = n * As; a.wq = AP(m, m->raw_wq); a.As = As; a.keys = w.keys; a.values = enc_out; a.v = AP(m, m->att_v); a.battn = AP(m, m->att_b);
      a.score_bias = AP(m, m->att_sb); a.align = w.alpha + (size_t)(t + 1) * T_in; a.align_prev = w.alpha + (size_t)t * T_in; a.ldalign = ldal;
      a.hist = align_hist; a.ctx = w.ctx + (size_t)t * Dc; a.ldctx = n * Dc;
      a.q_out = w.g_q + (size_t)t * A; a.ldq_out = n * A; a.e_out = w.g_e + (size_t)t * T_in; a.lde_out = n * T_in;
      a.T_in = T_in; a.A = A; a.D = D; a.type = hp.attention_type; a.step = t; a.n_steps = n;
      hipLaunchKernelGGL(k_attention, dim3(B), dim3(64 * ATT_NW), 0, st, a);
      HIPCHK(hipGetLastError()); }
    { SkJob j = sk_linear(m, m->concat_proj, w.hA + oa, n * As, As, w.ctx + (size_t)t * Dc, n * Dc, ACT_NONE, w.o[0] + (size_t)t * Hd, n * Hd);
      TRY(run_skinny(st, B, &j, 1)); }
    const size_t oh = (size_t)t * Hd;
    for (int i = 0; i < L; ++i) {
      const float* hp_ = (t == 0) ? ((dec_init && dec_init[i]) ? dec_init[i] : w.zero) : w.h[i] + (size_t)(t - 1) * Hd; const int ldhp = (t == 0) ? Hd : n * Hd;
      TRY(gru_cell_train(m, st, m->dec_gru[i], B, w.o[i] + oh, n * Hd, hp_, ldhp, w.h[i] + oh, w.rh[i] + oh, w.u[i] + oh, w.xc[i] + oh,
                         w.r[i] + oh, w.c[i] + oh, n * Hd, w.o[i + 1] + oh));
    }
    { SkJob j = sk_linear(m, m->frame_proj, w.o[L] + oh, n * Hd, Hd, nullptr, 0, ACT_NONE, mel + (size_t)t * rM, n * rM);
      j.o2 = reinterpret_cast<float*>(w.nz + (size_t)t * B);
      TRY(run_skinny(st, B, &j, 1)); }
  }
  return 0;
}

static SkJob sk_T(const taco_model* m, const SkW& wT, const float* dy, int lddy, float* out, int ldo) {
  return sk_linear(m, wT, dy, lddy, wT.K, nullptr, 0, ACT_NONE, out, ldo);
}
// one GRUCell backward: dout [B,H] (+carry) -> dx [B,I] (+dres), new carry; tape slices at step t.
// skip_a: the 'a' part (dht, dcp, dgp_u) was already produced by the k_gru_bwd_ca of the cell above.
// next: when given, the 'c' part also runs the 'a' part of the cell below (k_gru_bwd_ca).  relu_of: mask dx by relu_of > 0.
struct GruBwdNext { const float* carry; const float* u; const float* c; const float* hprev; float* dcp; float* dgp; int ldh; };
static int gru_cell_backward(const TrainCtx& x, const GruT& gt, int B, const float* dout, int lddo, float* carry, bool add_carry,
                             const float* u, const float* c, const float* r, const float* hprev, int ld, float* g_dcp, float* g_dgp,
                             const float* dres, int lddres, float* dx, int lddx, const DecTape& w, bool skip_a = false,
                             const GruBwdNext* next = nullptr, const float* relu_of = nullptr, int ldrelu = 0, int ldh = -1) {
  if (ldh < 0) ldh = ld;
  const taco_model* m = x.t->sm; hipStream_t st = x.st;
  const int H = gt.H, I = gt.I, W = I + H;
  if (!skip_a)
    hipLaunchKernelGGL(k_gru_bwd_a, EWGRID((size_t)B * H), 0, st, dout, lddo, add_carry ? carry : (const float*)nullptr, u, ld, c, ld, hprev, ldh,
                       w.dht, g_dcp, ld, g_dgp, 2 * ld, B, H);
  { SkJob j = sk_T(m, gt.cT, g_dcp, ld, w.tmp1, W); TRY(run_skinny(st, B, &j, 1)); }
  hipLaunchKernelGGL(k_gru_bwd_b, EWGRID((size_t)B * H), 0, st, w.tmp1, W, I, hprev, ldh, r, ld, u, ld, w.dht, g_dgp, 2 * ld, w.dhp, B, H);
  { SkJob j = sk_T(m, gt.gT, g_dgp, 2 * ld, w.tmp2, W); TRY(run_skinny(st, B, &j, 1)); }
  if (next)   // note: dht of the lower cell overwrites w.dht -- this cell's dht was consumed by its 'b' part above
    hipLaunchKernelGGL(k_gru_bwd_ca, EWGRID((size_t)B * W), 0, st, w.tmp1, w.tmp2, W, I, dres, lddres, w.dhp, dx, lddx, carry, B, H,
                       next->carry, next->u, ld, next->c, ld, next->hprev, next->ldh, w.dht, next->dcp, ld, next->dgp, 2 * ld);
  else
    hipLaunchKernelGGL(k_gru_bwd_c, EWGRID((size_t)B * W), 0, st, w.tmp1, w.tmp2, W, I, dres, lddres, w.dhp, dx, lddx, carry, B, H, relu_of, ldrelu);
  HIPCHK(hipGetLastError());
  return 0;
}
static int gru_weight_grads(const TrainCtx& x, const std::string& name, int I, int H, const float* xin, int ldx, const float* hseq,
                            const float* rh, const float* g_dgp, const float* g_dcp, int R, int n) {
  hipStream_t st = x.st;
  float* Gg = x.g(name + "/gates/kernel"); float* Gc = x.g(name + "/candidate/kernel");
  TRY(run_wgrad(st, xin, nullptr, ldx, g_dgp, 2 * H, Gg, 2 * H, R, 0, I, 2 * H));
  TRY(run_wgrad(st, hseq, nullptr, H, g_dgp, 2 * H, Gg + (size_t)I * 2 * H, 2 * H, R, n, H, 2 * H, 1, 1));   // previous state = one step earlier
  TRY(run_wgrad(st, xin, nullptr, ldx, g_dcp, H, Gc, H, R, 0, I, H));
  TRY(run_wgrad(st, rh, nullptr, H, g_dcp, H, Gc + (size_t)I * H, H, R, 0, H, H));
  TRY(run_colsum(st, g_dgp, 2 * H, nullptr, 0, nullptr, nullptr, x.g(name + "/gates/bias"), nullptr, R, 2 * H, 0));
  TRY(run_colsum(st, g_dcp, H, nullptr, 0, nullptr, nullptr, x.g(name + "/candidate/bias"), nullptr, R, H, 0));
  return 0;
}
static int decoder_backward(const TrainCtx& x, const float* enc_out, int B, int T_in, int n, const float* teach, const float* dmel,
                            float* denc, const DecTape& w, const float* att_init = nullptr, const float* const* dec_init = nullptr,
                            float* d_att_init = nullptr, float* const* d_dec_init = nullptr, float* dspk = nullptr) {
  const taco_model* m = x.t->sm; hipStream_t st = x.st;
  const TrainPacks& tp = x.t->tp;
  const taco_hparams& hp = m->hp;
  const int D = 2 * hp.enc_rnn_size, As = hp.attention_state_size, Hd = hp.dec_rnn_size, A = hp.attention_size;
  const int Mm = hp.num_mels, rM = Mm * hp.reduction_factor, L = hp.dec_layer_num, np = hp.dec_prenet_n, Pl = hp.dec_prenet[np - 1];
  const int ldal = (n + 1) * T_in, R = B * n;
  const int S = simple_S(m), Dc = D + S, Pz = Pl + S;
  if (S && !dspk) return fail(TACO_ERR_ARG, "speaker-embedding gradient buffer missing");
  { ZeroBatch zb(st);               // the loop's accumulators: one fill launch
    HIPCHK(zb.add(w.dkeys, (size_t)B * T_in * A * sizeof(float)));
    HIPCHK(zb.add(w.dvalues, (size_t)B * T_in * D * sizeof(float)));
    HIPCHK(zb.add(w.dv_acc, (size_t)B * A * sizeof(float)));
    HIPCHK(zb.add(w.dsb_acc, (size_t)B * sizeof(float)));
    HIPCHK(zb.add(w.dalpha, (size_t)B * T_in * sizeof(float)));
    HIPCHK(zb.add(w.dctx, (size_t)B * D * sizeof(float)));
    HIPCHK(zb.add(w.dhA, (size_t)B * As * sizeof(float)));
    for (int i = 0; i < L; ++i) HIPCHK(zb.add(w.dh[i], (size_t)B * Hd * sizeof(float)));
    HIPCHK(zb.run()); }
  const size_t attn_lds = (size_t)(2 * ((A + 3) & ~3) + ((D + 3) & ~3) + 5 * ((T_in + 3) & ~3) + ATB_NW * 256) * sizeof(float);
  if (attn_lds > 160 * 1024 || (As % 4) || (D % 4) || (A % 4)) return fail(TACO_ERR_UNSUPPORTED, "attention sizes not supported by the backward kernel");
  // the whole loop as ONE persistent launch (k_decoder_bwd_xcd, taco_decoder_bwd_xcd.h) when the forward left its tape in that
  // kernel's layout; the launch-per-stage loop below is the general path (other widths, 'simple', more than 64 rows)
  const bool persistent = x.t->bptt_persistent && w.tape256 && L == 2 && np == 2 && (size_t)DXT_N * w.tstride < (1u << 31) &&
                          dbx_usable(m, B, T_in);
  x.t->sm->last_bptt = persistent ? 1 : 0;
  if (persistent) {
    DbArgs a; memset(&a, 0, sizeof a);
    a.tape = w.tape256; a.tstride = w.tstride; a.tp_p2 = w.pz[1]; a.ld_p2 = Pz; a.tp_e = w.g_e; a.tp_alpha = w.alpha;
    a.keys = w.keys; a.values = enc_out;
    // d o2 of every step does not depend on the recurrence: ONE GEMM [B n, r M] x [r M, 256] ahead of the loop
    TRY(run_dgrad(m, st, tp.frame_d, dmel, rM, R, 0, w.g_do2, Hd));
    a.g_do2 = w.g_do2;
    a.h_att0 = att_init; a.h10 = dec_init ? dec_init[0] : nullptr; a.h20 = dec_init ? dec_init[1] : nullptr;
    a.g_dcp2 = w.g_dcp[1]; a.g_dgp2 = w.g_dgp[1]; a.g_dcp1 = w.g_dcp[0]; a.g_dgp1 = w.g_dgp[0]; a.g_do0 = w.g_do0;
    a.g_dcpA = w.g_dcpA; a.g_dgpA = w.g_dgpA; a.g_dz1 = w.g_dz[0]; a.g_dz2 = w.g_dz[1]; a.g_dq = w.g_dq; a.g_de = w.g_de; a.g_dctx = w.g_dctx;
    a.d_att_init = d_att_init; a.d_h10 = d_dec_init ? d_dec_init[0] : nullptr; a.d_h20 = d_dec_init ? d_dec_init[1] : nullptr;
    a.dsb_acc = w.dsb_acc;
    TRY(dbx_launch(m, st, a, B, T_in, n, w.xbuf, w.dxctl));
    if (S) {
      // 'simple': d speaker embedding = sum over steps of [d o0 . Wcc^T]_spk + [d c_pre(att) . Wc^T + d gates(att) . Wg^T]_spk.  Linear in the
      // pre-activation gradients: sum those over time first, then ONE transposed product each (the per-stage chain adds step by step).
      hipLaunchKernelGGL(k_time_sum, EWGRID((size_t)B * Hd), 0, st, (const float*)w.g_do0, w.dht, B, n, Hd);
      { SkJob j = sk_T(m, tp.concat_T, w.dht, Hd, w.dIn, As + Dc); TRY(run_skinny(st, B, &j, 1)); }
      hipLaunchKernelGGL(k_add2d, EWGRID((size_t)B * S), 0, st, dspk, S, w.dIn + As + D, As + Dc, B, S);
      hipLaunchKernelGGL(k_time_sum, EWGRID((size_t)B * As), 0, st, (const float*)w.g_dcpA, w.dht, B, n, As);
      { SkJob j = sk_T(m, tp.att.cT, w.dht, As, w.tmp1, Pz + As); TRY(run_skinny(st, B, &j, 1)); }
      hipLaunchKernelGGL(k_add2d, EWGRID((size_t)B * S), 0, st, dspk, S, w.tmp1 + Pl, Pz + As, B, S);
      hipLaunchKernelGGL(k_time_sum, EWGRID((size_t)B * 2 * As), 0, st, (const float*)w.g_dgpA, w.dpz, B, n, 2 * As);
      { SkJob j = sk_T(m, tp.att.gT, w.dpz, 2 * As, w.tmp2, Pz + As); TRY(run_skinny(st, B, &j, 1)); }
      hipLaunchKernelGGL(k_add2d, EWGRID((size_t)B * S), 0, st, dspk, S, w.tmp2 + Pl, Pz + As, B, S);
      HIPCHK(hipGetLastError());
    }
  }
  for (int t = n - 1; t >= 0 && !persistent; --t) {
    const size_t oh = (size_t)t * Hd, oa = (size_t)t * As;
    { SkJob j = sk_T(m, tp.frame_T, dmel + (size_t)t * rM, n * rM, w.do_[L], Hd); TRY(run_skinny(st, B, &j, 1)); }
    for (int i = L - 1; i >= 0; --i) {
      auto hp_of = [&](int l) -> const float* { return (t == 0) ? ((dec_init && dec_init[l]) ? dec_init[l] : nullptr) : w.h[l] + (size_t)(t - 1) * Hd; };
      const float* hprev = hp_of(i);
      const int ldh = (t == 0) ? Hd : n * Hd;
      float* dx = (i == 0) ? w.g_do0 + oh : w.do_[i]; const int lddx = (i == 0) ? n * Hd : Hd;
      GruBwdNext nx; const GruBwdNext* pnx = nullptr;
      if (i > 0) {   // the cell below consumes this dx at once: its 'a' part rides in this cell's 'c' launch
        nx.carry = w.dh[i - 1]; nx.u = w.u[i - 1] + oh; nx.c = w.c[i - 1] + oh; nx.hprev = hp_of(i - 1); nx.ldh = ldh;
        nx.dcp = w.g_dcp[i - 1] + oh; nx.dgp = w.g_dgp[i - 1] + 2 * oh; pnx = &nx;
      }
      TRY(gru_cell_backward(x, tp.dec[i], B, w.do_[i + 1], Hd, w.dh[i], true, w.u[i] + oh, w.c[i] + oh, w.r[i] + oh, hprev, n * Hd,
                            w.g_dcp[i] + oh, w.g_dgp[i] + 2 * oh, w.do_[i + 1], Hd, dx, lddx, w, i < L - 1, pnx, nullptr, 0, ldh));
    }
    // concat projection: [h_att | ctx] <- d o0; the attention backward adds the two halves to dhA / dctx itself
    { SkJob j = sk_T(m, tp.concat_T, w.g_do0 + oh, n * Hd, w.dIn, As + Dc); TRY(run_skinny(st, B, &j, 1)); }
    if (S) hipLaunchKernelGGL(k_add2d, EWGRID((size_t)B * S), 0, st, dspk, S, w.dIn + As + D, As + Dc, B, S);
    { AttnBArgs a; memset(&a, 0, sizeof a);
      a.q = w.g_q + (size_t)t * A; a.ldq = n * A; a.e = w.g_e + (size_t)t * T_in; a.lde = n * T_in; a.wqT = AP(m, tp.wqT);
      a.keys = w.keys; a.values = enc_out; a.v = AP(m, m->att_v); a.score_bias = AP(m, m->att_sb); a.battn = AP(m, m->att_b);
      a.alpha = w.alpha + (size_t)(t + 1) * T_in; a.alpha_prev = w.alpha + (size_t)t * T_in; a.ldal = ldal;
      a.dctx = w.dctx; a.lddctx = D; a.dctx_out = w.g_dctx + (size_t)t * D; a.lddco = n * D; a.dalpha = w.dalpha;
      a.de_out = w.g_de + (size_t)t * T_in; a.ldde = n * T_in; a.dsb_acc = w.dsb_acc;
      a.dq = w.g_dq + (size_t)t * A; a.lddq = n * A; a.dhq = w.dhA; a.lddhq = As; a.T_in = T_in; a.A = A; a.D = D; a.As = As; a.type = hp.attention_type;
      a.cat = w.dIn; a.ldcat = As + Dc;
      hipLaunchKernelGGL(k_attention_bwd, dim3(B), dim3(64 * ATB_NW), attn_lds, st, a);
      HIPCHK(hipGetLastError()); }
    { const float* hprev = (t == 0) ? att_init : w.hA + (size_t)(t - 1) * As;
      // dx of the attention GRU = gradient of the (ReLU) prenet output: masked here, written straight to the tape
      if (!S) {
        TRY(gru_cell_backward(x, tp.att, B, w.dhA, As, w.dhA, false, w.uA + oa, w.cA + oa, w.rA + oa, hprev, n * As, w.g_dcpA + oa,
                              w.g_dgpA + 2 * oa, nullptr, 0, w.g_dz[np - 1] + (size_t)t * Pl, n * Pl, w, false, nullptr,
                              w.pz[np - 1] + (size_t)t * Pl, n * Pl, (t == 0) ? As : n * As));
      } else {   // 'simple': the cell input is [prenet output | speaker embedding]: split its gradient
        TRY(gru_cell_backward(x, tp.att, B, w.dhA, As, w.dhA, false, w.uA + oa, w.cA + oa, w.rA + oa, hprev, n * As, w.g_dcpA + oa,
                              w.g_dgpA + 2 * oa, nullptr, 0, w.dpz, Pz, w, false, nullptr, nullptr, 0, (t == 0) ? As : n * As));
        hipLaunchKernelGGL(k_relu_bwd, EWGRID((size_t)B * Pl), 0, st, w.dpz, Pz, w.pz[np - 1] + (size_t)t * Pz, n * Pz, w.g_dz[np - 1] + (size_t)t * Pl, n * Pl, B, Pl);
        hipLaunchKernelGGL(k_add2d, EWGRID((size_t)B * S), 0, st, dspk, S, w.dpz + Pl, Pz, B, S);
      } }
    for (int i = np - 1; i >= 1; --i) {   // prenet layers np..2: d z_{i-1} = (d z_i . W_i^T) masked by the ReLU of layer i-1 (skinny epilogue)
      const int P = hp.dec_prenet[i], Pm = hp.dec_prenet[i - 1];
      SkJob j = sk_T(m, tp.decpre_T[i], w.g_dz[i] + (size_t)t * P, n * P, w.g_dz[i - 1] + (size_t)t * Pm, n * Pm);
      j.e0 = w.pz[i - 1] + (size_t)t * Pm; j.lde0 = n * Pm;
      TRY(run_skinny(st, B, &j, 1));
    }
    // layer 1: only the context columns of its input carry a gradient (the frame is the teacher's): d ctx(t-1) = d z_0 . W1[Mm:,:]^T
    { SkJob j = sk_T(m, tp.decpre0_ctxT, w.g_dz[0] + (size_t)t * hp.dec_prenet[0], n * hp.dec_prenet[0], w.dctx, D); TRY(run_skinny(st, B, &j, 1)); }
    HIPCHK(hipGetLastError());
  }
  // gradients of the initial states (deepvoice: they come from the speaker layers) = the carries left after step 0
  if (d_att_init && !persistent) hipLaunchKernelGGL(k_copy2d, EWGRID((size_t)B * As), 0, st, w.dhA, As, d_att_init, As, B, As);
  for (int i = 0; i < L && !persistent; ++i)
    if (d_dec_init && d_dec_init[i]) hipLaunchKernelGGL(k_copy2d, EWGRID((size_t)B * Hd), 0, st, w.dh[i], Hd, d_dec_init[i], Hd, B, Hd);
  HIPCHK(hipGetLastError());
  // first-step terms of the recurrent kernels' gates rows: the state before step 0 is the initial state, not a tape row
  if (att_init) TRY(run_wgrad(st, att_init, nullptr, As, w.g_dgpA, n * 2 * As, x.g("decoder/attention_gru/gates/kernel") + (size_t)Pz * 2 * As, 2 * As, B, 0, As, 2 * As));
  for (int i = 0; i < L; ++i)
    if (dec_init && dec_init[i])
      TRY(run_wgrad(st, dec_init[i], nullptr, Hd, w.g_dgp[i], n * 2 * Hd, x.g("decoder/gru_" + std::to_string(i + 1) + "/gates/kernel") + (size_t)Hd * 2 * Hd, 2 * Hd, B, 0, Hd, 2 * Hd));
  // ---- weight gradients, hoisted over all steps: rows (b, t) of the [B, n, .] tapes ----
  wg_begin(st);              // ~50 small products over final tapes: a few group launches (ended before d values is read below)
  struct WgEnd { ~WgEnd() { if (g_wgb.active) (void)wg_end(); } } wg_end_guard;
  TRY(run_wgrad(st, w.o[L], nullptr, Hd, dmel, rM, x.g("decoder/frame_projection/kernel"), rM, R, 0, Hd, rM));
  TRY(run_colsum(st, dmel, rM, nullptr, 0, nullptr, nullptr, x.g("decoder/frame_projection/bias"), nullptr, R, rM, 0));
  for (int i = 0; i < L; ++i)
    TRY(gru_weight_grads(x, "decoder/gru_" + std::to_string(i + 1), Hd, Hd, w.o[i], Hd, w.h[i], w.rh[i], w.g_dgp[i], w.g_dcp[i], R, n));
  { float* Gk = x.g("decoder/concat_projection/kernel");
    TRY(run_wgrad(st, w.hA, nullptr, As, w.g_do0, Hd, Gk, Hd, R, 0, As, Hd));
    TRY(run_wgrad(st, w.ctx, nullptr, Dc, w.g_do0, Hd, Gk + (size_t)As * Hd, Hd, R, 0, Dc, Hd));     // context (+ speaker) rows
    TRY(run_colsum(st, w.g_do0, Hd, nullptr, 0, nullptr, nullptr, x.g("decoder/concat_projection/bias"), nullptr, R, Hd, 0)); }
  TRY(run_wgrad(st, w.hA, nullptr, As, w.g_dq, A, x.g("attention/query_layer/kernel"), A, R, 0, As, A));
  TRY(gru_weight_grads(x, "decoder/attention_gru", Pz, As, w.pz[np - 1], Pz, w.hA, w.rhA, w.g_dgpA, w.g_dcpA, R, n));
  for (int i = np - 1; i >= 0; --i) {
    const int P = hp.dec_prenet[i];
    const std::string nm = "decoder/prenet/dense_" + std::to_string(i + 1);
    float* Gk = x.g(nm + "/kernel");
    if (i > 0) TRY(run_wgrad(st, w.pz[i - 1], nullptr, hp.dec_prenet[i - 1], w.g_dz[i], P, Gk, P, R, 0, hp.dec_prenet[i - 1], P));
    else {   // input = concat(previous teacher frame, previous context): both one step earlier, zero at t = 0
      TRY(run_wgrad(st, teach, nullptr, Mm, w.g_dz[0], P, Gk, P, R, n, Mm, P, 1, 1));
      TRY(run_wgrad(st, w.ctx, nullptr, Dc, w.g_dz[0], P, Gk + (size_t)Mm * P, P, R, n, D, P, 1, 1));
    }
    TRY(run_colsum(st, w.g_dz[i], P, nullptr, 0, nullptr, nullptr, x.g(nm + "/bias"), nullptr, R, P, 0));
  }
  { const bool vn = hp.attention_type == 1;   // bah_norm: the kernels see v_hat = g*v/|v|; its gradient is mapped back to v and g below
    float* dvdst = vn ? w.dv_acc : x.g("attention/attention_v");
    if (vn) HIPCHK(zero_async(w.dv_acc, (size_t)A * sizeof(float), st));
    AttnKArgs k; k.keys = w.keys; k.q = w.g_q; k.de = w.g_de; k.v = AP(m, m->att_v); k.battn = AP(m, m->att_b); k.dkeys = w.dkeys; k.dv = dvdst;
    k.T_in = T_in; k.A = A; k.n = n;
    const int nj = cdiv(T_in, ATK_J);
    k.part = (g_det.p && (size_t)B * nj * A <= g_det.cap) ? g_det.p : nullptr;
    hipLaunchKernelGGL(k_attention_keys_bwd, dim3(cdiv(A, 256), nj, B), dim3(256), 0, st, k);
    if (k.part) hipLaunchKernelGGL(k_rows_reduce, EWGRID(A), 0, st, (const float*)k.part, B * nj, A, dvdst);
    if (vn) {
      hipLaunchKernelGGL(k_vnorm_bwd, dim3(1), dim3(256), 0, st, x.p("attention/attention_v"), x.p("attention/attention_g"), w.dv_acc,
                         x.g("attention/attention_v"), x.g("attention/attention_g"), A);
      TRY(run_colsum(st, w.g_dq, A, nullptr, 0, nullptr, nullptr, x.g("attention/attention_b"), nullptr, R, A, 0));   // b enters like the query
    }
    HIPCHK(hipGetLastError()); }
  for (int b = 0; b < B; ++b)    // d values[b] = alpha[b]^T . dctx[b]  ([T_in x n] . [n x D])
    TRY(run_wgrad(st, w.alpha + ((size_t)b * (n + 1) + 1) * T_in, nullptr, T_in, w.g_dctx + (size_t)b * n * D, D,
                  w.dvalues + (size_t)b * T_in * D, D, n, 0, T_in, D));
  if (hp.attention_type == 2) { hipLaunchKernelGGL(k_sum_all, dim3(1), dim3(256), 0, st, w.dsb_acc, B, x.g("attention/attention_score_bias")); HIPCHK(hipGetLastError()); }
  TRY(run_wgrad(st, enc_out, nullptr, D, w.dkeys, A, x.g("attention/memory_layer/kernel"), A, B * T_in, 0, D, A));
  TRY(wg_end());             // d values (the per-row products above) is an operand of the data gradient below
  TRY(run_dgrad(m, st, tp.mem_d, w.dkeys, A, B * T_in, 0, denc, D, w.dvalues, D));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// whole step: forward (tape) + loss + backward
// ---------------------------------------------------------------------------------------------------------------
static int train_forward_backward(taco_train* t, hipStream_t st, float* P, float* G, const int* ids, const int* lengths, const int* speaker_id, const float* mel_tgt,
                                  const float* lin_tgt, const float* loss_coeff, int B, int T_in, int T_out, int prioritize_loss,
                                  int sample_rate, float* d_losses, float* mel_out, float* lin_out, float* align_out, void* ws, size_t ws_bytes,
                                  bool do_backward, bool feed_back, bool freeze_moving = false) {
  taco_model* m = t->sm;
  const taco_hparams& hp = m->hp;
  const int r = hp.reduction_factor, Mm = hp.num_mels, F = hp.num_freq;
  if (T_out % r) return fail(TACO_ERR_SHAPE, "T_out %d is not a multiple of the reduction factor %d", T_out, r);
  const int n = T_out / r;
  if (n > hp.max_iters) return fail(TACO_ERR_SHAPE, "T_out/r = %d exceeds max_iters %d", n, hp.max_iters);
  TRY(check_common(m, B, T_in));
  struct EngineGuard {  // the GEMM helpers see this trainer's engine switches for the duration of this step only
    int w, d, e, p;
    EngineGuard(const taco_train* t) : w(g_wgrad_bf3), d(g_dgrad_bf3), e(g_dgrad_exact), p(g_wgrad_planes) { g_wgrad_bf3 = t->wgrad_bf3; g_dgrad_bf3 = t->dgrad_bf3; g_dgrad_exact = t->dgrad_exact; g_wgrad_planes = t->wgrad_planes; }
    ~EngineGuard() { g_wgrad_bf3 = w; g_dgrad_bf3 = d; g_dgrad_exact = e; g_wgrad_planes = p; }
  } engine_guard(t);
  if ((m->bf3 || m->bf3x6 || g_dgrad_bf3) && !t->bf3_current) return fail(TACO_ERR_STATE, "taco_train_set_exact_gemm(0) needs a taco_train_refresh before the next step (the split-bf16 weight planes are stale)");
  Carver cv(ws, ws_bytes);
  TrainWs w; carve_train(cv, t, B, T_in, n, w);
  if (!cv.ok()) return fail(TACO_ERR_STATE, "workspace too small: need %zu bytes, have %zu", cv.off, ws_bytes);
  TrainCtx x{t, st, P, G, do_backward && !freeze_moving};
  struct DetGuard {     // the reduction helpers see the scratch for the duration of this step only
    DetGuard(float* p, size_t cap) { g_det.p = p; g_det.cap = cap; }
    ~DetGuard() { g_det.p = nullptr; g_det.cap = 0; }
  } det_guard(t->deterministic ? w.detscr : nullptr, t->deterministic ? DET_SCRATCH_FLOATS : 0);
  struct WpGuard {      // the plane scratch of the weight gradients, for the duration of this step only
    const taco_train* t;
    WpGuard(const taco_train* t_, uint4* p, size_t cap) : t(t_) { g_wps.p = p; g_wps.cap = cap; g_wp_count = 0; }
    ~WpGuard() { g_wps.p = nullptr; g_wps.cap = 0; t->planes_problems = g_wp_count; }
  } wp_guard(t, t->wgrad_planes ? w.wpscr : nullptr, t->wgrad_planes ? w.wps_uint4 : 0);
  const int Me = B * T_in, Mp = B * T_out;
  // ---- forward ----
  const float* cur = x.p("embedding"); int curd = hp.embedding_size;
  for (int i = 0; i < hp.enc_prenet_n; ++i) {
    GemmCall g; g.x = cur; g.ldx = curd; g.gather = (i == 0) ? ids : nullptr; g.M = Me; g.act = ACT_RELU; g.out = w.pre[i]; g.ldo = hp.enc_prenet[i];
    TRY(run_gemm(m, st, &m->enc_prenet[i], 1, false, g));
    cur = w.pre[i]; curd = hp.enc_prenet[i];
  }
  const bool dv = is_deepvoice(m), simple = is_simple(m);
  const int S = hp.speaker_embedding_size;
  if ((dv || simple) && !speaker_id) return fail(TACO_ERR_ARG, "speaker_id required for a multi-speaker model");
  if (simple) {      // the embedding row of every utterance (tacotron.py:47-50); concatenated in the decoder and the linear head
    hipLaunchKernelGGL(k_gather_rows, EWGRID((size_t)B * S), 0, st, AP(m, m->spk_emb), speaker_id, B, S, w.spk.emb);
    HIPCHK(hipGetLastError());
  }
  if (dv) TRY(spk_forward(m, st, speaker_id, B, w.spk));       // before_highway, encoder / attention / decoder initial states (tacotron.py:52-79)
  const int L = hp.dec_layer_num;
  const float* dec_init[4] = {nullptr, nullptr, nullptr, nullptr}; float* d_dec_init[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < L && dv; ++i) { dec_init[i] = w.spk.vec[3 + i]; d_dec_init[i] = w.dvec[3 + i]; }
  TRY(cbhg_forward_train(x, m->enc, t->tp.enc, "encoder_cbhg", cur, B, T_in, lengths, w.enc, dv ? w.spk.vec[0] : nullptr, dv ? w.spk.vec[1] : nullptr));
  const float* enc_out = w.enc.out;
  // teacher inputs: every r-th target frame (helpers.py:44): teach[b, t] = mel_targets[b, t*r + r-1]
  hipLaunchKernelGGL(k_copy2d, EWGRID((size_t)B * n * Mm), 0, st, mel_tgt + (size_t)(r - 1) * Mm, r * Mm, w.teach, Mm, B * n, Mm);
  HIPCHK(hipGetLastError());
  float* mel = mel_out ? mel_out : w.mel; float* lin = lin_out ? lin_out : w.linear;
  if (feed_back && do_backward) return fail(TACO_ERR_UNSUPPORTED, "rnn_decoder_test_mode is forward-only (the reference uses it for the test model's loss, train.py:158-166)");
  TRY(decoder_forward_train(x, enc_out, B, T_in, n, w.teach, mel, align_out, w.dec, feed_back, dv ? w.spk.vec[2] : nullptr, dv ? dec_init : nullptr,
                            simple ? w.spk.emb : nullptr));
  TRY(cbhg_forward_train(x, m->post, t->tp.post, "post_cbhg", mel, B, T_out, nullptr, w.post));
  { GemmCall g; g.x = w.post.out; g.ldx = 2 * hp.post_rnn_size; g.M = Mp; g.out = lin; g.ldo = F;
    if (simple) {    // linear(concat(tiled speaker_embed, post)) (tacotron.py:226-235): the speaker rows give one vector per utterance
      SkJob j = sk_linear(m, m->lin_spk, w.spk.emb, S, S, nullptr, 0, ACT_NONE, w.linrv, F);
      TRY(run_skinny(st, B, &j, 1));
      g.T = T_out; g.rowvec = w.linrv; g.ldrv = F;
    }
    TRY(run_gemm(m, st, &m->linear, 1, false, g)); }
  // ---- loss (tacotron.py:274-302) ----
  if (d_losses) TRY(taco_loss_f32((void*)st, mel, mel_tgt, lin, lin_tgt, loss_coeff, B, T_out, Mm, F, prioritize_loss, sample_rate, d_losses,
                                  w.losspart, (size_t)TR_MAXBLK * 8 * sizeof(double)));
  if (!do_backward) return 0;
  // ---- backward ----
  HIPCHK(zero_async(G, t->NP * sizeof(float), st));
  int c_lo = 0, c_hi = 0; float s_lin = 1.0f / ((float)Mp * F), s_band = 0.f;
  if (prioritize_loss) {
    c_hi = (int)(5000.0 / (sample_rate * 0.5) * F); c_lo = (int)(165.0 / (sample_rate * 0.5) * F);
    s_lin = 0.5f / ((float)Mp * F); s_band = 0.5f / ((float)Mp * (c_hi - c_lo));
  }
  hipLaunchKernelGGL(k_l1_grad, EWGRID((size_t)Mp * Mm), 0, st, mel, mel_tgt, loss_coeff, Mp, T_out, Mm, 1.0f / ((float)Mp * Mm), 0, 0, 0.f, w.dmel);
  hipLaunchKernelGGL(k_l1_grad, EWGRID((size_t)Mp * F), 0, st, lin, lin_tgt, loss_coeff, Mp, T_out, F, s_lin, c_lo, c_hi, s_band, w.dlin);
  HIPCHK(hipGetLastError());
  const int Hp2 = 2 * hp.post_rnn_size;
  TRY(run_wgrad(st, w.post.out, nullptr, Hp2, w.dlin, F, x.g("linear/kernel") + (simple ? (size_t)S * F : 0), F, Mp, 0, Hp2, F));
  if (simple) {      // speaker rows of the head: the embedding is constant over time, so they see the time-summed gradient
    HIPCHK(zero_async(w.dspk_emb, (size_t)B * S * sizeof(float), st));
    hipLaunchKernelGGL(k_time_sum, EWGRID((size_t)B * F), 0, st, w.dlin, w.dlin_sum, B, T_out, F);
    HIPCHK(hipGetLastError());
    TRY(run_wgrad(st, w.spk.emb, nullptr, S, w.dlin_sum, F, x.g("linear/kernel"), F, B, 0, S, F));
    SkJob j = sk_T(m, t->tp.lin_spk_T, w.dlin_sum, F, w.dspk_emb, S);
    TRY(run_skinny(st, B, &j, 1));
  }
  TRY(run_colsum(st, w.dlin, F, nullptr, 0, nullptr, nullptr, x.g("linear/bias"), nullptr, Mp, F, 0));
  float* dpost = w.dpost; float* dmel_post = w.dmel_post;
  TRY(run_dgrad(m, st, t->tp.lin_d, w.dlin, F, Mp, 0, dpost, Hp2));
  TRY(cbhg_backward(x, m->post, t->tp.post, "post_cbhg", mel, nullptr, B, T_out, nullptr, dpost, dmel_post, w.post));
  hipLaunchKernelGGL(k_add2d, EWGRID((size_t)Mp * Mm), 0, st, w.dmel, Mm, dmel_post, Mm, Mp, Mm);
  HIPCHK(hipGetLastError());
  TRY(decoder_backward(x, enc_out, B, T_in, n, w.teach, w.dmel, w.denc, w.dec, dv ? w.spk.vec[2] : nullptr, dv ? dec_init : nullptr,
                       dv ? w.dvec[2] : nullptr, dv ? d_dec_init : nullptr, simple ? w.dspk_emb : nullptr));
  if (simple) {
    run_embed_bwd(st, w.dspk_emb, speaker_id, x.g("speaker_embedding"), B, S, hp.num_speakers);
    HIPCHK(hipGetLastError());
  }
  float* dpre = w.dpre[hp.enc_prenet_n - 1];
  TRY(cbhg_backward(x, m->enc, t->tp.enc, "encoder_cbhg", cur, nullptr, B, T_in, lengths, w.denc, dpre, w.enc,
                    dv ? w.spk.vec[1] : nullptr, dv ? w.dvec[1] : nullptr, dv ? w.dvec[0] : nullptr, w.rowidx));
  if (dv) {   // speaker conditioning backward (tacotron.py:52-79)
    std::vector<std::string> names = {kSpkNames[0], kSpkNames[1], kSpkNames[2]};
    for (int i = 0; i < L; ++i) names.push_back("decoder_rnn_init_" + std::to_string(i + 1));
    const int dims[3] = {hp.enc_prenet[hp.enc_prenet_n - 1], hp.enc_rnn_size * 2, hp.attention_state_size};
    if (S == 1) {
      for (size_t i = 0; i < names.size(); ++i) {
        const int dd = i < 3 ? dims[i] : hp.dec_rnn_size;
        run_embed_bwd(st, w.dvec[i], speaker_id, x.g("spk/" + names[i] + "/table"), B, dd, hp.num_speakers);
      }
    } else {
      HIPCHK(zero_async(w.dspk_emb, (size_t)B * S * sizeof(float), st));
      for (size_t i = 0; i < names.size(); ++i) {
        const int dd = i < 3 ? dims[i] : hp.dec_rnn_size;
        hipLaunchKernelGGL(k_softsign_bwd, EWGRID((size_t)B * dd), 0, st, w.dvec[i], w.spk.vec[i], w.dzs, B * dd);
        TRY(run_wgrad(st, w.spk.emb, nullptr, S, w.dzs, dd, x.g("spk/" + names[i] + "/kernel"), dd, B, 0, S, dd));
        TRY(run_colsum(st, w.dzs, dd, nullptr, 0, nullptr, nullptr, x.g("spk/" + names[i] + "/bias"), nullptr, B, dd, 0));
        SkJob j = sk_T(m, t->tp.spk_T[i], w.dzs, dd, w.dec.tmp1, S);
        TRY(run_skinny(st, B, &j, 1));
        hipLaunchKernelGGL(k_add2d, EWGRID((size_t)B * S), 0, st, w.dspk_emb, S, w.dec.tmp1, S, B, S);
      }
      run_embed_bwd(st, w.dspk_emb, speaker_id, x.g("speaker_embedding"), B, S, hp.num_speakers);
    }
    HIPCHK(hipGetLastError());
  }
  for (int i = hp.enc_prenet_n - 1; i >= 0; --i) {
    const int N = hp.enc_prenet[i];
    const std::string nm = "prenet/dense_" + std::to_string(i + 1);
    hipLaunchKernelGGL(k_relu_bwd, EWGRID((size_t)Me * N), 0, st, dpre, N, w.pre[i], N, dpre, N, Me, N);
    HIPCHK(hipGetLastError());
    if (i > 0) TRY(run_wgrad(st, w.pre[i - 1], nullptr, hp.enc_prenet[i - 1], dpre, N, x.g(nm + "/kernel"), N, Me, 0, hp.enc_prenet[i - 1], N));
    else TRY(run_wgrad(st, x.p("embedding"), ids, hp.embedding_size, dpre, N, x.g(nm + "/kernel"), N, Me, 0, hp.embedding_size, N));
    TRY(run_colsum(st, dpre, N, nullptr, 0, nullptr, nullptr, x.g(nm + "/bias"), nullptr, Me, N, 0));
    float* dnext = (i > 0) ? w.dpre[i - 1] : w.demb;
    const int nd = (i > 0) ? hp.enc_prenet[i - 1] : hp.embedding_size;
    TRY(run_dgrad(m, st, t->tp.encpre_d[i], dpre, N, Me, 0, dnext, nd));
    dpre = dnext;
  }
  run_embed_bwd(st, dpre, ids, x.g("embedding"), (int)Me, hp.embedding_size, hp.num_symbols);
  HIPCHK(hipGetLastError());
  return 0;
}
