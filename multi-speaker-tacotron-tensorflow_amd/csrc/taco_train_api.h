// taco_train_api.h -- C ABI of the training path (declared in include/taco_abi.h); included inside extern "C".

int taco_train_create(const taco_hparams* hp, int device, taco_train** out) {
  if (!hp || !out) return fail(TACO_ERR_ARG, "null argument");
  taco_train* t = new taco_train();
  int rc = taco_model_create(hp, device, &t->sm);
  if (rc != 0) { delete t; return rc; }
  taco_model* sm = t->sm;
  size_t off = 0;
  for (auto& s : sm->spec) {          // flat layout = spec order, TF tensor layout, no padding
    size_t cnt = 1;
    for (int64_t d : s.second) cnt *= (size_t)d;
    t->poff[s.first] = off;
    HostTensor ht; ht.shape = s.second; ht.data.resize(cnt); ht.set = true;
    for (size_t k = 0; k < cnt; ++k) ht.data[k] = (float)(off + k + 1);     // index-valued "weights": the packs become index maps
    sm->raw[s.first] = ht;
    off += cnt;
  }
  t->NP = off;
  if (off >= (1u << 24)) { taco_model_destroy(sm); delete t; return fail(TACO_ERR_UNSUPPORTED, "more than 2^24 parameters"); }
  sm->tp = &t->tp;
  sm->bf3 = 1;                        // finalize builds the split-bf16 packs (as index lists) next to the fp32 ones ...
  rc = taco_model_finalize(sm);
  if (rc != 0) { taco_model_destroy(sm); delete t; return rc; }
  sm->bf3 = 0;                        // ... and the step starts on the exact-fp32 MFMA (k_gemm): taco_train_set_exact_gemm(t, 0) switches
  if (!sm->bf3_idx.empty()) {
    if (hipMalloc((void**)&t->d_bf3_idx, sm->bf3_idx.size() * sizeof(unsigned)) != hipSuccess ||
        hipMalloc((void**)&t->d_bf3_segs, sm->bf3_segs.size() * sizeof(Bf3Seg)) != hipSuccess ||
        hipMemcpy(t->d_bf3_idx, sm->bf3_idx.data(), sm->bf3_idx.size() * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t->d_bf3_segs, sm->bf3_segs.data(), sm->bf3_segs.size() * sizeof(Bf3Seg), hipMemcpyHostToDevice) != hipSuccess) {
      taco_model_destroy(sm); delete t; return fail(TACO_ERR_HIP, "split-bf16 index list allocation failed");
    }
    t->n_bf3 = sm->bf3_idx.size(); t->n_bf3_segs = (int)sm->bf3_segs.size();
    std::vector<unsigned>().swap(sm->bf3_idx);          // the host copy (tens of MB) is not needed again
  }
  t->arena_n = sm->arena_n;
  if (hipMalloc((void**)&t->d_map, t->arena_n * sizeof(float)) != hipSuccess ||
      hipMemcpy(t->d_map, sm->darena, t->arena_n * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) {
    taco_model_destroy(sm); delete t; return fail(TACO_ERR_HIP, "index map allocation failed");
  }
  if (sm->dx_fold_n && hipMalloc((void**)&t->d_fold, sm->dx_fold_n * sizeof(float)) != hipSuccess) {
    taco_model_destroy(sm); delete t; return fail(TACO_ERR_HIP, "fold buffer allocation failed");
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_rows_bwd<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_rows_bwd<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  *out = t;
  return 0;
}

void taco_train_destroy(taco_train* t) {
  if (!t) return;
  if (t->d_map) (void)hipFree(t->d_map);
  if (t->d_fold) (void)hipFree(t->d_fold);
  if (t->d_bf3_idx) (void)hipFree(t->d_bf3_idx);
  if (t->d_bf3_segs) (void)hipFree(t->d_bf3_segs);
  if (t->sm) taco_model_destroy(t->sm);
  delete t;
}

taco_model* taco_train_model(taco_train* t) { return t ? t->sm : nullptr; }

int taco_train_set_sync_bn(taco_train* t, taco_sync_sum_fn fn, void* user, int world_size) {
  if (!t) return fail(TACO_ERR_ARG, "null argument");
  if (fn && world_size < 1) return fail(TACO_ERR_ARG, "world_size %d", world_size);
  t->sync_fn = fn; t->sync_user = user; t->sync_world = fn ? world_size : 1;
  return 0;
}
int taco_train_set_exact_wgrad(taco_train* t, int on) {
  if (!t) return fail(TACO_ERR_ARG, "null argument");
  g_wgrad_bf3 = on ? 0 : 1;           // process-wide switch (an A/B and test hook, not a per-trainer setting)
  return 0;
}
int taco_train_set_exact_gemm(taco_train* t, int on) {
  if (!t) return fail(TACO_ERR_ARG, "null argument");
  t->sm->bf3 = on == 1 ? 0 : 1;
  g_dgrad_exact = on == 2 ? 1 : 0;
  return 0;
}
int taco_train_set_bptt_engine(taco_train* t, int persistent) {
  if (!t) return fail(TACO_ERR_ARG, "null argument");
  t->bptt_persistent = persistent ? 1 : 0;
  return 0;
}
int taco_train_set_deterministic(taco_train* t, int on) {
  if (!t) return fail(TACO_ERR_ARG, "null argument");
  t->deterministic = on ? 1 : 0;      // changes taco_train_workspace_bytes
  return 0;
}
// Test hook: the post-net BiGRU scan alone, forward with the gate tape and backward, on caller data -- ragged lengths and initial
// states included, which the post-net itself never has.  persistent = 1: k_bigru_duo<RG, true> + k_bigru_duo_bwd (needs H = 256 on a
// whole MI355X); 0: k_bigru_res<..., true> + k_bigru_rows_bwd.  All buffers device, caller-owned:
//   xproj [B*T, 6H] (the hoisted input projection, backward direction time-reversed per row), lengths [B] or null, h0 [B, 2H] or null,
//   dout [B*T, 2H]  ->  out [B*T, 2H], gsave [B*T, 6H], dg [B*T, 6H], rh [B*T, 2H], dh0 [B, 2H] (nullable); scratch: >= 1 MB.
int taco_train_debug_bigru(taco_train* t, void* hip_stream, const float* xproj, const int32_t* lengths, const float* h0, const float* dout,
                           int B, int T, int persistent, float* out, float* gsave, float* dg, float* rh, float* dh0, void* scratch, size_t scratch_bytes) {
  if (!t || !xproj || !dout || !out || !gsave || !dg || !rh || !scratch) return fail(TACO_ERR_ARG, "null argument");
  const taco_model* m = t->sm; const Cbhg& c = m->post; const int H = c.rnn;
  hipStream_t st = (hipStream_t)hip_stream;
  HIPCHK(hipSetDevice(m->device));
  const size_t need = std::max(gd_xbuf_granules(8), gb_xbuf_granules(8)) * sizeof(unsigned long long) + 512;
  if (scratch_bytes < need) return fail(TACO_ERR_STATE, "scratch too small: need %zu bytes", need);
  unsigned long long* gxbuf = (unsigned long long*)scratch; unsigned* gxctl = (unsigned*)((char*)scratch + need - 256);
  const size_t M = (size_t)B * T;
  HIPCHK(zero_async(gsave, M * 6 * H * sizeof(float), st));
  HIPCHK(zero_async(dg, M * 6 * H * sizeof(float), st));
  HIPCHK(zero_async(rh, M * 2 * H * sizeof(float), st));
  if (persistent) {
    if (!duo_usable(m, c, B, T) || !c.gb_pack) return fail(TACO_ERR_UNSUPPORTED, "the whole-chip scans need H = 256, at most 64 rows and an unpartitioned MI355X");
    TRY(duo_launch(m, st, c, B, T, xproj, lengths, h0, out, gsave, gxbuf, gxctl));
    GbArgs a; memset(&a, 0, sizeof a);
    a.wpack = AP(m, c.gb_pack); a.dout = dout; a.out = out; a.gsave = gsave; a.h0 = h0; a.lengths = lengths; a.dg = dg; a.rh = rh; a.dh0 = dh0;
    a.xbuf = gxbuf; a.ctl = gxctl; a.err = m->d_err; a.B = B; a.T = T; a.force_wt = m->dx_mode == 2 ? 1 : 0;
    int RG = 1;
    while (RG * DX_NGROUP < B) RG *= 2;
    HIPCHK(zero_async(gxbuf, need - 256 + 256, st));
    const size_t lds = std::max(gb_lds_floats(RG) * sizeof(float), (size_t)96 * 1024);
    const dim3 grid(DX_NGROUP * GD_MEMBERS), blk(512);
    switch (RG) {
      case 1: hipLaunchKernelGGL((k_bigru_duo_bwd<1>), grid, blk, lds, st, a); break;
      case 2: hipLaunchKernelGGL((k_bigru_duo_bwd<2>), grid, blk, lds, st, a); break;
      case 4: hipLaunchKernelGGL((k_bigru_duo_bwd<4>), grid, blk, lds, st, a); break;
      default: hipLaunchKernelGGL((k_bigru_duo_bwd<8>), grid, blk, lds, st, a); break;
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (H != 256) return fail(TACO_ERR_UNSUPPORTED, "post-net rnn size %d", H);
  { BigruSArgs a; memset(&a, 0, sizeof a);
    a.xproj = xproj; a.g2_0 = (const float2*)AP(m, c.res_g2[0]); a.g2_1 = (const float2*)AP(m, c.res_g2[1]);
    a.c1_0 = AP(m, c.raw_ch[0]); a.c1_1 = AP(m, c.raw_ch[1]); a.lengths = lengths; a.out = out; a.gsave = gsave; a.B = B; a.T = T; a.h0 = h0;
    hipLaunchKernelGGL((k_bigru_res<256, 64, 24, 1, true>), dim3(2 * B), dim3(512), bigru_res_lds(256, 24, 1), st, a); }
  { const CbhgT& ct = t->tp.post;
    const int R = (B >= 2) ? 2 : 1;
    const size_t lds = ((size_t)4 * R * H + (size_t)RP_NT * R * 4 + 64) * sizeof(float);
    BigruBArgs a; memset(&a, 0, sizeof a);
    a.dout = dout; a.out = out; a.gsave = gsave; a.wgT0 = AP(m, ct.ghT[0]); a.wgT1 = AP(m, ct.ghT[1]); a.wcT0 = AP(m, ct.chT[0]); a.wcT1 = AP(m, ct.chT[1]);
    a.lengths = lengths; a.dg = dg; a.rh = rh; a.B = B; a.T = T; a.H = H; a.h0 = h0; a.dh0 = dh0;
    if (R == 2) hipLaunchKernelGGL(k_bigru_rows_bwd<2>, dim3(2 * cdiv(B, R)), dim3(RP_NT), lds, st, a);
    else hipLaunchKernelGGL(k_bigru_rows_bwd<1>, dim3(2 * cdiv(B, R)), dim3(RP_NT), lds, st, a); }
  HIPCHK(hipGetLastError());
  return 0;
}
size_t taco_train_num_params(const taco_train* t) { return t ? t->NP : 0; }

int taco_train_param_offset(const taco_train* t, const char* name, size_t* offset) {
  if (!t || !name || !offset) return fail(TACO_ERR_ARG, "null argument");
  auto it = t->poff.find(name);
  if (it == t->poff.end()) return fail(TACO_ERR_ARG, "unknown parameter '%s'", name);
  *offset = it->second;
  return 0;
}

int taco_train_refresh(taco_train* t, void* hip_stream, const float* d_params) {
  if (!t || !d_params) return fail(TACO_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->sm->device));
  const taco_model* sm = t->sm;
  if (sm->dx_fold_n) {   // the persistent decoder's one computed operand: concat projection folded into GRU 1, from the current parameters
    const int H = sm->hp.dec_rnn_size, Z = sm->hp.attention_state_size + 2 * sm->hp.enc_rnn_size + simple_S(sm);
    hipLaunchKernelGGL(k_dx_fold, dim3(Z + 1), dim3(256), 0, (hipStream_t)hip_stream, d_params + t->poff.at("decoder/concat_projection/kernel"),
                       d_params + t->poff.at("decoder/concat_projection/bias"), d_params + t->poff.at("decoder/gru_1/gates/kernel"),
                       d_params + t->poff.at("decoder/gru_1/gates/bias"), d_params + t->poff.at("decoder/gru_1/candidate/kernel"), t->d_fold, Z, H);
  }
  hipLaunchKernelGGL(k_pack_gather, dim3(2048), dim3(256), 0, (hipStream_t)hip_stream, t->d_map, d_params, t->sm->darena, t->arena_n, (unsigned)t->NP,
                     (const float*)t->d_fold, (unsigned)sm->dx_fold_n);
  if (t->d_bf3_idx)                     // the same weights as split-bf16 planes (the map holds zeros there: this runs after the fp32 gather)
    hipLaunchKernelGGL(k_bf3_gather, dim3(2048), dim3(256), 0, (hipStream_t)hip_stream, (const unsigned*)t->d_bf3_idx, (const Bf3Seg*)t->d_bf3_segs,
                       t->n_bf3_segs, d_params, t->sm->darena, t->n_bf3, (unsigned)t->NP);
  if (t->sm->hp.attention_type == 1)    // bah_norm: the pack holds v_hat = g * v / |v| (computed, not copied)
    hipLaunchKernelGGL(k_vnorm_fold, dim3(1), dim3(256), 0, (hipStream_t)hip_stream, d_params + t->poff.at("attention/attention_v"),
                       d_params + t->poff.at("attention/attention_g"), t->sm->darena + (t->sm->att_v - 1), t->sm->hp.attention_size);
  HIPCHK(hipGetLastError());
  return 0;
}

size_t taco_train_workspace_bytes(const taco_train* t, int B, int T_in, int T_out) {
  if (!t || B <= 0 || T_in <= 0 || T_out <= 0) return 0;
  const int r = t->sm->hp.reduction_factor;
  Carver cv(nullptr, 0);
  TrainWs w; carve_train(cv, t, B, T_in, (T_out + r - 1) / r, w);
  return cv.off;
}

int taco_train_forward_backward(taco_train* t, void* hip_stream, float* d_params, float* d_grads, const int32_t* d_inputs,
                                const int32_t* d_input_lengths, const int32_t* d_speaker_id, const float* d_mel_targets, const float* d_linear_targets,
                                const float* d_loss_coeff, int B, int T_in, int T_out, int prioritize_loss, int sample_rate, float* d_losses,
                                float* d_mel_out, float* d_linear_out, float* d_alignments, int rnn_decoder_test_mode, void* d_workspace,
                                size_t workspace_bytes) {
  if (!t || !d_params || !d_inputs || !d_input_lengths || !d_mel_targets || !d_linear_targets || !d_workspace)
    return fail(TACO_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(t->sm->device));
  return train_forward_backward(t, (hipStream_t)hip_stream, d_params, d_grads, d_inputs, d_input_lengths, d_speaker_id, d_mel_targets, d_linear_targets,
                                d_loss_coeff, B, T_in, T_out, prioritize_loss, sample_rate, d_losses, d_mel_out, d_linear_out, d_alignments,
                                d_workspace, workspace_bytes, d_grads != nullptr, (rnn_decoder_test_mode & 1) != 0, (rnn_decoder_test_mode & 2) != 0);
}
