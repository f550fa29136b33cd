// taco_lib.hip -- host side of libtaco_hip.so: weight pack, stage orchestration, hipGraph plans and
// the C ABI of include/taco_abi.h.  Replaces the TF1 graph built by Tacotron.initialize()
// (models/tacotron.py:21-271 of the reference) and its sess.run execution (synthesizer.py:166-167).
// No CPU compute path exists here: every stage is a gfx950 kernel from taco_kernels.h.
#include "taco_kernels.h"
#include "taco_decoder_xcd.h"
#include "taco_bigru_xcd.h"
#include "taco_chain.h"
#include "taco_front.h"
#include "taco_head.h"
#include "taco_train_kernels.h"
#include "taco_backward_kernels.h"
#include "taco_wgrad_planes.h"
#include "taco_decoder_bwd_xcd.h"
#include "../../include/taco_abi.h"
#include "../../include/taco_debug.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <map>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
// ------------------------------------------------------------------------------------------------
// one whole-chip kernel at a time per device and process
// ------------------------------------------------------------------------------------------------
// The persistent kernels (k_decoder_xcd, k_bigru_oct / duo / xcd and their backward twins) need all 256 workgroups resident at once.  Two of them
// dispatched to one device together -- two models or plans of ONE process on different streams or threads -- can each end up waiting for compute
// units the other one holds until their bounded spins expire.  A ChipTurn in the scope of every such launch (and of every replay of a plan that
// contains one) puts them in a total order: under a per-device mutex, a launch on ANOTHER stream than the previous whole-chip launch first records
// an event on that previous stream (behind everything enqueued there so far) and makes its own stream wait for it.  Launches that follow each other
// on one stream -- the serving loop, bench.py -- cost a mutex and a pointer compare, no event (an event record behind every replay was measured:
// +0.01 ms per C2 forward).  Everything else of a forward still overlaps across streams.  Inside a stream capture nothing is waited for or
// recorded; the capture is only marked (g_captured_whole_chip), and taco_plan_launch takes the turn for the whole graph.  Other PROCESSES on the
// device are outside its reach: one process per GPU stays the deployment rule.
#include <mutex>
struct ChipTurnState { std::mutex mu; hipEvent_t ev = nullptr; hipStream_t last = nullptr; bool armed = false; };
static ChipTurnState g_turn[32];
static int g_chip_turns = 1;                                   // taco_debug_set_chip_turns: 0 switches the ordering off (A/B)
static thread_local bool g_captured_whole_chip = false;        // a whole-chip launch was captured on this thread since the flag was last cleared
struct ChipTurn {
  ChipTurnState* t = nullptr; hipStream_t st;
  ChipTurn(int device, hipStream_t stream) : st(stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { g_captured_whole_chip = true; return; }
    (void)hipGetLastError();
    if (!g_chip_turns || device < 0 || device >= 32) return;
    t = &g_turn[device];
    t->mu.lock();
    if (t->armed && t->last != st) {
      bool ordered = false;
      if (t->ev || hipEventCreateWithFlags(&t->ev, hipEventDisableTiming) == hipSuccess)
        ordered = hipEventRecord(t->ev, t->last) == hipSuccess && hipStreamWaitEvent(st, t->ev, 0) == hipSuccess;
      if (!ordered) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }      // the previous stream is gone (destroyed by its owner): its work may still run
    }
  }
  ~ChipTurn() {
    if (!t) return;
    t->armed = true; t->last = st;
    t->mu.unlock();
  }
  ChipTurn(const ChipTurn&) = delete; ChipTurn& operator=(const ChipTurn&) = delete;
};

// Stream-ordered zero fill as a kernel launch.  hipMemsetAsync nodes of a captured graph were seen to fill with a stale 16-byte
// pattern after another plan of the process had been destroyed (round 2: the census words and the exchange tags of a decoder plan,
// the BatchNorm sums of a captured train step); a kernel node has no such state.  p 4-byte aligned, bytes a multiple of 4.
__global__ __launch_bounds__(256) void k_zero_fill(uint32_t* __restrict__ p, size_t n16, size_t nw) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint4* q = reinterpret_cast<uint4*>(p);
  for (size_t i = i0; i < n16; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = n16 * 4 + i0; i < nw; i += stride) p[i] = 0u;
}
// several regions in ONE launch (a forward clears the decoder's exchange buffers, the stop flags and the post-net scan's exchange
// buffers up front: three fill launches and their graph edges less); every region 4-byte aligned and sized
struct ZeroRegions { uint32_t* p[4]; unsigned long long nw[4]; };
__global__ void k_zero_fill_multi(const ZeroRegions z) {
  uint32_t* p = z.p[blockIdx.y];
  const size_t nw = z.nw[blockIdx.y];
  const size_t n16 = (reinterpret_cast<uintptr_t>(p) & 15) ? 0 : nw / 4;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint4* q = reinterpret_cast<uint4*>(p);
  for (size_t i = i0; i < n16; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = n16 * 4 + i0; i < nw; i += stride) p[i] = 0u;
}
// up to ZM_MAX regions in one launch (the backward pass clears a dozen small accumulators in a row)
#define ZM_MAX 12
struct ZeroMany { uint32_t* p[ZM_MAX]; unsigned long long nw[ZM_MAX]; };
__global__ void k_zero_fill_many(const ZeroMany z) {
  uint32_t* p = z.p[blockIdx.y];
  const size_t nw = z.nw[blockIdx.y];
  const size_t n16 = (reinterpret_cast<uintptr_t>(p) & 15) ? 0 : nw / 4;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint4* q = reinterpret_cast<uint4*>(p);
  for (size_t i = i0; i < n16; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = n16 * 4 + i0; i < nw; i += stride) p[i] = 0u;
}
// Regions a caller up the stack has already cleared on this stream (forward_enqueue: ONE fill launch for every word the persistent kernels of a
// forward poll).  A stage clears its polled words through clear_polled(): skipped only when exactly that region -- same start, at least as many
// bytes -- is registered, so a stage whose region changes (or a new stage) clears itself instead of trusting a list kept elsewhere.
struct ClearedSet { const void* p[8]; size_t n[8]; int cnt; };
static thread_local ClearedSet g_cleared = {};
static void register_cleared(const ZeroRegions& z) {      // called by whoever has JUST enqueued the launch that clears z's regions
  for (int i = 0; i < 4; ++i)
    if (z.p[i] && g_cleared.cnt < 8) { g_cleared.p[g_cleared.cnt] = z.p[i]; g_cleared.n[g_cleared.cnt++] = z.nw[i] * 4; }
}
static hipError_t zero_async(void* p, size_t bytes, hipStream_t st);
static hipError_t clear_polled(void* p, size_t bytes, hipStream_t st) {
  for (int i = 0; i < g_cleared.cnt; ++i)
    if (g_cleared.p[i] == p && g_cleared.n[i] >= bytes) return hipSuccess;
  return zero_async(p, bytes, st);
}
static hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
  if (!bytes) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(p) & 3) || (bytes & 3)) return hipMemsetAsync(p, 0, bytes, st);
  const size_t nw = bytes / 4;
  const size_t n16 = (reinterpret_cast<uintptr_t>(p) & 15) ? 0 : nw / 4;
  const size_t items = n16 ? n16 + (nw - n16 * 4) : nw;
  const unsigned blocks = (unsigned)std::min<size_t>((items + 255) / 256, 2048);
  hipLaunchKernelGGL(k_zero_fill, dim3(blocks), dim3(256), 0, st, static_cast<uint32_t*>(p), n16, nw);
  return hipGetLastError();
}

// collects zero fills that follow one another on a stream and issues them as one launch (run() -- also from the destructor's caller)
struct ZeroBatch {
  ZeroMany z; int n = 0; size_t big = 0; hipStream_t st;
  explicit ZeroBatch(hipStream_t s) : st(s) { memset(&z, 0, sizeof z); }
  hipError_t add(void* p, size_t bytes) {
    if (!bytes) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(p) & 3) || (bytes & 3)) return zero_async(p, bytes, st);
    if (n == ZM_MAX) { hipError_t e = run(); if (e != hipSuccess) return e; }
    z.p[n] = static_cast<uint32_t*>(p); z.nw[n] = bytes / 4; big = std::max(big, bytes / 16); ++n;
    return hipSuccess;
  }
  hipError_t run() {
    if (!n) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((big + 255) / 256, 1), 1024);
    hipLaunchKernelGGL(k_zero_fill_many, dim3(blocks, n), dim3(256), 0, st, z);
    n = 0; big = 0; memset(&z, 0, sizeof z);
    return hipGetLastError();
  }
};

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(TACO_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define TRY(expr)            \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int rup(int a, int b) { return cdiv(a, b) * b; }
static inline size_t rup_sz(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool set = false;
};

struct ConvL {  // a k_gemm layer (conv1d+BN, dense, highway, hoisted GRU input projection)
  int kw = 1, cin = 0, cin_pad = 0, N = 0;
  size_t wp = 0, wp2 = 0, bias = 0, bias2 = 0, bns = 0, bnb = 0;  // arena offsets (+1; 0 = absent)
  size_t bh = 0, bl = 0, bh2 = 0, bl2 = 0;                          // split-bf16 packs (k_gemm_bf3), 0 = not built
  size_t bl3 = 0, bl3_2 = 0;                                       // third planes (training shadow model: the six-product instantiation)
  bool x6 = false;                                                 // inference: run this layer on the six-product (fp32-grade) instantiation (its third plane is built)
  int K16 = 0, cin_pad16 = 0;
  size_t wtail = 0; int ntail = 0;                                 // the last N % 32 (<= 4) columns as fp32 rows [ntail][cin]: the vector-ALU tail of k_head_sweep (taco_head.h)
  int var_index = -1;                                              // index into the device GemmVar array
};
struct SkW {  // a k_skinny weight
  int K = 0, Kq = 0, N = 0;
  size_t wp = 0, bias = 0;
};
struct GruDec { SkW gx, ch; int I = 0, H = 0; };   // gx: [I+H, 3H] = [gates (r|u) | x-rows of candidate]; ch: h-rows of candidate
struct Cbhg {
  int in_dim = 0, K = 0, C = 0, maxpool = 1, depth = 0, rnn = 0, pw = 3;
  int nproj = 0, proj_dim[4] = {0, 0, 0, 0};   // known from hparams alone (workspace sizing before finalize)
  std::vector<ConvL> bank;  // ordered widest first (longest workgroups are dispatched first)
  int bank_var0 = -1;
  std::vector<ConvL> proj;
  bool has_dense = false;
  ConvL dense;
  std::vector<ConvL> hw;
  ConvL xproj;              // N = 2 directions x (2H gates | H candidate), biases folded in
  SkW gh[2], ch[2];
  size_t raw_gh[2] = {0, 0}, raw_ch[2] = {0, 0};   // h-rows of the GRU kernels in TF layout (row-parallel scan)
  size_t res_g2[2] = {0, 0};                       // h-rows of gates/kernel as (r, u) pairs per unit: [H][H][2] (k_bigru_res)
  size_t gd_pack = 0, gb_pack = 0;                     // k_bigru_duo / k_bigru_duo_bwd: [2 dirs][32 members][12][512]
  size_t gob_pack = 0;                                 // k_bigru_oct_bwd (training): [2 dirs][8 members][48][512]
  size_t go_pack[3] = {0, 0, 0};                       // k_bigru_oct<UPW>, UPW = 1, 2, 4: [2 dirs][32 / UPW members][12 UPW][512]
  size_t gx_pack[2] = {0, 0}, gx_pack4[2] = {0, 0};   // per-thread weight packs of k_bigru_xcd (8-wave and 4-wave workgroups), H = 256 only
  // fused front (taco_front.h): per bank width (same order as `bank`) the produce pack [k32 step][16-channel tile][lane][8] (hi, lo)
  // and its step count; front_kind 0 = not built, 1 = <TN 2, XS 80, CINP 80, KWMAX 8> (post-net), 2 = <TN 1, XS 144, CINP 128, KWMAX 16> (encoder)
  std::vector<size_t> fr_wh, fr_wl; std::vector<int> fr_ns; int front_kind = 0;
  size_t res_g2p[2] = {0, 0}, res_c1p[2] = {0, 0}; // the same and the candidate h-rows with the columns in k_bigru_resw's thread order
                                                   // (column jb*4 + u = unit jb + 64*u): a thread's four units are 32 / 16 contiguous bytes
};

struct taco_model {
  taco_hparams hp;
  int device = 0;
  bool finalized = false;
  std::vector<std::pair<std::string, std::vector<int64_t>>> spec;  // required tensors
  std::map<std::string, HostTensor> raw;
  // packed
  std::vector<float> harena;
  float* darena = nullptr;
  size_t arena_n = 0;
  std::vector<GemmVar> hvars;
  std::map<std::string, ConvL> convs;
  std::map<std::string, SkW> skinny;
  std::map<std::string, GruDec> grus;
  std::vector<ConvL> enc_prenet;
  Cbhg enc, post;
  ConvL memory_layer, linear;
  std::vector<SkW> dec_prenet;
  GruDec att_gru;
  std::vector<GruDec> dec_gru;
  SkW query, concat_proj, frame_proj, lin_spk;   // lin_spk: speaker rows of the linear head ('simple')
  SkW prenet1_next;   // decoder prenet layer 1 of step t+1 as a function of step t's [GRU-stack output | context] (frame projection folded in)
  int fuse_prenet1 = 1;
  GruDec gru1_fold;   // decoder GRU 1 with the concat projection folded into its x rows: input [h_att | ctx (| spk)], extra columns = o0
  int fuse_concat = 0;
  int att_split = -1;   // -1: heuristic; 0/1: one workgroup per row; >1: k_att_scores + k_att_context with that many slices per row
  size_t att_v = 0, att_b = 0, att_sb = 0, emb = 0, spk_emb = 0, raw_wq = 0;
  std::vector<SkW> spk_dense;      // deepvoice: before_highway, enc_init, att_init, dec_init_i
  std::vector<size_t> spk_table;   // speaker_embedding_size == 1 variant
  int force_cfg = -1;
  unsigned* d_err = nullptr;   // set by a persistent kernel whose bounded spin expired
  int persist = 1;             // use the persistent BiGRU kernel when it fits
  int ff_rot = 1;              // k_pointwise_chain: workgroups of an XCD start their K loops at different steps (0: taco_model_set_batch_invariant)
  int bf3 = 1;                 // feed-forward GEMMs (both CBHGs, linear head) on the bf16 matrix cores with 3-term split operands
  int bf3x6 = 0;               // training shadow model: feed-forward GEMMs on the six-product (fp32-grade) split-bf16 instantiation
  int bf3_tn = 0;              // debug: force the bf3 tile width (1: 128x64, 2: 128x128)
  int chain = 1;               // point-wise tail of a CBHG as one launch (taco_chain.h); 0: one launch per layer
  int front_entry = 1;         // proj_1's epilogue + proj_2 inside the point-wise chain's entry (taco_chain.h) when the fused front ran
  int front_prio = 1; int front_delay = 0;      // shader clocks by which the second K half of a k_cbhg_front workgroup starts late (taco_front.h)
  int head_sweep = 1;          // wide dense layers over many rows (the linear head) as a row sweep (taco_head.h); 0: k_gemm_bf3 tiles
  int front = 1;               // conv bank -> max-pool -> proj_1 of a CBHG as one launch (taco_front.h); 0: bank and proj_1 as two k_gemm_bf3 launches
  int overlap = 0;             // >0: run the post-net feed-forward stages behind the decoder on a second stream, chunks of
                               // max(overlap,16) steps.  Measured SLOWER on MI355X (13.2 -> 14.4-17 ms @C2): off by default
  // persistent XCD-local decoder (taco_decoder_xcd.h): per-thread weight pack and the bias vectors its epilogues read
  size_t dx_fold_n = 0;        // training shadow model: elements of the GRU-1 fold buffer (k_dx_fold), addressed by the index map as NP + 1 + i
  size_t dx_spkw = 0;          // 'simple': speaker rows of the attention GRU and of the folded GRU 1, [S][DXRB_N][256] (k_dx_rowbias)
  // training shadow model: split-bf16 packs as index maps (pack_bf3): element e of the concatenated list copies parameter bf3_idx[e] - 1
  std::vector<unsigned> bf3_idx; std::vector<Bf3Seg> bf3_segs;
  int last_bptt = 0;           // the last decoder backward ran as the persistent launch (taco_debug_decoder_info out16[9])
  size_t dbx_pack = 0;         // training shadow model: the persistent BPTT kernel's rows (dbx_build_pack)
  size_t dx_pack = 0, dx_p1o_raw = 0, dx_qpack[4] = {0, 0, 0, 0}, dx_b_p1_0 = 0, dx_b_p1c = 0, dx_b_p2 = 0, dx_b_p3 = 0, dx_b_ag = 0, dx_b_ac = 0, dx_b_g1f = 0,
         dx_b_g1c = 0, dx_b_g2g = 0, dx_b_g2c = 0, dx_b_f = 0;
  int cu_count = 0;            // compute units of the device (the whole-chip persistent kernels need one workgroup per CU on 256 CUs)
  int dx_mode = 1;             // 0: launch-per-stage decoder; 1: persistent decoder when the configuration fits; 2: same, write-through exchanges
  int dx_rows = 0;             // debug: force the rows per group (1,2,4,8); 0 = smallest that covers the batch
  long long* d_trace = nullptr;   // debug: phase stamps of group 0 / member 0 (taco_debug_decoder_trace): [decoder half | scan half];
                                  // allocated on first use and kept until the model is destroyed (captured plans may hold the pointer)
  int trace_on = 0;
  int skip_scans = 0;          // timing hook (taco_debug_set_skip_scans): the recurrent scans of both CBHGs are not launched
  hipStream_t side = nullptr;  // that second stream
  std::vector<hipEvent_t> events;
  struct TrainPacks* tp = nullptr;   // set on the shadow model of a taco_train: finalize also builds the backward packs
};
static int build_train_packs(taco_model* m);   // taco_train.h

static size_t arena_put(taco_model* m, const float* src, size_t n) {
  size_t off = rup_sz(m->harena.size(), 16);
  m->harena.resize(off + rup_sz(std::max<size_t>(n, 1), 16), 0.f);
  if (src) memcpy(&m->harena[off], src, n * sizeof(float));
  return off + 1;
}
static inline const float* AP(const taco_model* m, size_t off1) { return off1 ? m->darena + (off1 - 1) : nullptr; }

// ---- required tensor list (tests/test_host.py checks it against the CPU checker) ----
static void spec_add(taco_model* m, const std::string& n, std::vector<int64_t> s) { m->spec.push_back({n, s}); }
static void spec_dense(taco_model* m, const std::string& n, int i, int o, bool bias = true) {
  spec_add(m, n + "/kernel", {i, o});
  if (bias) spec_add(m, n + "/bias", {o});
}
static void spec_conv(taco_model* m, const std::string& n, int k, int i, int o) {
  spec_add(m, n + "/kernel", {k, i, o});
  spec_add(m, n + "/bias", {o});
  for (const char* p : {"gamma", "beta", "moving_mean", "moving_variance"}) spec_add(m, n + "/" + p, {o});
}
static void spec_gru(taco_model* m, const std::string& n, int i, int h) {
  spec_add(m, n + "/gates/kernel", {i + h, 2 * h});
  spec_add(m, n + "/gates/bias", {2 * h});
  spec_add(m, n + "/candidate/kernel", {i + h, h});
  spec_add(m, n + "/candidate/bias", {h});
}
static void spec_cbhg(taco_model* m, const std::string& sc, int in_dim, int K, int C, int depth, int rnn,
                      const int* projs, int nproj, int pw) {
  for (int k = 1; k <= K; ++k) spec_conv(m, sc + "/conv_bank/conv1d_" + std::to_string(k), k, in_dim, C);
  int d = K * C;
  for (int i = 0; i < nproj; ++i) {
    spec_conv(m, sc + "/proj_" + std::to_string(i + 1), pw, d, projs[i]);
    d = projs[i];
  }
  if (d != rnn) spec_dense(m, sc + "/dense", d, rnn);
  for (int i = 0; i < depth; ++i) {
    spec_dense(m, sc + "/highway_" + std::to_string(i + 1) + "/H", rnn, rnn);
    spec_dense(m, sc + "/highway_" + std::to_string(i + 1) + "/T", rnn, rnn);
  }
  spec_gru(m, sc + "/bigru/fw", rnn, rnn);
  spec_gru(m, sc + "/bigru/bw", rnn, rnn);
}
static const char* kSpkNames[3] = {"before_highway", "encoder_rnn_init", "attention_rnn_init"};

static int build_spec(taco_model* m) {
  const taco_hparams& hp = m->hp;
  m->spec.clear();
  spec_add(m, "embedding", {hp.num_symbols, hp.embedding_size});
  const bool multi = hp.num_speakers > 1;
  const int spk = hp.speaker_embedding_size;
  if (multi) {
    if (spk != 1) spec_add(m, "speaker_embedding", {hp.num_speakers, spk});
    if (hp.model_type == 2) {
      std::vector<std::pair<std::string, int>> dims = {{kSpkNames[0], hp.enc_prenet[hp.enc_prenet_n - 1]},
                                                       {kSpkNames[1], hp.enc_rnn_size * 2},
                                                       {kSpkNames[2], hp.attention_state_size}};
      for (int i = 0; i < hp.dec_layer_num; ++i) dims.push_back({"decoder_rnn_init_" + std::to_string(i + 1), hp.dec_rnn_size});
      for (auto& d : dims) {
        if (spk == 1) spec_add(m, "spk/" + d.first + "/table", {hp.num_speakers, d.second});
        else spec_dense(m, "spk/" + d.first, spk, d.second);
      }
    }
  }
  int d = hp.embedding_size;
  for (int i = 0; i < hp.enc_prenet_n; ++i) {
    spec_dense(m, "prenet/dense_" + std::to_string(i + 1), d, hp.enc_prenet[i]);
    d = hp.enc_prenet[i];
  }
  spec_cbhg(m, "encoder_cbhg", d, hp.enc_bank_size, hp.enc_bank_channels, hp.enc_highway_depth, hp.enc_rnn_size,
            hp.enc_proj, hp.enc_proj_n, hp.enc_proj_width);
  const int enc_out = 2 * hp.enc_rnn_size, A = hp.attention_size;
  spec_dense(m, "attention/memory_layer", enc_out, A, false);
  spec_dense(m, "attention/query_layer", hp.attention_state_size, A, false);
  spec_add(m, "attention/attention_v", {A});
  if (hp.attention_type == 2) spec_add(m, "attention/attention_score_bias", {});
  if (hp.attention_type == 1) {
    spec_add(m, "attention/attention_g", {});
    spec_add(m, "attention/attention_b", {A});
  }
  d = hp.num_mels + enc_out;
  for (int i = 0; i < hp.dec_prenet_n; ++i) {
    spec_dense(m, "decoder/prenet/dense_" + std::to_string(i + 1), d, hp.dec_prenet[i]);
    d = hp.dec_prenet[i];
  }
  const int sspk = (multi && hp.model_type == 1) ? spk : 0;   // 'simple': speaker embedding concatenated (tacotron.py:82-86)
  spec_gru(m, "decoder/attention_gru", d + sspk, hp.attention_state_size);                       // rnn_wrappers.py:372-376
  spec_dense(m, "decoder/concat_projection", hp.attention_state_size + enc_out + sspk, hp.dec_rnn_size);   // :405-413
  for (int i = 0; i < hp.dec_layer_num; ++i) spec_gru(m, "decoder/gru_" + std::to_string(i + 1), hp.dec_rnn_size, hp.dec_rnn_size);
  spec_dense(m, "decoder/frame_projection", hp.dec_rnn_size, hp.num_mels * hp.reduction_factor);
  spec_cbhg(m, "post_cbhg", hp.num_mels, hp.post_bank_size, hp.post_bank_channels, hp.post_highway_depth,
            hp.post_rnn_size, hp.post_proj, hp.post_proj_n, hp.post_proj_width);
  spec_dense(m, "linear", 2 * hp.post_rnn_size + sspk, hp.num_freq);   // tacotron.py:226-235 (embedding in front)
  return 0;
}

// ---- packing ----
// W32 pack of a [kw, cin, N] kernel (dense: kw = 1): see taco_kernels.h.
static size_t pack_w32(taco_model* m, const float* W, int kw, int cin, int N, int* cin_pad_out, int* Kq_out, int* NT_out) {
  const int cin_pad = rup(cin, 8), Kq = kw * cin_pad / 4, NT = cdiv(N, 32);
  std::vector<float> p((size_t)NT * Kq * 32 * 4, 0.f);
  for (int nt = 0; nt < NT; ++nt)
    for (int kq = 0; kq < Kq; ++kq)
      for (int j = 0; j < 32; ++j)
        for (int e = 0; e < 4; ++e) {
          const int k = 4 * kq + e, tap = k / cin_pad, c = k % cin_pad, n = 32 * nt + j;
          if (c < cin && n < N) p[(((size_t)nt * Kq + kq) * 32 + j) * 4 + e] = W[((size_t)tap * cin + c) * N + n];
        }
  *cin_pad_out = cin_pad; *Kq_out = Kq; *NT_out = NT;
  return arena_put(m, p.data(), p.size());
}
// split-bf16 pack of a [kw, cin, N] kernel for k_gemm_bf3: hi = bf16(w), lo = bf16(w - hi), layout in taco_kernels.h
static unsigned short bf16_rne_host(float x) {
  unsigned u; memcpy(&u, &x, 4);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static void pack_bf3(taco_model* m, const float* W, int kw, int cin, int N, size_t* hi_out, size_t* lo_out, int* K16_out, int* cp16_out, size_t* l3_out = nullptr) {
  const int cp16 = rup(cin, 32), K16 = kw * cp16 / 16, NT = cdiv(N, 32);   // multiple of 32: k_gemm_bf3 walks k16 groups in pairs
  std::vector<unsigned short> hi((size_t)NT * K16 * 2 * 32 * 8, 0), lo(hi.size(), 0);
  std::vector<unsigned> idx(m->tp ? hi.size() : 0, 0u);
  const bool want_l3 = l3_out && !m->tp && *l3_out == (size_t)1;     // inference model, caller asks (*l3_out preset to 1) for the third plane too
  std::vector<unsigned short> l3v(want_l3 ? hi.size() : 0, 0);
  for (int nt = 0; nt < NT; ++nt)
    for (int k16 = 0; k16 < K16; ++k16)
      for (int h = 0; h < 2; ++h)
        for (int j = 0; j < 32; ++j)
          for (int e = 0; e < 8; ++e) {
            const int k = 16 * k16 + 8 * h + e, tap = k / cp16, c = k % cp16, n = 32 * nt + j;
            if (c < cin && n < N) {
              const float w = W[((size_t)tap * cin + c) * N + n];
              const size_t oi = ((((size_t)k16 * NT + nt) * 2 + h) * 32 + j) * 8 + e;
              if (m->tp) { idx[oi] = (unsigned)w; continue; }       // shadow model: `w` is 1 + a parameter index (k_bf3_gather splits the live value)
              const unsigned short hb = bf16_rne_host(w);
              unsigned hu = (unsigned)hb << 16; float hf; memcpy(&hf, &hu, 4);
              const size_t o = ((((size_t)k16 * NT + nt) * 2 + h) * 32 + j) * 8 + e;   // k16-major: see the layout note in taco_kernels.h
              hi[o] = hb; lo[o] = bf16_rne_host(w - hf);
              if (want_l3) { unsigned lu = (unsigned)lo[o] << 16; float lf; memcpy(&lf, &lu, 4); l3v[o] = bf16_rne_host(w - hf - lf); }
            }
          }
  auto put = [&](const std::vector<unsigned short>& v) {
    std::vector<float> f((v.size() + 1) / 2, 0.f);
    memcpy(f.data(), v.data(), v.size() * sizeof(unsigned short));
    return arena_put(m, f.data(), f.size());
  };
  *hi_out = put(hi); *lo_out = put(lo); *K16_out = K16; *cp16_out = cp16;
  if (want_l3) *l3_out = put(l3v);
  else if (l3_out && !m->tp) *l3_out = 0;
  if (m->tp) {
    // third plane (zeros here: k_bf3_gather fills all three from the live parameters): the operand of the six-product instantiation
    const size_t l3 = put(std::vector<unsigned short>(hi.size(), 0));
    if (l3_out) *l3_out = l3;
    m->bf3_segs.push_back(Bf3Seg{(unsigned)m->bf3_idx.size(), (unsigned)idx.size(), (unsigned long long)(*hi_out - 1), (unsigned long long)(*lo_out - 1), (unsigned long long)(l3 - 1)});
    m->bf3_idx.insert(m->bf3_idx.end(), idx.begin(), idx.end());
  }
}

// W16 pack of rows [r0, r0+K) and columns [c0, c0+N) of a row-major [*, ldw] matrix.
static SkW pack_w16(taco_model* m, const float* W, int ldw, int r0, int K, int c0, int N, const float* bias) {
  SkW s;
  s.K = K; s.N = N;
  const int Kpad = rup(K, 16), NT = cdiv(N, 16);
  s.Kq = Kpad / 4;
  std::vector<float> p((size_t)NT * s.Kq * 16 * 4, 0.f);
  for (int nt = 0; nt < NT; ++nt)
    for (int kq = 0; kq < s.Kq; ++kq)
      for (int j = 0; j < 16; ++j)
        for (int e = 0; e < 4; ++e) {
          const int k = 4 * kq + e, n = 16 * nt + j;
          if (k < K && n < N) p[(((size_t)nt * s.Kq + kq) * 16 + j) * 4 + e] = W[(size_t)(r0 + k) * ldw + c0 + n];
        }
  s.wp = arena_put(m, p.data(), p.size());
  if (bias) s.bias = arena_put(m, bias + c0, N);
  return s;
}

static const HostTensor& T_(taco_model* m, const std::string& n) { return m->raw[n]; }

static ConvL make_conv(taco_model* m, const std::string& name, bool bn, bool has_bias = true, bool bf3 = false, bool x6 = false) {
  const HostTensor& k = T_(m, name + "/kernel");
  ConvL L;
  if (k.shape.size() == 3) { L.kw = (int)k.shape[0]; L.cin = (int)k.shape[1]; L.N = (int)k.shape[2]; }
  else { L.kw = 1; L.cin = (int)k.shape[0]; L.N = (int)k.shape[1]; }
  int Kq, NT;
  L.wp = pack_w32(m, k.data.data(), L.kw, L.cin, L.N, &L.cin_pad, &Kq, &NT);
  if (bf3 && !m->tp) L.bl3 = 1;              // (preset 1 = "build the third plane in an inference model too": every layer can run the six-product instantiation, taco_debug_set_bf3 bit 64)
  if (x6) L.x6 = true;                       // ... and this one does by default
  if (bf3) pack_bf3(m, k.data.data(), L.kw, L.cin, L.N, &L.bh, &L.bl, &L.K16, &L.cin_pad16, &L.bl3);
  if (bf3 && L.kw == 1 && L.N >= 512 && L.N % 32 > 0 && L.N % 32 <= 4 && (L.cin == 256 || L.cin == 512)) {
    L.ntail = L.N % 32;
    std::vector<float> wt((size_t)L.ntail * L.cin);
    for (int t = 0; t < L.ntail; ++t)
      for (int kk = 0; kk < L.cin; ++kk) wt[(size_t)t * L.cin + kk] = k.data[(size_t)kk * L.N + (L.N - L.ntail + t)];
    L.wtail = arena_put(m, wt.data(), wt.size());
  }
  if (has_bias) L.bias = arena_put(m, T_(m, name + "/bias").data.data(), L.N);
  if (bn) {  // BatchNorm inference folded to y*scale + shift (A.2; epsilon 1e-3 = tf.layers default)
    const auto& g = T_(m, name + "/gamma").data; const auto& b = T_(m, name + "/beta").data;
    const auto& mu = T_(m, name + "/moving_mean").data; const auto& var = T_(m, name + "/moving_variance").data;
    std::vector<float> sc(L.N), sh(L.N);
    for (int i = 0; i < L.N; ++i) {
      const double s = (double)g[i] / std::sqrt((double)var[i] + 1e-3);
      sc[i] = (float)s; sh[i] = (float)((double)b[i] - (double)mu[i] * s);
    }
    L.bns = arena_put(m, sc.data(), L.N); L.bnb = arena_put(m, sh.data(), L.N);
  }
  return L;
}

static GruDec make_grudec(taco_model* m, const std::string& name, int I, int H) {
  GruDec g; g.I = I; g.H = H;
  const auto& gk = T_(m, name + "/gates/kernel").data; const auto& gb = T_(m, name + "/gates/bias").data;
  const auto& ck = T_(m, name + "/candidate/kernel").data; const auto& cb = T_(m, name + "/candidate/bias").data;
  // one [I+H, 3H] matrix: columns [0,2H) = gates/kernel, columns [2H,3H) = candidate/kernel rows of x
  // (rows of h zero: the candidate's h part needs r first and runs in the second launch)
  std::vector<float> W((size_t)(I + H) * 3 * H, 0.f), b(3 * H, 0.f);
  for (int i = 0; i < I + H; ++i) {
    for (int j = 0; j < 2 * H; ++j) W[(size_t)i * 3 * H + j] = gk[(size_t)i * 2 * H + j];
    if (i < I) for (int j = 0; j < H; ++j) W[(size_t)i * 3 * H + 2 * H + j] = ck[(size_t)i * H + j];
  }
  for (int j = 0; j < 2 * H; ++j) b[j] = gb[j];
  g.gx = pack_w16(m, W.data(), 3 * H, 0, I + H, 0, 3 * H, b.data());
  g.ch = pack_w16(m, ck.data(), H, I, H, 0, H, cb.data());    // h rows of candidate/kernel (+ candidate bias)
  return g;
}

static bool is_simple(const taco_model* m);

// ---- persistent XCD-local decoder: per-thread weight packs (mirror of the pass mapping in taco_decoder_xcd.h) ----
// The persistent decoder exists for 256-wide cells and memory, two decoder layers, and the attention widths / prenet depths of the presets of
// hparams.py:71-117: attention_size 128 / 256 / 512, dec_prenet_sizes [256, 128] or [256, 128, 64].
static int dx_prenet_depth(const taco_model* m) {
  const taco_hparams& hp = m->hp;
  if (hp.dec_prenet_n == 2 && hp.dec_prenet[0] == DX_W && hp.dec_prenet[1] == DX_P2) return 2;
  if (hp.dec_prenet_n == 3 && hp.dec_prenet[0] == DX_W && hp.dec_prenet[1] == DX_P2 && hp.dec_prenet[2] == DX_P3) return 3;
  return 0;
}
static bool dx_reference_widths(const taco_model* m) { return m->hp.attention_size == DX_W && dx_prenet_depth(m) == 2; }
static bool dx_widths_ok(const taco_model* m) {
  const taco_hparams& hp = m->hp;
  const bool aw = hp.attention_size == 128 || hp.attention_size == 256 || hp.attention_size == 512;
  return hp.attention_state_size == DX_W && hp.dec_rnn_size == DX_W && aw && 2 * hp.enc_rnn_size == DX_W && dx_prenet_depth(m) != 0 &&
         hp.dec_layer_num == 2 && hp.num_mels * hp.reduction_factor <= 16 * DX_GROUP &&
         (dx_reference_widths(m) || !m->tp);       // the training forward / BPTT kernels: reference widths only
}
// one weight column of a pass: registers reg0 .. reg0+kw-1 of every thread = W[row0 + kw*lane + e][col(wave)]  (col < 0: none)
template <class ColFn>
static void dx_fill_col(std::vector<float>& pack, int nreg, int member, int reg0, int kw, const float* W, int ldw, int row0, ColFn col) {
  for (int tid = 0; tid < DX_NT; ++tid) {
    const int wave = tid >> 6, lane = tid & 63, n = col(wave);
    if (n < 0) continue;
    for (int e = 0; e < kw; ++e)
      pack[((size_t)member * nreg + reg0 + e) * DX_NT + tid] = W[(size_t)(row0 + kw * lane + e) * ldw + n];
  }
}
// Wc/bc: composite prenet layer 1 of the next step ([o | ctx] rows); Wf/bf: GRU 1 with the concat projection folded in
// ([h_att | ctx | h1] rows x [r | u | candidate-x | o0] columns), both formed in double by taco_model_finalize
static int dx_build_pack(taco_model* m, const std::vector<float>& Wc, const std::vector<float>& bc, const std::vector<float>& Wf,
                         const std::vector<float>& bf) {
  if (!dx_widths_ok(m)) return 0;
  const taco_hparams& hp = m->hp;
  const int H = DX_W, rM = hp.num_mels * hp.reduction_factor, NCF = cdiv(rM, DX_GROUP);
  const int S = is_simple(m) ? hp.speaker_embedding_size : 0;     // 'simple': S speaker rows behind the prenet output / behind [h_att | ctx]
  const int PD = dx_prenet_depth(m), AW = hp.attention_size;
  const int IA = (PD == 3 ? DX_P3 : DX_P2) + S, Z = 2 * H + S;    // first h row of the attention GRU kernels / of the folded GRU 1 matrix
  std::vector<float> pack((size_t)DX_GROUP * DX_NREG * DX_NT, 0.f);
  const auto& W2 = T_(m, "decoder/prenet/dense_2/kernel").data;
  const auto& agk = T_(m, "decoder/attention_gru/gates/kernel").data; const auto& ack = T_(m, "decoder/attention_gru/candidate/kernel").data;
  const auto& wq = T_(m, "attention/query_layer/kernel").data;
  const auto& c1k = T_(m, "decoder/gru_1/candidate/kernel").data;
  const auto& g2k = T_(m, "decoder/gru_2/gates/kernel").data; const auto& c2k = T_(m, "decoder/gru_2/candidate/kernel").data;
  const auto& fk = T_(m, "decoder/frame_projection/kernel").data;
  for (int mem = 0; mem < DX_GROUP; ++mem) {
    auto c8 = [&](int w) { return mem * 8 + w; };
    auto cu = [&](int w) { return H + mem * 8 + w; };                       // update-gate column (A.6: gates kernel = r | u)
    auto c4 = [&](int w) { return w < 4 ? mem * 4 + w : -1; };
    auto f0 = [&](int w) { const int n = mem * NCF + w; return (w < NCF && n < rM) ? n : -1; };
    auto f1 = [&](int w) { const int n = mem * NCF + w + 8; return (w + 8 < NCF && n < rM) ? n : -1; };
    auto put = [&](int reg0, int kw, const float* W, int ldw, int row0, auto col) { dx_fill_col(pack, DX_NREG, mem, reg0, kw, W, ldw, row0, col); };
    put(DXR_P2, 4, W2.data(), DX_P2, 0, c4);
    put(DXR_AGH, 4, agk.data(), 2 * H, IA, c8); put(DXR_AGH + 4, 4, agk.data(), 2 * H, IA, cu);                  // h rows
    if (PD == 3) {      // 64 input rows, one per lane: r, u, candidate-x in one register each; behind them prenet layer 3 (p2 -> column 2m + w, waves 0-1)
      put(DXR_AGX, 1, agk.data(), 2 * H, 0, c8); put(DXR_AGX + 1, 1, agk.data(), 2 * H, 0, cu); put(DXR_AGX + 2, 1, ack.data(), H, 0, c8);
      put(DXR_AGX + 3, 2, T_(m, "decoder/prenet/dense_3/kernel").data.data(), DX_P3, 0, [&](int w) { return w < 2 ? mem * 2 + w : -1; });
    } else {
      put(DXR_AGX, 2, agk.data(), 2 * H, 0, c8); put(DXR_AGX + 2, 2, agk.data(), 2 * H, 0, cu); put(DXR_AGX + 4, 2, ack.data(), H, 0, c8);
    }
    put(DXR_AC, 4, ack.data(), H, IA, c8);
    auto fx = [&](int w) { return 2 * H + mem * 8 + w; };
    auto fo = [&](int w) { return 3 * H + mem * 8 + w; };
    put(DXR_G1H, 4, Wf.data(), 4 * H, Z, c8); put(DXR_G1H + 4, 4, Wf.data(), 4 * H, Z, cu);                        // h1 rows
    put(DXR_G1A, 4, Wf.data(), 4 * H, 0, c8); put(DXR_G1A + 4, 4, Wf.data(), 4 * H, 0, cu);                        // h_att rows
    put(DXR_G1A + 8, 4, Wf.data(), 4 * H, 0, fx); put(DXR_G1A + 12, 4, Wf.data(), 4 * H, 0, fo);
    put(DXR_G1B, 4, Wf.data(), 4 * H, H, c8); put(DXR_G1B + 4, 4, Wf.data(), 4 * H, H, cu);                        // context rows
    put(DXR_G1B + 8, 4, Wf.data(), 4 * H, H, fx); put(DXR_G1B + 12, 4, Wf.data(), 4 * H, H, fo);
    put(DXR_G1C, 4, c1k.data(), H, H, c8);
    put(DXR_G2H, 4, g2k.data(), 2 * H, H, c8); put(DXR_G2H + 4, 4, g2k.data(), 2 * H, H, cu);
    put(DXR_G2X, 4, g2k.data(), 2 * H, 0, c8); put(DXR_G2X + 4, 4, g2k.data(), 2 * H, 0, cu); put(DXR_G2X + 8, 4, c2k.data(), H, 0, c8);
    put(DXR_G2C, 4, c2k.data(), H, H, c8);
    put(DXR_P1C, 4, Wc.data(), DX_W, H, c8);                                                                        // context rows of [o | ctx]
    put(DXR_P1O, 4, Wc.data(), DX_W, 0, c8);
    put(DXR_F, 4, fk.data(), rM, 0, f0); put(DXR_F + 4, 4, fk.data(), rM, 0, f1);
  }
  m->dx_pack = arena_put(m, pack.data(), pack.size());
  if (!m->tp && dx_reference_widths(m)) {
    // teacher-forced decoding on an inference model (taco_decoder_forward with teacher frames): the TAPE instantiation reads the teacher's
    // frame with the RAW frame rows of prenet layer 1 where this pack holds the frame-projection composite -- four registers per thread
    const auto& W1 = T_(m, "decoder/prenet/dense_1/kernel").data;
    std::vector<float> raw((size_t)DX_W * DX_W, 0.f), alt((size_t)DX_GROUP * 4 * DX_NT, 0.f);
    for (int k = 0; k < hp.num_mels; ++k) for (int q = 0; q < DX_W; ++q) raw[(size_t)k * DX_W + q] = W1[(size_t)k * DX_W + q];
    for (int mem = 0; mem < DX_GROUP; ++mem) dx_fill_col(alt, 4, mem, 0, 4, raw.data(), DX_W, 0, [&](int w) { return mem * 8 + w; });
    m->dx_p1o_raw = arena_put(m, alt.data(), alt.size());
  }
  if (S) {   // speaker rows: [k][slot][n], slots in DXRB_* order
    std::vector<float> sw((size_t)S * DXRB_N * H);
    for (int k = 0; k < S; ++k)
      for (int n = 0; n < H; ++n) {
        float* o = &sw[(size_t)k * DXRB_N * H + n];
        const int px = IA - S;                                      // prenet output rows in front of the speaker rows
        o[DXRB_AR * H] = agk[(size_t)(px + k) * 2 * H + n]; o[DXRB_AU * H] = agk[(size_t)(px + k) * 2 * H + H + n];
        o[DXRB_AX * H] = ack[(size_t)(px + k) * H + n];
        const float* wf = &Wf[(size_t)(2 * H + k) * 4 * H];
        o[DXRB_G1R * H] = wf[n]; o[DXRB_G1U * H] = wf[H + n]; o[DXRB_G1X * H] = wf[2 * H + n]; o[DXRB_O0 * H] = wf[3 * H + n];
      }
    m->dx_spkw = arena_put(m, sw.data(), sw.size());
  }
  // query layer: slot s of a row (s = member % Pr) scores channel block s % Pc and holds columns (s % Pc)*DS + w*QC + i of it; one pack
  // per rows-per-group
  for (int q = 0; q < 4; ++q) {
    const int RG = 1 << q, Pr = DX_GROUP / RG, Pc = dx_score_blocks(RG), DS = AW / Pc, QC = DS / 8, QR = dx_q_regs(RG, AW);
    if (DS < 8 || DS > 64) continue;                                // (no instantiation: attention_size 128 below / 512 above 4 rows per group)
    std::vector<float> qp((size_t)DX_GROUP * QR * DX_NT, 0.f);
    for (int mem = 0; mem < DX_GROUP; ++mem)
      for (int i = 0; i < QC; ++i) {
        const int cb = (mem % Pr) % Pc;
        dx_fill_col(qp, QR, mem, 4 * i, 4, wq.data(), AW, 0, [&](int w) { return cb * DS + w * QC + i; });
      }
    m->dx_qpack[q] = arena_put(m, qp.data(), qp.size());
  }
  auto putv = [&](const std::vector<float>& v) { return arena_put(m, v.data(), v.size()); };
  m->dx_b_p1_0 = putv(T_(m, "decoder/prenet/dense_1/bias").data);
  m->dx_b_p1c = putv(bc);
  m->dx_b_p2 = putv(T_(m, "decoder/prenet/dense_2/bias").data);
  if (PD == 3) m->dx_b_p3 = putv(T_(m, "decoder/prenet/dense_3/bias").data);
  m->dx_b_ag = putv(T_(m, "decoder/attention_gru/gates/bias").data);
  m->dx_b_ac = putv(T_(m, "decoder/attention_gru/candidate/bias").data);
  m->dx_b_g1f = putv(bf);
  m->dx_b_g1c = putv(T_(m, "decoder/gru_1/candidate/bias").data);
  m->dx_b_g2g = putv(T_(m, "decoder/gru_2/gates/bias").data);
  m->dx_b_g2c = putv(T_(m, "decoder/gru_2/candidate/bias").data);
  m->dx_b_f = putv(T_(m, "decoder/frame_projection/bias").data);
  return 0;
}

// The persistent BPTT kernel's registers (taco_decoder_bwd_xcd.h, DBR_*): ROWS of the kernels -- wave w of member m holds row 8m + w
// (128-wide inputs: row 4m + w on waves 0-3) of every matrix a gradient is pulled back through, inputs 4*lane .. 4*lane + 3 per register quad.
static int dbx_build_pack(taco_model* m) {
  if (!dx_widths_ok(m)) return 0;
  const taco_hparams& hp = m->hp;
  const int H = DX_W, Mm = hp.num_mels;
  // 'simple': S speaker rows sit between the prenet rows and the h rows of the attention GRU kernels (and behind [h_att | ctx] in the
  // concat projection).  The embedding is constant over the loop, so its gradient is a sum over steps of products with those rows --
  // formed once after the launch from the time-summed pre-activation gradients (decoder_backward); the kernel only skips the rows.
  const int IA = DX_P2 + (is_simple(m) ? hp.speaker_embedding_size : 0);
  std::vector<float> pack((size_t)DX_GROUP * DB_NREG * DX_NT, 0.f);
  const auto& g2k = T_(m, "decoder/gru_2/gates/kernel").data; const auto& c2k = T_(m, "decoder/gru_2/candidate/kernel").data;
  const auto& g1k = T_(m, "decoder/gru_1/gates/kernel").data; const auto& c1k = T_(m, "decoder/gru_1/candidate/kernel").data;
  const auto& cck = T_(m, "decoder/concat_projection/kernel").data;
  const auto& wq = T_(m, "attention/query_layer/kernel").data;
  const auto& agk = T_(m, "decoder/attention_gru/gates/kernel").data; const auto& ack = T_(m, "decoder/attention_gru/candidate/kernel").data;
  const auto& W2 = T_(m, "decoder/prenet/dense_2/kernel").data; const auto& W1 = T_(m, "decoder/prenet/dense_1/kernel").data;
  for (int mem = 0; mem < DX_GROUP; ++mem)
    for (int tid = 0; tid < DX_NT; ++tid) {
      const int wave = tid >> 6, lane = tid & 63, en = mem * 8 + wave, en2 = wave < 4 ? mem * 4 + wave : -1;
      float* R = &pack[((size_t)mem * DB_NREG) * DX_NT + tid];
      // registers reg0 .. reg0 + kw - 1 = W[row][col0 + kw*lane + e] (zero past ncols)
      auto row = [&](int reg0, int kw, const std::vector<float>& W, int ldw, int r, int col0, int ncols) {
        if (r < 0) return;
        for (int e = 0; e < kw; ++e) { const int c = col0 + kw * lane + e; if (c < ncols) R[(size_t)(reg0 + e) * DX_NT] = W[(size_t)r * ldw + c]; }
      };
      row(DBR_C2X, 4, c2k, H, en, 0, H); row(DBR_C2H, 4, c2k, H, H + en, 0, H);
      row(DBR_G2X, 4, g2k, 2 * H, en, 0, 2 * H); row(DBR_G2X + 4, 4, g2k, 2 * H, en, 256, 2 * H);
      row(DBR_G2H, 4, g2k, 2 * H, H + en, 0, 2 * H); row(DBR_G2H + 4, 4, g2k, 2 * H, H + en, 256, 2 * H);
      row(DBR_C1X, 4, c1k, H, en, 0, H); row(DBR_C1H, 4, c1k, H, H + en, 0, H);
      row(DBR_G1X, 4, g1k, 2 * H, en, 0, 2 * H); row(DBR_G1X + 4, 4, g1k, 2 * H, en, 256, 2 * H);
      row(DBR_G1H, 4, g1k, 2 * H, H + en, 0, 2 * H); row(DBR_G1H + 4, 4, g1k, 2 * H, H + en, 256, 2 * H);
      row(DBR_CCA, 4, cck, H, en, 0, H); row(DBR_CCC, 4, cck, H, H + en, 0, H);
      row(DBR_Q, 4, wq, H, en, 0, H);
      row(DBR_CAX, 4, ack, H, en2, 0, H); row(DBR_CAH, 4, ack, H, IA + en, 0, H);
      row(DBR_GAX, 4, agk, 2 * H, en2, 0, 2 * H); row(DBR_GAX + 4, 4, agk, 2 * H, en2, 256, 2 * H);
      row(DBR_GAH, 4, agk, 2 * H, IA + en, 0, 2 * H); row(DBR_GAH + 4, 4, agk, 2 * H, IA + en, 256, 2 * H);
      row(DBR_P2, 2, W2, DX_P2, en, 0, DX_P2);
      row(DBR_P1C, 4, W1, H, Mm + en, 0, H);
    }
  m->dbx_pack = arena_put(m, pack.data(), pack.size());
  return 0;
}

static void cbhg_dims(Cbhg& c, int in_dim, int K, int C, int maxpool, int depth, int rnn, const int* projs, int nproj, int pw) {
  c.in_dim = in_dim; c.K = K; c.C = C; c.maxpool = maxpool; c.depth = depth; c.rnn = rnn; c.pw = pw;
  c.nproj = nproj;
  for (int i = 0; i < nproj; ++i) c.proj_dim[i] = projs[i];
}

static void make_cbhg(taco_model* m, Cbhg& c, const std::string& sc, int in_dim, int K, int C, int maxpool,
                      int depth, int rnn, const int* projs, int nproj, int pw, bool bf3 = false) {
  cbhg_dims(c, in_dim, K, C, maxpool, depth, rnn, projs, nproj, pw);
  for (int k = K; k >= 1; --k) {
    const std::string n = sc + "/conv_bank/conv1d_" + std::to_string(k);
    ConvL L = make_conv(m, n, true, true, bf3);
    c.bank.push_back(L);
    m->convs[n] = L;
  }
  for (int i = 0; i < nproj; ++i) {
    const std::string n = sc + "/proj_" + std::to_string(i + 1);
    c.proj.push_back(make_conv(m, n, true, true, bf3));
    m->convs[n] = c.proj.back();
  }
  const int last = projs[nproj - 1];
  c.has_dense = last != rnn;
  if (c.has_dense) { c.dense = make_conv(m, sc + "/dense", false, true, bf3); m->convs[sc + "/dense"] = c.dense; }
  for (int i = 0; i < depth; ++i) {
    const std::string n = sc + "/highway_" + std::to_string(i + 1);
    ConvL L = make_conv(m, n + "/H", false, true, bf3);
    ConvL Tt = make_conv(m, n + "/T", false, true, bf3);
    L.wp2 = Tt.wp; L.bias2 = Tt.bias; L.bh2 = Tt.bh; L.bl2 = Tt.bl; L.bl3_2 = Tt.bl3;
    c.hw.push_back(L);
    m->convs[n] = L;
  }
  // BiGRU: hoisted input projection [rnn, 2*(2H+H)] = [fw gates(r|u) | fw cand | bw gates | bw cand]
  const int H = rnn, I = rnn;
  std::vector<float> Wx((size_t)I * 6 * H), bx(6 * H);
  for (int dir = 0; dir < 2; ++dir) {
    const std::string n = sc + "/bigru/" + (dir ? "bw" : "fw");
    const auto& gk = T_(m, n + "/gates/kernel").data; const auto& gb = T_(m, n + "/gates/bias").data;
    const auto& ck = T_(m, n + "/candidate/kernel").data; const auto& cb = T_(m, n + "/candidate/bias").data;
    for (int i = 0; i < I; ++i) {
      for (int j = 0; j < 2 * H; ++j) Wx[(size_t)i * 6 * H + dir * 3 * H + j] = gk[(size_t)i * 2 * H + j];
      for (int j = 0; j < H; ++j) Wx[(size_t)i * 6 * H + dir * 3 * H + 2 * H + j] = ck[(size_t)i * H + j];
    }
    for (int j = 0; j < 2 * H; ++j) bx[dir * 3 * H + j] = gb[j];
    for (int j = 0; j < H; ++j) bx[dir * 3 * H + 2 * H + j] = cb[j];
    c.gh[dir] = pack_w16(m, gk.data(), 2 * H, I, H, 0, 2 * H, nullptr);
    c.ch[dir] = pack_w16(m, ck.data(), H, I, H, 0, H, nullptr);
    c.raw_gh[dir] = arena_put(m, gk.data() + (size_t)I * 2 * H, (size_t)H * 2 * H);
    c.raw_ch[dir] = arena_put(m, ck.data() + (size_t)I * H, (size_t)H * H);
    { std::vector<float> g2((size_t)H * H * 2);
      for (int k = 0; k < H; ++k)
        for (int j = 0; j < H; ++j) { g2[((size_t)k * H + j) * 2] = gk[(size_t)(I + k) * 2 * H + j]; g2[((size_t)k * H + j) * 2 + 1] = gk[(size_t)(I + k) * 2 * H + H + j]; }
      c.res_g2[dir] = arena_put(m, g2.data(), g2.size());
      if (H == 256) {
        std::vector<float> g2p(g2.size()), c1p((size_t)H * H);
        for (int k = 0; k < H; ++k)
          for (int jb = 0; jb < 64; ++jb)
            for (int u = 0; u < 4; ++u) {
              const int j = jb + 64 * u, cp = jb * 4 + u;
              g2p[((size_t)k * H + cp) * 2] = g2[((size_t)k * H + j) * 2]; g2p[((size_t)k * H + cp) * 2 + 1] = g2[((size_t)k * H + j) * 2 + 1];
              c1p[(size_t)k * H + cp] = ck[(size_t)(I + k) * H + j];
            }
        c.res_g2p[dir] = arena_put(m, g2p.data(), g2p.size());
        c.res_c1p[dir] = arena_put(m, c1p.data(), c1p.size());
        if (!m->tp) {   // k_bigru_xcd<RG, NWV>: member mem, wave w owns units 16 mem + UPW w .. + UPW - 1 (UPW = 16 / NWV); lane l holds
                        // h rows 4l..4l+3 of the columns r_0, u_0, r_1, u_1, ... and then of the candidates c_0, c_1, ...
          for (int nwv = 8; nwv >= 4; nwv -= 4) {
            const int NT = 64 * nwv, UPW = 16 / nwv, NREG = gx_nreg(nwv);
            std::vector<float> gp((size_t)GX_MEMBERS * NREG * NT, 0.f);
            for (int mem = 0; mem < GX_MEMBERS; ++mem)
              for (int tid = 0; tid < NT; ++tid) {
                const int w = tid >> 6, l = tid & 63, u0 = mem * 16 + UPW * w;
                for (int e = 0; e < 4; ++e) {
                  const size_t kr = (size_t)(I + 4 * l + e);
                  auto put = [&](int reg, float v) { gp[((size_t)mem * NREG + reg) * NT + tid] = v; };
                  for (int i = 0; i < UPW; ++i) {
                    put(8 * i + e, gk[kr * 2 * H + u0 + i]);  put(8 * i + 4 + e, gk[kr * 2 * H + H + u0 + i]);   // r_i, u_i
                    put(8 * UPW + 4 * i + e, ck[kr * H + u0 + i]);                                              // c_i
                  }
                }
              }
            (nwv == 8 ? c.gx_pack : c.gx_pack4)[dir] = arena_put(m, gp.data(), gp.size());
          }
        }
      } }
  }
  if (H == GX_H) {
    // k_bigru_duo<RG>: member mem, wave w owns unit 8 mem + w of BOTH directions; lane l holds h rows 4l..4l+3 of its r, u and c columns
    std::vector<float> gp((size_t)2 * GD_MEMBERS * 12 * 512, 0.f);
    for (int dir = 0; dir < 2; ++dir) {
      const std::string n = sc + "/bigru/" + (dir ? "bw" : "fw");
      const auto& gk = T_(m, n + "/gates/kernel").data; const auto& ck = T_(m, n + "/candidate/kernel").data;
      for (int mem = 0; mem < GD_MEMBERS; ++mem)
        for (int tid = 0; tid < 512; ++tid) {
          const int w = tid >> 6, l = tid & 63, u = mem * 8 + w;
          for (int e = 0; e < 4; ++e) {
            const size_t kr = (size_t)(I + 4 * l + e);
            float* base = &gp[(((size_t)dir * GD_MEMBERS + mem) * 12) * 512 + tid];
            base[(size_t)(0 + e) * 512] = gk[kr * 2 * H + u];
            base[(size_t)(4 + e) * 512] = gk[kr * 2 * H + H + u];
            base[(size_t)(8 + e) * 512] = ck[kr * H + u];
          }
        }
    }
    c.gd_pack = arena_put(m, gp.data(), gp.size());
    // k_bigru_oct<UPW>: member mem of a cluster of 32 / UPW, wave w owns units 8 UPW mem + UPW w + i of BOTH directions; lane l holds h rows
    // 4l..4l+3 of the columns r_0, u_0, r_1, u_1, ... (registers 8i + 4g + e) and then of the candidates c_i (8 UPW + 4i + e)
    for (int lg = 0; lg < 3; ++lg) {
      const int UPW = 1 << lg, MB = DX_GROUP / UPW, NR = 12 * UPW;
      std::vector<float> op((size_t)2 * MB * NR * 512, 0.f);
      for (int dir = 0; dir < 2; ++dir) {
        const std::string n = sc + "/bigru/" + (dir ? "bw" : "fw");
        const auto& gk = T_(m, n + "/gates/kernel").data; const auto& ck = T_(m, n + "/candidate/kernel").data;
        for (int mem = 0; mem < MB; ++mem)
          for (int tid = 0; tid < 512; ++tid) {
            const int w = tid >> 6, l = tid & 63;
            float* base = &op[(((size_t)dir * MB + mem) * NR) * 512 + tid];
            for (int i = 0; i < UPW; ++i) {
              const int u = mem * 8 * UPW + w * UPW + i;
              for (int e = 0; e < 4; ++e) {
                const size_t kr = (size_t)(I + 4 * l + e);
                base[(size_t)(8 * i + e) * 512] = gk[kr * 2 * H + u];
                base[(size_t)(8 * i + 4 + e) * 512] = gk[kr * 2 * H + H + u];
                base[(size_t)(8 * UPW + 4 * i + e) * 512] = ck[kr * H + u];
              }
            }
          }
      }
      c.go_pack[lg] = arena_put(m, op.data(), op.size());
    }
    if (m->tp) {   // k_bigru_duo_bwd (training only): ROWS of the recurrent kernels -- unit u's row of Wc_h, then of Wg_h (r half, u half)
      std::vector<float> bp((size_t)2 * GD_MEMBERS * 12 * 512, 0.f);
      for (int dir = 0; dir < 2; ++dir) {
        const std::string n = sc + "/bigru/" + (dir ? "bw" : "fw");
        const auto& gk = T_(m, n + "/gates/kernel").data; const auto& ck = T_(m, n + "/candidate/kernel").data;
        for (int mem = 0; mem < GD_MEMBERS; ++mem)
          for (int tid = 0; tid < 512; ++tid) {
            const int w = tid >> 6, l = tid & 63, u = mem * 8 + w;
            float* base = &bp[(((size_t)dir * GD_MEMBERS + mem) * 12) * 512 + tid];
            for (int e = 0; e < 4; ++e) {
              base[(size_t)(0 + e) * 512] = ck[(size_t)(I + u) * H + 4 * l + e];
              base[(size_t)(4 + e) * 512] = gk[(size_t)(I + u) * 2 * H + 4 * l + e];
              base[(size_t)(8 + e) * 512] = gk[(size_t)(I + u) * 2 * H + H + 4 * l + e];
            }
          }
      }
      c.gb_pack = arena_put(m, bp.data(), bp.size());
      // k_bigru_oct_bwd: member mem of a cluster of 8, wave w owns units 32 mem + 4 w + i; lane l holds inputs 4l..4l+3 of the unit's ROWS: registers
      // 8 p + 4 j + e = Wc_h row of unit 2 p + j (p = 0, 1); 16 + 8 i + e / 16 + 8 i + 4 + e = Wg_h row of unit i, r half / u half
      std::vector<float> ob((size_t)2 * (DX_GROUP / GOB_UPW) * 48 * 512, 0.f);
      for (int dir = 0; dir < 2; ++dir) {
        const std::string n = sc + "/bigru/" + (dir ? "bw" : "fw");
        const auto& gk = T_(m, n + "/gates/kernel").data; const auto& ck = T_(m, n + "/candidate/kernel").data;
        for (int mem = 0; mem < DX_GROUP / GOB_UPW; ++mem)
          for (int tid = 0; tid < 512; ++tid) {
            const int w = tid >> 6, l = tid & 63;
            float* base = &ob[(((size_t)dir * (DX_GROUP / GOB_UPW) + mem) * 48) * 512 + tid];
            for (int i = 0; i < GOB_UPW; ++i) {
              const int u = mem * GOB_UPM + w * GOB_UPW + i;
              for (int e = 0; e < 4; ++e) {
                base[(size_t)(8 * (i >> 1) + 4 * (i & 1) + e) * 512] = ck[(size_t)(I + u) * H + 4 * l + e];
                base[(size_t)(16 + 8 * i + e) * 512] = gk[(size_t)(I + u) * 2 * H + 4 * l + e];
                base[(size_t)(16 + 8 * i + 4 + e) * 512] = gk[(size_t)(I + u) * 2 * H + H + 4 * l + e];
              }
            }
          }
      }
      c.gob_pack = arena_put(m, ob.data(), ob.size());
    }
  }
  ConvL X; X.kw = 1; X.cin = I; X.N = 6 * H;
  int Kq, NT;
  X.wp = pack_w32(m, Wx.data(), 1, I, 6 * H, &X.cin_pad, &Kq, &NT);
  if (bf3 && !m->tp) X.bl3 = 1;
  if (bf3) pack_bf3(m, Wx.data(), 1, I, 6 * H, &X.bh, &X.bl, &X.K16, &X.cin_pad16, &X.bl3);
  X.bias = arena_put(m, bx.data(), 6 * H);
  c.xproj = X;
  // fused front (taco_front.h): inference models only (the training forward needs the bank tensor for its batch statistics)
  c.front_kind = 0;
  if (bf3 && !m->tp && maxpool == 2 && pw == 3 && nproj >= 1 && C % FR_CH == 0 && K * (C / FR_CH) <= FR_MAXCH && K <= FR_MAXW) {
    const int cinp = rup(in_dim, 16), N1 = projs[0];
    // (N1 % 32: k_front_combine and the chain's fused entry read the partial sums, bias and BatchNorm operands as float4 at n = i % N1 and
    // consume them in 32-column tiles -- a custom first projection size such as 250 takes the two-launch path)
    if (cinp == 80 && K <= 8 && N1 <= 256 && N1 % 32 == 0) c.front_kind = 1;
    else if (cinp == 128 && K <= 16 && N1 <= 128 && N1 % 32 == 0) c.front_kind = 2;
    if (c.front_kind) {
      for (const ConvL& L : c.bank) {
        const HostTensor& kt = T_(m, sc + "/conv_bank/conv1d_" + std::to_string(L.kw) + "/kernel");
        const int ns = rup(cdiv(L.kw * cinp, 32), 2), nct = C / 16;     // even: the kernel walks the steps in pairs (a padding step has zero weights)
        std::vector<unsigned short> hi((size_t)ns * nct * 512, 0), lo(hi.size(), 0);
        for (int st = 0; st < ns; ++st)
          for (int ct = 0; ct < nct; ++ct)
            for (int q = 0; q < 4; ++q)
              for (int i = 0; i < 16; ++i)
                for (int e = 0; e < 8; ++e) {
                  const int kk = 32 * st + 8 * q + e, tap = kk / cinp, cc = kk % cinp;
                  if (tap >= L.kw || cc >= in_dim) continue;
                  const float w = kt.data[((size_t)tap * in_dim + cc) * C + 16 * ct + i];
                  const unsigned short hb = bf16_rne_host(w);
                  unsigned hu = (unsigned)hb << 16; float hf; memcpy(&hf, &hu, 4);
                  const size_t o = (((size_t)st * nct + ct) * 64 + q * 16 + i) * 8 + e;
                  hi[o] = hb; lo[o] = bf16_rne_host(w - hf);
                }
        auto put = [&](const std::vector<unsigned short>& v) {
          std::vector<float> f((v.size() + 1) / 2, 0.f);
          memcpy(f.data(), v.data(), v.size() * sizeof(unsigned short));
          return arena_put(m, f.data(), f.size());
        };
        c.fr_wh.push_back(put(hi)); c.fr_wl.push_back(put(lo)); c.fr_ns.push_back(ns);
      }
    }
  }
}

static int add_var(taco_model* m, ConvL& L, int coff) {
  GemmVar v;
  memset(&v, 0, sizeof v);
  // pointers are resolved after the arena upload (see finalize): store offsets for now
  v.wp = (const float*)L.wp; v.wp2 = (const float*)L.wp2; v.bias = (const float*)L.bias; v.bias2 = (const float*)L.bias2;
  v.bn_scale = (const float*)L.bns; v.bn_shift = (const float*)L.bnb;
  v.bh = (const unsigned short*)L.bh; v.bl = (const unsigned short*)L.bl; v.bh2 = (const unsigned short*)L.bh2; v.bl2 = (const unsigned short*)L.bl2;
  v.bl3 = (const unsigned short*)L.bl3; v.bl3_2 = (const unsigned short*)L.bl3_2;
  v.K16 = L.K16; v.cin_pad16 = L.cin_pad16;
  v.kw = L.kw; v.padl = (L.kw - 1) / 2; v.Kq = L.kw * L.cin_pad / 4; v.NT = cdiv(L.N, 32); v.N = L.N; v.coff = coff;
  L.var_index = (int)m->hvars.size();
  m->hvars.push_back(v);
  return L.var_index;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
struct GemmCall {
  const float* x = nullptr; const int* gather = nullptr; int ldx = 0;
  int M = 0, T = 0, mpw = 1, act = ACT_NONE;
  const float* res = nullptr; int ldres = 0;
  const float* rowvec = nullptr; int ldrv = 0;
  const int* rev_len = nullptr; int rev_col0 = -1;
  int t_begin = 0, t_len = 0;      // time window [t_begin, t_begin + t_len) of every batch row (t_len 0 = all rows)
  float* out = nullptr; int ldo = 0;
  float* aux0 = nullptr; float* aux1 = nullptr;   // highway H / T saved for the backward pass (training tape)
};

template <int WM, int WN, int TM, int TN, int KS, bool DUAL>
static int launch_gemm_cfg(hipStream_t st, const GemmArgs& a, int nvar, int kw_max, int Nmax) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NTHR = 64 * WM * WN * KS;
  size_t lds = (size_t)(BM + kw_max - 1) * TACO_LDSW * sizeof(float);
  if (KS > 1) lds = std::max(lds, (size_t)(KS - 1) * WM * WN * TM * TN * 1024 * (DUAL ? 2 : 1) * sizeof(float));
  GemmArgs aa = a;
  int gx = cdiv(a.M, BM);
  if (a.t_len > 0) { aa.tiles_per_b = cdiv(a.t_len, BM); gx = (a.M / a.T) * aa.tiles_per_b; }
  dim3 grid(gx, cdiv(Nmax, BN), nvar);
  hipLaunchKernelGGL((k_gemm<WM, WN, TM, TN, KS, DUAL>), grid, dim3(NTHR), lds, st, aa);
  HIPCHK(hipGetLastError());
  return 0;
}

template <int WM, int WN, int TM, int TN, bool DUAL, int KS = 1, bool X6 = false>
static int launch_gemm_bf3(hipStream_t st, const GemmArgs& a, int nvar, int kw_max, int Nmax, int gpi = 0) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  size_t lds = (size_t)KS * (X6 ? 3 : 2) * (BM + 15) * BF3_LDSW * sizeof(unsigned short);
  if (KS > 1) lds = std::max(lds, (size_t)(KS - 1) * WM * WN * TM * TN * 16 * 64 * (DUAL ? 2 : 1) * sizeof(float));   // split-K reduction
  GemmArgs aa = a;
  int gx = cdiv(a.M, BM);
  if (a.t_len > 0) { aa.tiles_per_b = cdiv(a.t_len, BM); gx = (a.M / a.T) * aa.tiles_per_b; }
  (void)kw_max;
  dim3 grid(gx, cdiv(Nmax, BN), nvar);
  if constexpr (KS > 1) {
    static bool attr_set = false;      // > 64 KB of dynamic LDS has to be asked for once per kernel
    if (!attr_set) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_bf3<WM, WN, TM, TN, DUAL, 1, KS, X6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
    hipLaunchKernelGGL((k_gemm_bf3<WM, WN, TM, TN, DUAL, 1, KS, X6>), grid, dim3(64 * WM * WN * KS), lds, st, aa);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if constexpr (X6) {      // six products per k16 group: one prefetch depth only
    hipLaunchKernelGGL((k_gemm_bf3<WM, WN, TM, TN, DUAL, 1, 1, true>), grid, dim3(64 * WM * WN), lds, st, aa);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // prefetch depth: two k16 steps ahead when the grid leaves at most ~2 workgroups per CU (occupancy is grid-limited there)
  if (gpi == 0) gpi = (!DUAL && BN == 256 && (long)grid.x * grid.y * grid.z <= 320) ? 2 : 1;
  for (int i = 0; i < nvar; ++i) if (a.v[i].cin_pad16 % 64) gpi = 1;     // GPI = 2 needs four k16 steps in every chunk (even group count)
  if (gpi == 2) hipLaunchKernelGGL((k_gemm_bf3<WM, WN, TM, TN, DUAL, 2>), grid, dim3(64 * WM * WN), lds, st, aa);
  else hipLaunchKernelGGL((k_gemm_bf3<WM, WN, TM, TN, DUAL, 1>), grid, dim3(64 * WM * WN), lds, st, aa);
  HIPCHK(hipGetLastError());
  return 0;
}

static int pick_cfg(const taco_model* m, int M, int N, int nvar) {
  if (m->force_cfg >= 0) return m->force_cfg;
  // measured (tools/time_gemm_layers.py): the 64x64 tile (32 VGPRs, 8 waves/SIMD) beats 128x64 and 128x128 on
  // every large layer -- the kernel is bound by latency hiding, not operand reuse; small-M layers need split-K
  const long b1 = (long)cdiv(M, 64) * cdiv(N, 64) * nvar;
  if (b1 >= 512) return 1;
  return 2;
}

static thread_local int g_gemm_force_bf3 = 0;     // set around a call by run_dgrad (taco_train.h): this GEMM on the split-bf16 kernel although the model's switch is off
static int run_gemm(const taco_model* m, hipStream_t st, const ConvL* layers, int nvar, bool dual, const GemmCall& c) {
  GemmArgs a;
  memset(&a, 0, sizeof a);
  const ConvL& L0 = layers[0];
  a.x = c.x; a.gather = c.gather; a.res = c.res; a.rowvec = c.rowvec; a.out = c.out;
  a.ldx = c.ldx; a.M = c.M; a.T = c.T > 0 ? c.T : c.M; a.Cin = L0.cin; a.cin_pad = L0.cin_pad; a.mpw = c.mpw;
  a.act = c.act; a.ldres = c.ldres; a.ldrv = c.ldrv; a.ldo = c.ldo; a.rev_len = c.rev_len; a.rev_col0 = c.rev_col0;
  a.t_begin = c.t_begin; a.t_len = c.t_len; a.aux0 = c.aux0; a.aux1 = c.aux1;
  a.vec_ok = (c.ldx % 4 == 0) && (L0.cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(c.x) & 15) == 0);
  int kw_max = 1, Nmax = 0;
  for (int i = 0; i < nvar; ++i) {
    kw_max = std::max(kw_max, layers[i].kw); Nmax = std::max(Nmax, layers[i].N);
    if (layers[i].var_index < 0) return fail(TACO_ERR_STATE, "layer has no GemmVar");
    a.v[i] = m->hvars[layers[i].var_index];
  }
  if (nvar > 16) return fail(TACO_ERR_UNSUPPORTED, "conv bank wider than 16 is not supported");
  if ((m->bf3 || m->bf3x6 || g_gemm_force_bf3) && m->force_cfg < 0 && L0.bh) {   // split-bf16 path (every feed-forward layer of inference)
    // tiles (rows x cols): 1 = 128x64, 2 = 128x128, 3 = 64x256 (one staged 64-row tile feeds 8 MFMA column tiles)
    // measured (tools/time_gemm_layers.py): 64x256 wins when K or N is large (proj_1, linear, GRU projection), 128x64 otherwise
    const int Ktot = L0.kw * L0.cin;
    const long Meff = c.t_len > 0 ? (long)(c.M / a.T) * c.t_len : c.M;
    // small-M layers (encoder, short utterances): 64x64 tiles; if even those leave most CUs idle, four wave groups per
    // workgroup split K (tile 4 / 5).  Otherwise one row of waves side by side over the columns (tile 7: 64x256 by 1x8 waves,
    // tile 9: 64x128 by 1x4): every wave streams its OWN weight columns and all of them share the staged activation tile, so no
    // weight fragment is fetched twice by a workgroup (the 2x2 arrangements of tiles 1-3 fetch each one twice through a 16 KB L1
    // that cannot hold them: measured 20-30 % slower on every large layer, tools/time_gemm_layers.py).
    // a wide dense layer over many rows (the linear head): every workgroup keeps its 64 rows for ALL columns (taco_head.h)
    bool x6 = (m->bf3x6 && !g_gemm_force_bf3) || (m->bf3 && L0.x6);
    if (m->head_sweep && !m->bf3_tn && !x6 && nvar == 1 && !dual && L0.kw == 1 && c.mpw <= 1 && !c.gather && !c.res && !c.rev_len && c.rev_col0 < 0 &&
        c.t_len == 0 && !c.aux0 && !c.aux1 && c.act == ACT_NONE && !L0.bns && a.vec_ok && (L0.cin == 256 || L0.cin == 512) && L0.N >= 512 &&
        L0.N % 32 == L0.ntail && (L0.ntail == 0 || L0.wtail) && c.M >= 256 && (long)c.M * c.ldo < (1L << 30)) {
      HeadArgs h;
      memset(&h, 0, sizeof h);
      h.x = c.x; h.ldx = c.ldx; h.bh = a.v[0].bh; h.bl = a.v[0].bl; h.NT = a.v[0].NT; h.K16 = L0.cin / 16; h.bias = a.v[0].bias;
      h.wtail = L0.wtail ? AP(m, L0.wtail) : nullptr; h.ntail = L0.ntail; h.rowvec = c.rowvec; h.ldrv = c.ldrv; h.T = a.T;
      h.out = c.out; h.ldo = c.ldo; h.M = c.M; h.K = L0.cin; h.N = L0.N;
      if (L0.cin == 512) hipLaunchKernelGGL(k_head_sweep<512>, dim3(cdiv(c.M, HD_BM)), dim3(512), (size_t)2 * HD_BM * (512 + 8) * sizeof(unsigned short), st, h);
      else hipLaunchKernelGGL(k_head_sweep<256>, dim3(cdiv(c.M, HD_BM)), dim3(512), (size_t)2 * HD_BM * (256 + 8) * sizeof(unsigned short), st, h);
      HIPCHK(hipGetLastError());
      return 0;
    }
    int tn = m->bf3_tn;
    if (!tn) {
      const long g128 = (long)cdiv(Meff, 128) * cdiv(Nmax, 64) * nvar, g64 = (long)cdiv(Meff, 64) * cdiv(Nmax, 64) * nvar;
      if (g128 < 384 && !((Ktot >= 1024 || Nmax > 512) && (long)cdiv(Meff, 64) * cdiv(Nmax, 256) * nvar >= 192)) tn = (g64 >= 384) ? 4 : 5;
      else if (Nmax <= 128) tn = 9;
      else tn = (!dual && (long)cdiv(Meff, 64) * cdiv(Nmax, 256) * nvar <= 320) ? 10 : 7;   // few workgroups: two wave groups split K (16 waves/CU)
      // (tile 11, 128 x 256 by 1 x 8 waves -- every weight fragment meets four row tiles -- measured SLOWER on the linear head, 116 vs 85 us:
      // one workgroup of 8 waves per CU and three rounds of workgroups; selectable for A/B only)
    }
    // six-product (fp32-grade) instantiations: the training forward (taco_train_set_exact_gemm mode 4); the tiles the heuristic picks
    for (int i = 0; i < nvar; ++i) x6 = x6 && a.v[i].bl3 && (!dual || a.v[i].bl3_2);
    if (x6) {
      if (tn == 10) tn = 7;
      if (dual) {
        if (tn == 7) return launch_gemm_bf3<1, 8, 2, 1, true, 1, true>(st, a, nvar, kw_max, Nmax);
        if (tn == 9) return launch_gemm_bf3<1, 4, 2, 1, true, 1, true>(st, a, nvar, kw_max, Nmax);
        return launch_gemm_bf3<2, 2, 1, 1, true, 1, true>(st, a, nvar, kw_max, Nmax);      // (the four-wave-group tile would spill with two accumulator sets and three planes)
      }
      if (tn == 7) return launch_gemm_bf3<1, 8, 2, 1, false, 1, true>(st, a, nvar, kw_max, Nmax);
      if (tn == 9) return launch_gemm_bf3<1, 4, 2, 1, false, 1, true>(st, a, nvar, kw_max, Nmax);
      if (tn == 5) return launch_gemm_bf3<2, 2, 1, 1, false, 4, true>(st, a, nvar, kw_max, Nmax);
      return launch_gemm_bf3<2, 2, 1, 1, false, 1, true>(st, a, nvar, kw_max, Nmax);
    }
    if (dual) {
      if (tn == 7) return launch_gemm_bf3<1, 8, 2, 1, true>(st, a, nvar, kw_max, Nmax);
      if (tn == 9) return launch_gemm_bf3<1, 4, 2, 1, true>(st, a, nvar, kw_max, Nmax);
      if (tn == 5) return launch_gemm_bf3<2, 2, 1, 1, true, 4>(st, a, nvar, kw_max, Nmax);
      if (tn == 4) return launch_gemm_bf3<2, 2, 1, 1, true>(st, a, nvar, kw_max, Nmax);
      if (tn == 3) return launch_gemm_bf3<2, 2, 1, 4, true>(st, a, nvar, kw_max, Nmax);
      return tn == 2 ? launch_gemm_bf3<2, 2, 2, 2, true>(st, a, nvar, kw_max, Nmax) : launch_gemm_bf3<2, 2, 2, 1, true>(st, a, nvar, kw_max, Nmax);
    }
    if (tn == 11) return launch_gemm_bf3<1, 8, 4, 1, false>(st, a, nvar, kw_max, Nmax, 1);      // 128 x 256: every weight fragment meets four row tiles
    if (tn == 10) return launch_gemm_bf3<1, 8, 2, 1, false, 2>(st, a, nvar, kw_max, Nmax);
    if (tn == 9) return launch_gemm_bf3<1, 4, 2, 1, false>(st, a, nvar, kw_max, Nmax);
    if (tn == 7) return launch_gemm_bf3<1, 8, 2, 1, false>(st, a, nvar, kw_max, Nmax);
    if (tn == 5) return launch_gemm_bf3<2, 2, 1, 1, false, 4>(st, a, nvar, kw_max, Nmax);
    if (tn == 4) return launch_gemm_bf3<2, 2, 1, 1, false>(st, a, nvar, kw_max, Nmax);
    if (tn == 3) return launch_gemm_bf3<2, 2, 1, 4, false>(st, a, nvar, kw_max, Nmax);
    return tn == 2 ? launch_gemm_bf3<2, 2, 2, 2, false>(st, a, nvar, kw_max, Nmax) : launch_gemm_bf3<2, 2, 2, 1, false>(st, a, nvar, kw_max, Nmax);
  }
  const int cfg = pick_cfg(m, c.t_len > 0 ? (c.M / a.T) * c.t_len : c.M, Nmax, nvar);
  if (dual) {
    switch (cfg) {
      case 0: case 3: return launch_gemm_cfg<2, 2, 2, 1, 1, true>(st, a, nvar, kw_max, Nmax);
      case 1: return launch_gemm_cfg<2, 2, 1, 1, 1, true>(st, a, nvar, kw_max, Nmax);
      default: return launch_gemm_cfg<1, 2, 1, 1, 4, true>(st, a, nvar, kw_max, Nmax);
    }
  }
  switch (cfg) {
    case 0: return launch_gemm_cfg<2, 2, 2, 1, 1, false>(st, a, nvar, kw_max, Nmax);
    case 1: return launch_gemm_cfg<2, 2, 1, 1, 1, false>(st, a, nvar, kw_max, Nmax);
    case 3: return launch_gemm_cfg<2, 2, 2, 2, 1, false>(st, a, nvar, kw_max, Nmax);
    default: return launch_gemm_cfg<1, 2, 1, 1, 4, false>(st, a, nvar, kw_max, Nmax);
  }
}

// ---- skinny jobs ----
static SkJob sk_base(const taco_model* m, const SkW& w, const float* x0, int ldx0, int K0, const float* x1, int ldx1) {
  SkJob j;
  memset(&j, 0, sizeof j);
  j.x0 = x0; j.ldx0 = ldx0; j.K0 = K0; j.x1 = x1; j.ldx1 = ldx1;
  j.K = w.K; j.Kq = w.Kq; j.N = w.N; j.wp = AP(m, w.wp); j.bias = AP(m, w.bias);
  return j;
}
static SkJob sk_linear(const taco_model* m, const SkW& w, const float* x0, int ldx0, int K0, const float* x1, int ldx1,
                       int act, float* out, int ldo) {
  SkJob j = sk_base(m, w, x0, ldx0, K0, x1, ldx1);
  j.act = act; j.o0 = out; j.ldo0 = ldo;
  return j;
}
// fp32 MFMA sustains only 256 FLOP/clk per CU, so the MFMA chain of a stage is kept short by giving every
// 16-row tile of the batch its own workgroup (grid.y) instead of looping row tiles inside one.
template <int EPI>
static void launch_skinny_rt(hipStream_t st, int rowgroups, int tiles, bool vec, const SkArgs& a) {
  const bool multi = a.njobs > 1;
  const dim3 grid(tiles, rowgroups), blk(64 * SK_NW);
  if (vec && !multi) hipLaunchKernelGGL((k_skinny<1, EPI, true, false>), grid, blk, 0, st, a);
  else if (vec) hipLaunchKernelGGL((k_skinny<1, EPI, true, true>), grid, blk, 0, st, a);
  else if (!multi) hipLaunchKernelGGL((k_skinny<1, EPI, false, false>), grid, blk, 0, st, a);
  else hipLaunchKernelGGL((k_skinny<1, EPI, false, true>), grid, blk, 0, st, a);
}
// all jobs of one launch share the epilogue type
static int run_skinny(hipStream_t st, int R, SkJob* jobs, int njobs, int epi = EPI_LINEAR) {
  if (R > 64) return fail(TACO_ERR_UNSUPPORTED, "batch %d > 64 rows per device is not supported: shard the batch", R);
  SkArgs a;
  memset(&a, 0, sizeof a);
  a.R = R; a.njobs = njobs;
  int tiles = 0;
  bool vec = true;
  // float4 activation loads are legal when every row segment is 16-byte aligned and a multiple of 4 long
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  for (int i = 0; i < njobs; ++i) {
    const SkJob& j = jobs[i];
    vec = vec && (j.K0 % 4 == 0) && (j.K % 4 == 0) && (j.ldx0 % 4 == 0) && al16(j.x0) && (!j.x1 || ((j.ldx1 % 4 == 0) && al16(j.x1)));
    jobs[i].tile0 = tiles; tiles += cdiv(jobs[i].N, 16); a.j[i] = jobs[i];
  }
  const int RG = cdiv(R, 16);
  if (epi == EPI_LINEAR) launch_skinny_rt<EPI_LINEAR>(st, RG, tiles, vec, a);
  else if (epi == EPI_GRU_GATES) launch_skinny_rt<EPI_GRU_GATES>(st, RG, tiles, vec, a);
  else launch_skinny_rt<EPI_GRU_CAND>(st, RG, tiles, vec, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// One GRUCell step for a decoder GRU (A.6), two launches: gates (+ x-part of candidate) then candidate.
//   x [R, I] (ldx), h [R, H] updated in place; out_res (optional) = h' + x (ResidualWrapper, tacotron.py:172)
//   ldh: row stride of h (0 = H).  ldxc > 0: the gates launch writes [x-part of candidate | extra columns] rows of that stride into xc
//   (folded cells: the extra columns are the cell input itself, which the residual then reads from there instead of from x).
static int run_gru_cell(const taco_model* m, hipStream_t st, const GruDec& g, int R, const float* x, int ldx,
                        float* h, float* rh, float* u, float* xc, float* out_res, int ldh = 0, int ldxc = 0) {
  if (!ldh) ldh = g.H;
  const int lxc = ldxc ? ldxc : g.H;
  SkJob ja = sk_base(m, g.gx, x, ldx, g.I, h, ldh);
  ja.H = g.H; ja.e0 = h; ja.lde0 = ldh; ja.o0 = rh; ja.ldo0 = g.H; ja.o1 = u; ja.ldo1 = g.H; ja.o2 = xc; ja.ldo2 = lxc;
  TRY(run_skinny(st, R, &ja, 1, EPI_GRU_GATES));
  SkJob jb = sk_base(m, g.ch, rh, g.H, g.H, nullptr, 0);
  jb.H = g.H; jb.e0 = h; jb.lde0 = ldh; jb.e1 = xc; jb.lde1 = lxc; jb.e2 = u; jb.lde2 = g.H;
  jb.o0 = h; jb.ldo0 = ldh;
  if (out_res) {
    if (ldxc) { jb.e3 = xc + g.H; jb.lde3 = lxc; } else { jb.e3 = x; jb.lde3 = ldx; }
    jb.o1 = out_res; jb.ldo1 = g.H;
  }
  TRY(run_skinny(st, R, &jb, 1, EPI_GRU_CAND));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------------
struct Carver {
  char* base; size_t off = 0, cap;
  Carver(void* p, size_t c) : base((char*)p), cap(c) {}
  float* f(size_t n) { return (float*)raw(n * sizeof(float)); }
  int* i(size_t n) { return (int*)raw(n * sizeof(int)); }
  void* raw(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += rup_sz(bytes, 256);
    return p;
  }
  bool ok() const { return !base || off <= cap; }
};

struct CbhgWs { float *bank, *p[4], *hi0, *hi1, *xproj, *h, *rh, *u; unsigned long long* gxbuf; size_t gxbuf_bytes; unsigned* gxctl; };
static void carve_cbhg(Carver& cv, const Cbhg& c, int B, int T, CbhgWs& w) {
  const size_t M = (size_t)B * T;
  w.bank = cv.f(M * c.K * c.C);
  for (int i = 0; i < c.nproj; ++i) w.p[i] = cv.f(M * c.proj_dim[i]);
  w.hi0 = cv.f(M * c.rnn); w.hi1 = cv.f(M * c.rnn);
  w.xproj = cv.f(M * 6 * c.rnn);
  w.h = cv.f((size_t)2 * B * c.rnn); w.rh = cv.f((size_t)2 * B * c.rnn); w.u = cv.f((size_t)2 * B * c.rnn);
  w.gxbuf_bytes = gx_xbuf_granules(16, 8) * sizeof(unsigned long long);  // k_bigru_xcd exchange granules (16 groups x 8 rows = 32 groups x 4 rows; k_bigru_duo / k_bigru_oct need no more)
  w.gxbuf = (unsigned long long*)cv.raw(w.gxbuf_bytes);
  w.gxctl = (unsigned*)cv.raw(256);
}

// Row-parallel persistent BiGRU: R rows per workgroup.  Returns false when it does not fit.
static bool bigru_rows_cfg(int B, int H, int* R_out, size_t* lds_out) {
  if (H % 4) return false;
  for (int R : {2, 1}) {                                   // R = 4 spills at the 128-VGPR budget of a 1024-thread workgroup
    if (R > 1 && B < R) continue;
    if ((size_t)R * 2 * H > 2 * RP_NT) continue;          // <= 2 gate outputs per thread
    const size_t fl = (size_t)3 * R * H + (size_t)RP_NT * R * 4 + 64;
    if (fl * sizeof(float) <= 160 * 1024) { *R_out = R; *lds_out = fl * sizeof(float); return true; }
  }
  return false;
}

static size_t bigru_res_lds(int H, int KL, int R) {
  const int NQ = 512 / H;
  return ((size_t)3 * R * H + (size_t)NQ * R * 3 * H) * sizeof(float) + (size_t)KL * NQ * H * 3 * sizeof(float);
}

// k_bigru_duo (taco_bigru_xcd.h) usable for this scan?  (H = 256, a whole MI355X, at most 64 rows)
static bool duo_usable(const taco_model* m, const Cbhg& c, int B, int T) {
  return (m->persist == 1 || m->persist == 10 || m->persist == 11) && m->dx_mode && c.gd_pack && c.rnn == GX_H && B <= 64 && T >= 2 && m->cu_count >= 256;
}
// both directions of RG rows on one group of 32 CUs, software-pipelined against each other; gsave != null: the TAPE instantiation
static int duo_launch(const taco_model* m, hipStream_t st, const Cbhg& c, int B, int T, const float* xproj, const int* lengths, const float* init_state,
                      float* out, float* gsave, unsigned long long* gxbuf, unsigned* gxctl) {
  ChipTurn turn(m->device, st);
  GdArgs a; memset(&a, 0, sizeof a);
  a.wpack = AP(m, c.gd_pack); a.xproj = xproj; a.h0 = init_state; a.lengths = lengths; a.out = out; a.gsave = gsave;
  a.xbuf = gxbuf; a.ctl = gxctl; a.err = m->d_err; a.trace = (m->trace_on && m->d_trace) ? m->d_trace + DX_TRACE_STEPS * DX_TRACE_SLOTS : nullptr;
  a.B = B; a.T = T; a.force_wt = m->dx_mode == 2 ? 1 : 0;
  int RG = 1;
  while (RG * DX_NGROUP < B) RG *= 2;
  HIPCHK(clear_polled(gxbuf, (size_t)((char*)gxctl - (char*)gxbuf) + 256, st));
  const size_t lds = std::max(gd_lds_floats(RG) * sizeof(float), (size_t)96 * 1024);      // one workgroup per CU
  const dim3 grid(DX_NGROUP * GD_MEMBERS), blk(512);
  if (gsave) a.trace = nullptr;
  if (a.trace) {          // the stamped instantiations (tools/trace_bigru.py): inference only
    switch (RG) {
      case 1: hipLaunchKernelGGL((k_bigru_duo<1, false, true>), grid, blk, lds, st, a); break;
      case 2: hipLaunchKernelGGL((k_bigru_duo<2, false, true>), grid, blk, lds, st, a); break;
      case 4: hipLaunchKernelGGL((k_bigru_duo<4, false, true>), grid, blk, lds, st, a); break;
      default: hipLaunchKernelGGL((k_bigru_duo<8, false, true>), grid, blk, lds, st, a); break;
    }
  } else if (gsave) {
    switch (RG) {
      case 1: hipLaunchKernelGGL((k_bigru_duo<1, true>), grid, blk, lds, st, a); break;
      case 2: hipLaunchKernelGGL((k_bigru_duo<2, true>), grid, blk, lds, st, a); break;
      case 4: hipLaunchKernelGGL((k_bigru_duo<4, true>), grid, blk, lds, st, a); break;
      default: hipLaunchKernelGGL((k_bigru_duo<8, true>), grid, blk, lds, st, a); break;
    }
  } else {
    switch (RG) {
      case 1: hipLaunchKernelGGL((k_bigru_duo<1, false>), grid, blk, lds, st, a); break;
      case 2: hipLaunchKernelGGL((k_bigru_duo<2, false>), grid, blk, lds, st, a); break;
      case 4: hipLaunchKernelGGL((k_bigru_duo<4, false>), grid, blk, lds, st, a); break;
      default: hipLaunchKernelGGL((k_bigru_duo<8, false>), grid, blk, lds, st, a); break;
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}
// k_bigru_oct: one row per cluster of 32 / UPW CUs (UPW = 4: up to 32 rows, 2: 16, 1: 8).  persist 1 (default): from 9 rows on -- up to eight
// rows k_bigru_duo<1> IS the one-row geometry on 32 CUs --; persist 10: wherever it fits (A/B); persist 11: never (round 4's k_bigru_duo)
static int oct_upw(const taco_model* m, const Cbhg& c, int B, int T) {
  if (!(m->persist == 1 || m->persist == 10) || !m->dx_mode || !c.go_pack[0] || c.rnn != GX_H || B > 32 || T < 2 || m->cu_count < 256) return 0;
  if (B > 16) return 4;
  if (B > 8) return 2;
  return m->persist == 10 ? 1 : 0;
}
// the backward scan on the same geometry (k_bigru_oct_bwd; four units per wave: up to 32 rows), where the forward scan runs on k_bigru_oct
static bool oct_bwd_usable(const taco_model* m, const Cbhg& c, int B, int T) { return c.gob_pack && oct_upw(m, c, B, T) != 0; }
static int oct_bwd_launch(const taco_model* m, hipStream_t st, const Cbhg& c, int B, int T, const float* dout, const float* out, const float* gsave,
                          const float* h0, const int* lengths, float* dg, float* rh, float* dh0, unsigned long long* gxbuf, unsigned* gxctl) {
  ChipTurn turn(m->device, st);
#ifdef GO_KNOB_RT
  {   // A/B build: entries 12..15 of TACO_GO_KNOB are the backward scan's four second requests
    int d[16]; for (int i = 0; i < 16; ++i) d[i] = i < 12 ? go_knob_default(i) : gob_knob_default(i - 12);
    if (const char* e = getenv("TACO_GO_KNOB")) { int i = 0; const char* p = e; while (*p && i < 16) { d[i++] = atoi(p); while (*p && *p != ',') ++p; if (*p == ',') ++p; } }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_go_knob), d, sizeof d));
  }
#endif
  GbArgs a; memset(&a, 0, sizeof a);
  a.wpack = AP(m, c.gob_pack); a.dout = dout; a.out = out; a.gsave = gsave; a.h0 = h0; a.lengths = lengths; a.dg = dg; a.rh = rh; a.dh0 = dh0;
  a.xbuf = gxbuf; a.ctl = gxctl; a.err = m->d_err; a.B = B; a.T = T; a.force_wt = m->dx_mode == 2 ? 1 : 0;
  const size_t lds = std::max(gob_lds_floats() * sizeof(float), (size_t)96 * 1024);      // one workgroup per CU
  hipLaunchKernelGGL(k_bigru_oct_bwd, dim3(DX_NGROUP * DX_GROUP), dim3(512), lds, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}
static int oct_launch(const taco_model* m, hipStream_t st, const Cbhg& c, int UPW, int B, int T, const float* xproj, const int* lengths, const float* init_state,
                      float* out, float* gsave, unsigned long long* gxbuf, unsigned* gxctl) {
  ChipTurn turn(m->device, st);
#ifdef GO_KNOB_RT
  {   // A/B build: the twelve knobs of k_bigru_oct from TACO_GO_KNOB (taco_bigru_xcd.h); eager launches only
    int d[16]; for (int i = 0; i < 16; ++i) d[i] = i < 12 ? go_knob_default(i) : gob_knob_default(i - 12);
    if (const char* e = getenv("TACO_GO_KNOB")) { int i = 0; const char* p = e; while (*p && i < 16) { d[i++] = atoi(p); while (*p && *p != ',') ++p; if (*p == ',') ++p; } }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_go_knob), d, sizeof d));
  }
#endif
  GdArgs a; memset(&a, 0, sizeof a);
  a.wpack = AP(m, c.go_pack[UPW == 4 ? 2 : UPW == 2 ? 1 : 0]); a.xproj = xproj; a.h0 = init_state; a.lengths = lengths; a.out = out; a.gsave = gsave;
  a.xbuf = gxbuf; a.ctl = gxctl; a.err = m->d_err; a.trace = (m->trace_on && m->d_trace && !gsave) ? m->d_trace + DX_TRACE_STEPS * DX_TRACE_SLOTS : nullptr;
  a.B = B; a.T = T; a.force_wt = m->dx_mode == 2 ? 1 : 0;
  HIPCHK(clear_polled(gxbuf, (size_t)((char*)gxctl - (char*)gxbuf) + 256, st));
  const size_t lds = std::max(go_lds_floats(UPW) * sizeof(float), (size_t)96 * 1024);      // one workgroup per CU
  const dim3 grid(DX_NGROUP * DX_GROUP), blk(512);
  if (a.trace) {          // the stamped instantiation (tools/trace_bigru.py): inference only
    switch (UPW) {
      case 1: hipLaunchKernelGGL((k_bigru_oct<1, false, true>), grid, blk, lds, st, a); break;
      case 2: hipLaunchKernelGGL((k_bigru_oct<2, false, true>), grid, blk, lds, st, a); break;
      default: hipLaunchKernelGGL((k_bigru_oct<4, false, true>), grid, blk, lds, st, a); break;
    }
  } else if (gsave) {
    switch (UPW) {
      case 1: hipLaunchKernelGGL((k_bigru_oct<1, true>), grid, blk, lds, st, a); break;
      case 2: hipLaunchKernelGGL((k_bigru_oct<2, true>), grid, blk, lds, st, a); break;
      default: hipLaunchKernelGGL((k_bigru_oct<4, true>), grid, blk, lds, st, a); break;
    }
  } else {
    switch (UPW) {
      case 1: hipLaunchKernelGGL((k_bigru_oct<1, false>), grid, blk, lds, st, a); break;
      case 2: hipLaunchKernelGGL((k_bigru_oct<2, false>), grid, blk, lds, st, a); break;
      default: hipLaunchKernelGGL((k_bigru_oct<4, false>), grid, blk, lds, st, a); break;
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// BiGRU (modules.py:82-96 -> TF bidirectional_dynamic_rnn, A.7): hoisted x.[Wg_x|Wc_x]+b for both
// directions as one GEMM, then T sequential steps of two launches (gates; candidate+update), both
// directions side by side in each launch.  x [B*T, rnn], out [B*T, 2*rnn].
static int bigru_scan(const taco_model* m, hipStream_t st, const Cbhg& c, int B, int T,
                      const int* lengths, const float* init_state, float* out, const CbhgWs& w) {
  const int H = c.rnn;
  if (m->skip_scans) return 0;
  if (const int upw = oct_upw(m, c, B, T)) return oct_launch(m, st, c, upw, B, T, w.xproj, lengths, init_state, out, nullptr, w.gxbuf, w.gxctl);
  if (duo_usable(m, c, B, T)) return duo_launch(m, st, c, B, T, w.xproj, lengths, init_state, out, nullptr, w.gxbuf, w.gxctl);
  if ((m->persist == 8 || m->persist == 9) && m->dx_mode && c.gx_pack[0] && H == GX_H && B <= 64 && T >= 2 && m->cu_count >= 256) {
    // the 2B chains spread over the whole chip, recurrent weights stationary in registers (taco_bigru_xcd.h): 256 workgroups of 8
    // waves, one per CU.  persist 9: 512 workgroups of 4 waves, two per CU from independent chains -- measured slower (6088 vs 5224
    // clocks per step at C2): the phases of a step are chains of dependent instructions, a wave alone on its SIMD is no faster
    const int NWV = m->persist == 9 ? 4 : 8, rowgroups = gx_ngroups(NWV) / 2;     // persist 8: round 2's default geometry
    ChipTurn turn(m->device, st);
    GxArgs a; memset(&a, 0, sizeof a);
    const size_t* pk = NWV == 8 ? c.gx_pack : c.gx_pack4;
    a.wpack0 = AP(m, pk[0]); a.wpack1 = AP(m, pk[1]); a.xproj = w.xproj; a.h0 = init_state; a.lengths = lengths; a.out = out;
    a.xbuf = w.gxbuf; a.ctl = w.gxctl; a.err = m->d_err; a.trace = (m->trace_on && m->d_trace) ? m->d_trace + DX_TRACE_STEPS * DX_TRACE_SLOTS : nullptr; a.B = B; a.T = T; a.force_wt = m->dx_mode == 2 ? 1 : 0;
    int RG = 1;
    while (RG * rowgroups < B) RG *= 2;
    // granules and census words are carved back to back: one fill launch covers both
    HIPCHK(zero_async(w.gxbuf, (size_t)((char*)w.gxctl - (char*)w.gxbuf) + 256, st));
    // the kernel needs only a few KB of LDS and < 110 VGPRs: the LDS request is what fixes the number of workgroups a CU takes
    // (one of 96 KB, or two of 72 KB), so that all of them are resident at once (the census checks placement per XCD, not per CU)
    const size_t lds = std::max(gx_lds_floats(RG) * sizeof(float), (size_t)(NWV == 8 ? 96 : 72) * 1024);
    const dim3 grid(gx_ngroups(NWV) * GX_MEMBERS), blk(64 * NWV);
    if (NWV == 8) {
      switch (RG) {
        case 1: hipLaunchKernelGGL((k_bigru_xcd<1, 8>), grid, blk, lds, st, a); break;
        case 2: hipLaunchKernelGGL((k_bigru_xcd<2, 8>), grid, blk, lds, st, a); break;
        case 4: hipLaunchKernelGGL((k_bigru_xcd<4, 8>), grid, blk, lds, st, a); break;
        default: hipLaunchKernelGGL((k_bigru_xcd<8, 8>), grid, blk, lds, st, a); break;
      }
    } else {
      switch (RG) {
        case 1: hipLaunchKernelGGL((k_bigru_xcd<1, 4>), grid, blk, lds, st, a); break;
        case 2: hipLaunchKernelGGL((k_bigru_xcd<2, 4>), grid, blk, lds, st, a); break;
        default: hipLaunchKernelGGL((k_bigru_xcd<4, 4>), grid, blk, lds, st, a); break;
      }
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  if ((m->persist == 1 || (m->persist >= 4 && m->persist <= 7)) && H == 256) {
    // weights resident on the CU, 4 hidden units per thread: k_bigru_resw 3.74 us/step; its predecessor k_bigru_resu (persist 6)
    // 5.16, one unit per thread (k_bigru_res, persist 3) 5.7, re-streaming everything (k_bigru_rows, persist 2) 7.1
    BigruSArgs a; memset(&a, 0, sizeof a);
    a.xproj = w.xproj; a.g2_0 = (const float2*)AP(m, c.res_g2[0]); a.g2_1 = (const float2*)AP(m, c.res_g2[1]);
    a.c1_0 = AP(m, c.raw_ch[0]); a.c1_1 = AP(m, c.raw_ch[1]); a.h0 = init_state; a.lengths = lengths; a.out = out; a.B = B; a.T = T;
    auto lds = [&](int UJ, int KL) { const int NQ = 512 / (H / UJ); return ((size_t)3 * H + (size_t)NQ * 3 * H) * sizeof(float) + (size_t)KL * NQ * H * 12; };
    if (m->persist == 1 || m->persist == 7) {      // wave-local exchanges, a thread's four units adjacent in the (column-permuted) packs
      a.g2_0 = (const float2*)AP(m, c.res_g2p[0]); a.g2_1 = (const float2*)AP(m, c.res_g2p[1]); a.c1_0 = AP(m, c.res_c1p[0]); a.c1_1 = AP(m, c.res_c1p[1]);
      hipLaunchKernelGGL((k_bigru_resw<16, 4, 2>), dim3(2 * B), dim3(512), lds(4, 4), st, a);
    }
    else if (m->persist == 4) hipLaunchKernelGGL((k_bigru_resu<256, 4, 12, 5, 3, false>), dim3(2 * B), dim3(512), lds(4, 5), st, a);
    else if (m->persist == 5) hipLaunchKernelGGL((k_bigru_resu<256, 2, 32, 10, 2, false>), dim3(2 * B), dim3(512), lds(2, 10), st, a);
    else hipLaunchKernelGGL((k_bigru_resu<256, 4, 16, 4, 2, false>), dim3(2 * B), dim3(512), lds(4, 4), st, a);      // persist 6: the predecessor
    HIPCHK(hipGetLastError());
    return 0;
  }
  if ((m->persist == 1 || m->persist == 3) && (H == 256 || H == 128)) {
    // weights resident on the CU (registers + LDS), one batch row per workgroup: no re-streaming of the recurrent kernels
    BigruSArgs a; memset(&a, 0, sizeof a);
    a.xproj = w.xproj; a.g2_0 = (const float2*)AP(m, c.res_g2[0]); a.g2_1 = (const float2*)AP(m, c.res_g2[1]);
    a.c1_0 = AP(m, c.raw_ch[0]); a.c1_1 = AP(m, c.raw_ch[1]); a.h0 = init_state; a.lengths = lengths; a.out = out; a.B = B; a.T = T;
    const dim3 grid(2 * B);
    if (H == 256) hipLaunchKernelGGL((k_bigru_res<256, 64, 24, 1, false>), grid, dim3(512), bigru_res_lds(256, 24, 1), st, a);
    else if (m->persist == 1) hipLaunchKernelGGL(k_bigru_quad<false>, grid, dim3(512), 0, st, a);       // quad-local K split: two barriers per step (persist 3: k_bigru_res)
    else hipLaunchKernelGGL((k_bigru_res<128, 32, 0, 1, false>), grid, dim3(512), bigru_res_lds(128, 0, 1), st, a);
    HIPCHK(hipGetLastError());
    return 0;
  }
  {  // row-parallel persistent kernel: the whole scan in one launch, weights streamed from L2 every step
    int R = 0; size_t lds = 0;
    if (m->persist && bigru_rows_cfg(B, H, &R, &lds)) {   // persist == 2: this kernel even where the resident one applies (A/B tests)
      BigruRArgs a; memset(&a, 0, sizeof a);
      a.xproj = w.xproj; a.wg0 = AP(m, c.raw_gh[0]); a.wg1 = AP(m, c.raw_gh[1]); a.wc0 = AP(m, c.raw_ch[0]); a.wc1 = AP(m, c.raw_ch[1]);
      a.h0 = init_state; a.lengths = lengths; a.out = out; a.B = B; a.T = T; a.H = H;
      const dim3 grid(2 * cdiv(B, R));
      if (R == 2) hipLaunchKernelGGL((k_bigru_rows<2, false>), grid, dim3(RP_NT), lds, st, a);
      else hipLaunchKernelGGL((k_bigru_rows<1, false>), grid, dim3(RP_NT), lds, st, a);
      HIPCHK(hipGetLastError());
      return 0;
    }
  }
  for (int dir = 0; dir < 2; ++dir)
    hipLaunchKernelGGL(k_copy2d, dim3(cdiv(B * H, 256)), dim3(256), 0, st, init_state ? init_state + dir * H : nullptr,
                       2 * H, w.h + (size_t)dir * B * H, H, B, H);
  HIPCHK(hipGetLastError());
  for (int s = 0; s < T; ++s) {
    SkJob ja[2], jb[2];
    for (int dir = 0; dir < 2; ++dir) {
      float* h = w.h + (size_t)dir * B * H; float* rh = w.rh + (size_t)dir * B * H; float* u = w.u + (size_t)dir * B * H;
      ja[dir] = sk_base(m, c.gh[dir], h, H, H, nullptr, 0);
      ja[dir].H = H; ja[dir].e0 = h; ja[dir].lde0 = H;
      ja[dir].e1 = w.xproj + dir * 3 * H; ja[dir].lde1 = 6 * H;
      ja[dir].o0 = rh; ja[dir].ldo0 = H; ja[dir].o1 = u; ja[dir].ldo1 = H;
      ja[dir].step = s; ja[dir].T = T; ja[dir].dir = dir;
      jb[dir] = sk_base(m, c.ch[dir], rh, H, H, nullptr, 0);
      jb[dir].H = H; jb[dir].e0 = h; jb[dir].lde0 = H;
      jb[dir].e1 = w.xproj + dir * 3 * H + 2 * H; jb[dir].lde1 = 6 * H; jb[dir].e2 = u; jb[dir].lde2 = H;
      jb[dir].o0 = h; jb[dir].ldo0 = H; jb[dir].o2 = out; jb[dir].ldo2 = 2 * H; jb[dir].seq_coff = dir * H;
      jb[dir].lengths = lengths; jb[dir].step = s; jb[dir].T = T; jb[dir].dir = dir;
    }
    TRY(run_skinny(st, B, ja, 2, EPI_GRU_GATES));
    TRY(run_skinny(st, B, jb, 2, EPI_GRU_CAND));
  }
  return 0;
}

// ---- the point-wise tail of a CBHG as one launch (taco_chain.h) ----
static bool chain_fits(const Cbhg& c, int in_dim) {
  const int W = c.rnn;
  if (W != 128 && W != 256) return false;
  if ((int)c.hw.size() != c.depth || c.depth + (c.has_dense ? 2 : 1) > CH_MAXL) return false;
  if (c.has_dense) { if (!c.dense.bh || c.dense.cin != in_dim || c.dense.cin_pad16 > W || c.dense.N != W) return false; }
  else if (in_dim != W) return false;
  for (const ConvL& h : c.hw) if (!h.bh || !h.bh2 || h.cin != W || h.N != W) return false;
  return c.xproj.bh && c.xproj.cin == W && c.xproj.N == 6 * W;
}
static int run_chain(const taco_model* m, hipStream_t st, const Cbhg& c, const float* x, int in_dim, int M, int T, const int* lengths,
                     const CbhgWs& w, const float** ff_out, const ChainEntry* entry = nullptr) {
  ChainArgs a; memset(&a, 0, sizeof a);
  if (entry) a.e = *entry;
  a.x = x; a.ldx = in_dim; a.Cin = in_dim; a.out = w.xproj; a.ldo = 6 * c.rnn; a.rev_len = lengths; a.rev_col0 = 3 * c.rnn;
  a.M = M; a.T = T; a.krot = m->ff_rot;
  if (ff_out) { a.y = w.hi0; a.ldy = c.rnn; *ff_out = w.hi0; }
  auto add = [&](const ConvL& L, int type) {
    const GemmVar& v = m->hvars[L.var_index];
    ChainLayer& l = a.L[a.nlayers++];
    l.bh = v.bh; l.bl = v.bl; l.bh2 = v.bh2; l.bl2 = v.bl2; l.bias = v.bias; l.bias2 = v.bias2;
    l.type = type; l.K16 = v.K16; l.NT = v.NT; l.N = v.N; l.act = ACT_NONE;
  };
  if (c.has_dense) add(c.dense, CH_DENSE);
  for (int i = 0; i < c.depth; ++i) add(c.hw[i], CH_HIGHWAY);
  add(c.xproj, CH_XPROJ);
  const dim3 grid(cdiv(M, CH_BM)), blk(512);
  size_t lds = (size_t)2 * CH_BM * (c.rnn + 8) * sizeof(unsigned short);
  if (entry) lds += (size_t)2 * (CH_BM + 2) * (entry->N1 + 8) * sizeof(unsigned short);      // planes of proj_1's output (+ 1 frame each side)
  if (c.rnn == 256) hipLaunchKernelGGL((k_pointwise_chain<256>), grid, blk, lds, st, a);
  else hipLaunchKernelGGL((k_pointwise_chain<128>), grid, blk, lds, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- conv bank -> max-pool -> proj_1 as one launch (taco_front.h) ----
static bool front_usable(const taco_model* m, const Cbhg& c) {
  return m->bf3 && !m->bf3x6 && m->front && m->force_cfg < 0 && !m->bf3_tn && c.front_kind != 0 && !c.proj.empty() && c.proj[0].bh &&
         c.proj[0].cin == c.K * c.C && c.proj[0].cin_pad16 == c.K * c.C;
}
static int run_front(const taco_model* m, hipStream_t st, const Cbhg& c, const float* x, int B, int T, const CbhgWs& w, bool combine, int* P_out) {
  FrArgs a; memset(&a, 0, sizeof a);
  const int TN = c.front_kind == 1 ? 2 : 1, cinp = c.front_kind == 1 ? 80 : 128, kwmax = c.front_kind == 1 ? 8 : 16;
  const ConvL& P1 = c.proj[0];
  const GemmVar& pv = m->hvars[P1.var_index];
  a.x = x; a.ldx = c.in_dim; a.B = B; a.T = T; a.Cin = c.in_dim; a.tiles_per_b = cdiv(T, FR_BM);
  a.ph = pv.bh; a.pl = pv.bl; a.pNT = pv.NT; a.pK16tap = P1.cin_pad16 / 16;
  a.part = w.bank; a.N1 = P1.N;
  a.delay = m->front_delay;
  a.prio = m->front_prio;
  const int cpw = c.C / FR_CH, nchunks = c.K * cpw, padlmax = (kwmax - 1) / 2;
  for (int i = 0; i < c.K; ++i) {
    const ConvL& L = c.bank[i];
    FrWidth& fw = a.w[i];
    fw.wh = (const unsigned short*)AP(m, c.fr_wh[i]); fw.wl = (const unsigned short*)AP(m, c.fr_wl[i]);
    fw.bias = AP(m, L.bias); fw.scale = AP(m, L.bns); fw.shift = AP(m, L.bnb);
    fw.kw = L.kw; fw.xoff = padlmax - (L.kw - 1) / 2; fw.ns = c.fr_ns[i]; fw.nct = c.C / 16;
  }
  // parts: as many as fill the chip (the makespan is rounds of workgroups x the work of one part), at most what the partial-sum
  // slabs (kept in the bank buffer) and the chunk count allow; chunks go to the least loaded part, heaviest first
  const int ntiles = B * a.tiles_per_b, ncu = m->cu_count > 0 ? m->cu_count : 256;
  const int pmax = std::max(1, std::min(std::min(nchunks, FR_MAXP), (c.K * c.C) / P1.N));
  int P = 1; double best = 1e30;
  for (int p = 1; p <= pmax; ++p) {
    const double cost = (double)cdiv(ntiles * p, ncu) * ((double)cdiv(nchunks, p) / nchunks + 0.03);
    if (cost < best - 1e-9) { best = cost; P = p; }
  }
  a.P = P;
  { std::vector<std::pair<long, int>> order;        // (cost, chunk id = width index * cpw + sub-chunk)
    for (int i = 0; i < c.K; ++i)
      for (int j = 0; j < cpw; ++j) order.push_back({432L * c.fr_ns[i] + 4608L * TN, i * cpw + j});
    std::stable_sort(order.begin(), order.end(), [](const std::pair<long, int>& u, const std::pair<long, int>& v) { return u.first > v.first; });
    std::vector<long> load(P, 0); std::vector<std::vector<int>> mine(P);
    for (auto& o : order) {
      int tgt = 0;
      for (int p = 1; p < P; ++p) if (load[p] < load[tgt]) tgt = p;
      load[tgt] += o.first; mine[tgt].push_back(o.second);
    }
    int n = 0;
    for (int p = 0; p < P; ++p) {
      a.pstart[p] = n;
      std::sort(mine[p].begin(), mine[p].end());
      for (int id : mine[p]) {
        const int i = id / cpw, j = id % cpw;
        a.ch[n].wi = i; a.ch[n].ct0 = j * (FR_CH / 16); a.ch[n].cg0 = (c.bank[i].kw - 1) * c.C + j * FR_CH; ++n;
      }
    }
    a.pstart[P] = n; }
  const int xs = c.front_kind == 1 ? 80 : 144, xrows = c.front_kind == 1 ? 16 * FR_NRT + kwmax : FR_PR + kwmax;
  (void)cinp;
  if ((size_t)2 * xrows * xs * 2 + (size_t)2 * FR_PR * FR_ALD * 2 > FR_SMEM_CNT || (size_t)8 * 2 * TN * 16 * 64 * sizeof(float) > FR_SMEM_CNT)
    return fail(TACO_ERR_STATE, "k_cbhg_front: LDS layout does not fit");
  const size_t lds = 160 * 1024;           // planes + reduction scratch below FR_SMEM_CNT, the two group-barrier counters at it
  const dim3 grid(ntiles * P), blk(512);
  if (c.front_kind == 1) hipLaunchKernelGGL((k_cbhg_front<2, 80, 80, 8>), grid, blk, lds, st, a);      // (> 64 KB of LDS: attribute set at finalize)
  else hipLaunchKernelGGL((k_cbhg_front<1, 144, 128, 16>), grid, blk, lds, st, a);
  HIPCHK(hipGetLastError());
  if (P_out) *P_out = P;
  if (!combine) return 0;                  // the chain kernel's fused entry sums the parts itself (taco_chain.h)
  const size_t MN = (size_t)B * T * P1.N;
  hipLaunchKernelGGL(k_front_combine, dim3((unsigned)cdiv((int)(MN / 4), 256)), dim3(256), 0, st, (const float*)w.bank, P, MN, P1.N,
                     AP(m, P1.bias), AP(m, P1.bns), AP(m, P1.bnb), c.proj.size() > 1 ? 1 : 0, w.p[0]);
  HIPCHK(hipGetLastError());
  return 0;
}

// Feed-forward part of a CBHG (modules.py:27-77) with per-stage watermarks, so it can run chunk by chunk
// behind a producer of its input frames (the decoder): every stage is advanced as far as the frames
// available to it allow (a conv needs its right halo; the last chunk gets TF's zero padding).
struct FfProg { int w_bank = 0, w_p[4] = {0, 0, 0, 0}, w_pt = 0; };
static int cbhg_ff_advance(const taco_model* m, hipStream_t st, const Cbhg& c, const float* x, int B, int T,
                           const int* lengths, const float* before_highway, const CbhgWs& w, FfProg& pg, int avail,
                           const float** ff_out) {
  const int M = B * T;
  auto right = [](int k) { return k - 1 - (k - 1) / 2; };
  auto advance = [&](int win, int reach) { return win >= T ? T : std::max(0, win - reach); };
  // the whole window at once: bank -> max-pool -> proj_1 as ONE launch, the bank tensor never leaves the CU (taco_front.h)
  // ... and with it proj_1's epilogue and proj_2 (+ residual) move into the entry of the point-wise chain: front + chain = the whole
  // feed-forward part of a CBHG in two launches
  bool entry = false; int parts = 1;
  if (avail >= T && pg.w_bank == 0 && pg.w_p[0] == 0 && front_usable(m, c)) {
    const bool chain_next = m->chain && c.proj.size() == 2 && chain_fits(c, c.proj[1].N);
    const ConvL& P2 = c.proj.back();
    entry = chain_next && m->front_entry && P2.bh && P2.kw == 3 && P2.cin == c.proj[0].N && P2.cin_pad16 == c.proj[0].N &&
            c.proj[0].N == c.rnn && (c.rnn == 128 || c.rnn == 256) && P2.N <= c.rnn && c.in_dim == P2.N;     // (k16 steps of proj_2: 3 rnn / 16 = a multiple of the entry's block pairs)
    TRY(run_front(m, st, c, x, B, T, w, !entry, &parts));
    pg.w_bank = T; pg.w_p[0] = T;
    if (entry) pg.w_p[1] = T;
  }
  // conv bank: all K widths in one launch, written channel-concatenated (modules.py:35-44)
  { const int nw = advance(avail, right(c.K));
    if (nw > pg.w_bank) {
      GemmCall g; g.x = x; g.ldx = c.in_dim; g.M = M; g.T = T; g.act = ACT_RELU; g.out = w.bank; g.ldo = c.K * c.C;
      g.t_begin = pg.w_bank; g.t_len = nw - pg.w_bank;
      // (c.bank is ordered widest first, so the longest workgroups are dispatched first; narrowest first measured 0.47 vs 0.445 ms
      // for the encoder stage)
      TRY(run_gemm(m, st, c.bank.data(), c.K, false, g));
      pg.w_bank = nw;
    } }
  // maxpool (fused into the staging of proj_1) + projections (modules.py:47-59)
  const float* cur = w.bank; int curd = c.K * c.C;
  for (size_t i = 0; i < c.proj.size(); ++i) {
    const int win = (i == 0) ? pg.w_bank : pg.w_p[i - 1];
    const int nw = advance(win, right(c.pw) + ((i == 0) ? right(c.maxpool) : 0));
    if (nw > pg.w_p[i]) {
      GemmCall p;
      p.x = cur; p.ldx = curd; p.M = M; p.T = T; p.mpw = (i == 0) ? c.maxpool : 1;
      p.act = (i + 1 == c.proj.size()) ? ACT_NONE : ACT_RELU;
      p.out = w.p[i]; p.ldo = c.proj[i].N;
      p.t_begin = pg.w_p[i]; p.t_len = nw - pg.w_p[i];
      if (i + 1 == c.proj.size()) {  // residual (modules.py:62-69)
        p.res = x; p.ldres = c.in_dim;
        p.rowvec = before_highway; p.ldrv = c.in_dim;
      }
      TRY(run_gemm(m, st, &c.proj[i], 1, false, p));
      pg.w_p[i] = nw;
    }
    cur = w.p[i]; curd = c.proj[i].N;
  }
  // point-wise chain: optional dense (modules.py:72-73), highway x depth (:76-77), hoisted BiGRU input projection
  const int wlast = pg.w_p[c.proj.size() - 1];
  const int t0 = pg.w_pt, tl = wlast - pg.w_pt;
  if (entry && !(m->chain && t0 == 0 && tl == T && chain_fits(c, curd))) return fail(TACO_ERR_STATE, "fused chain entry planned but the chain kernel does not apply");
  if (m->bf3 && !m->bf3x6 && m->chain && m->force_cfg < 0 && !m->bf3_tn && t0 == 0 && tl == T && chain_fits(c, curd)) {
    // the whole tail as ONE launch, activations resident on the CU from layer to layer (taco_chain.h)
    ChainEntry E; memset(&E, 0, sizeof E);
    if (entry) {
      const ConvL& P1 = c.proj[0]; const ConvL& P2 = c.proj[1];
      const GemmVar& v2 = m->hvars[P2.var_index];
      E.part = w.bank; E.P = parts; E.MN = (size_t)M * P1.N; E.N1 = P1.N;
      E.b1 = AP(m, P1.bias); E.s1 = AP(m, P1.bns); E.h1 = AP(m, P1.bnb); E.relu1 = 1;
      E.bh = v2.bh; E.bl = v2.bl; E.NT2 = v2.NT; E.K16tap = P2.cin_pad16 / 16; E.N2 = P2.N;
      E.b2 = AP(m, P2.bias); E.s2 = AP(m, P2.bns); E.h2 = AP(m, P2.bnb);
      E.res = x; E.ldres = c.in_dim; E.rowvec = before_highway; E.ldrv = c.in_dim;
    }
    TRY(run_chain(m, st, c, cur, curd, M, T, lengths, w, ff_out, entry ? &E : nullptr));
    pg.w_pt = wlast;
    return 0;
  }
  if (c.has_dense) {
    if (tl > 0) { GemmCall d; d.x = cur; d.ldx = curd; d.M = M; d.T = T; d.out = w.hi0; d.ldo = c.rnn; d.t_begin = t0; d.t_len = tl;
      TRY(run_gemm(m, st, &c.dense, 1, false, d)); }
    cur = w.hi0;
  }
  float* bufs[2] = {w.hi0, w.hi1};
  int sel = (cur == w.hi0) ? 1 : 0;
  for (int i = 0; i < c.depth; ++i) {
    if (tl > 0) { GemmCall h; h.x = cur; h.ldx = c.rnn; h.M = M; h.T = T; h.out = bufs[sel]; h.ldo = c.rnn; h.t_begin = t0; h.t_len = tl;
      TRY(run_gemm(m, st, &c.hw[i], 1, true, h)); }
    cur = bufs[sel]; sel ^= 1;
  }
  if (tl > 0) {  // backward-direction columns are stored time-reversed per row (reverse_sequence), so scan step s reads row s
    const int H = c.rnn;
    GemmCall xp; xp.x = cur; xp.ldx = c.rnn; xp.M = M; xp.T = T; xp.out = w.xproj; xp.ldo = 6 * H;
    xp.rev_len = lengths; xp.rev_col0 = 3 * H; xp.t_begin = t0; xp.t_len = tl;
    TRY(run_gemm(m, st, &c.xproj, 1, false, xp));
    pg.w_pt = wlast;
  }
  if (ff_out) *ff_out = cur;
  return 0;
}

// modules.py:27-96.  x [B*T, in_dim], out [B*T, 2*rnn].
static int cbhg_forward(const taco_model* m, hipStream_t st, const Cbhg& c, const float* x, int B, int T,
                        const int* lengths, const float* before_highway, const float* init_state, float* out,
                        const CbhgWs& w) {
  FfProg pg;
  TRY(cbhg_ff_advance(m, st, c, x, B, T, lengths, before_highway, w, pg, T, nullptr));
  return bigru_scan(m, st, c, B, T, lengths, init_state, out, w);
}

// ---- speaker conditioning (tacotron.py:41-94) ----
struct SpkWs { float *emb, *vec[8]; };
static void carve_spk(Carver& cv, const taco_model* m, int B, SpkWs& w) {
  const taco_hparams& hp = m->hp;
  w.emb = cv.f((size_t)B * std::max(hp.speaker_embedding_size, 1));
  const int dims[3] = {hp.enc_prenet[hp.enc_prenet_n - 1], hp.enc_rnn_size * 2, hp.attention_state_size};
  for (int i = 0; i < 3 + hp.dec_layer_num; ++i) w.vec[i] = cv.f((size_t)B * (i < 3 ? dims[i] : hp.dec_rnn_size));
}
static bool is_deepvoice(const taco_model* m) { return m->hp.num_speakers > 1 && m->hp.model_type == 2; }
static bool is_simple(const taco_model* m) { return m->hp.num_speakers > 1 && m->hp.model_type == 1; }
static int simple_S(const taco_model* m) { return is_simple(m) ? m->hp.speaker_embedding_size : 0; }
// computes vec[0..2+L) = before_highway, encoder_rnn_init, attention_rnn_init, decoder_rnn_init_i
static int spk_forward(const taco_model* m, hipStream_t st, const int* speaker_id, int B, const SpkWs& w) {
  const taco_hparams& hp = m->hp;
  const int nv = 3 + hp.dec_layer_num;
  const int dims[3] = {hp.enc_prenet[hp.enc_prenet_n - 1], hp.enc_rnn_size * 2, hp.attention_state_size};
  if (hp.speaker_embedding_size == 1) {
    for (int i = 0; i < nv; ++i) {
      const int D = i < 3 ? dims[i] : hp.dec_rnn_size;
      hipLaunchKernelGGL(k_gather_rows, dim3(cdiv(B * D, 256)), dim3(256), 0, st, AP(m, m->spk_table[i]), speaker_id, B, D, w.vec[i]);
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  const int S = hp.speaker_embedding_size;
  hipLaunchKernelGGL(k_gather_rows, dim3(cdiv(B * S, 256)), dim3(256), 0, st, AP(m, m->spk_emb), speaker_id, B, S, w.emb);
  HIPCHK(hipGetLastError());
  for (int i0 = 0; i0 < nv; i0 += SK_MAXJOBS) {
    SkJob js[SK_MAXJOBS];
    const int nj = std::min(SK_MAXJOBS, nv - i0);
    for (int i = 0; i < nj; ++i) {
      const int D = (i0 + i) < 3 ? dims[i0 + i] : hp.dec_rnn_size;
      js[i] = sk_linear(m, m->spk_dense[i0 + i], w.emb, S, S, nullptr, 0, ACT_SOFTSIGN, w.vec[i0 + i], D);
    }
    TRY(run_skinny(st, B, js, nj));
  }
  return 0;
}

// ---- encoder (tacotron.py:34-112) ----
struct EncWs { float* pre[4]; CbhgWs cb; SpkWs spk; };
static void carve_enc(Carver& cv, const taco_model* m, int B, int T, EncWs& w) {
  for (int i = 0; i < m->hp.enc_prenet_n; ++i) w.pre[i] = cv.f((size_t)B * T * m->hp.enc_prenet[i]);
  carve_cbhg(cv, m->enc, B, T, w.cb);
  carve_spk(cv, m, B, w.spk);
}
// The encoder prenet (modules.py:18-25: two dense + ReLU layers over the looked-up embeddings) as ONE launch of the point-wise chain kernel
// (taco_chain.h): gathered rows -> planes, a 256-wide ReLU layer, and the second layer as the chain's last link (stored straight to the
// prenet output).  Two k_gemm_bf3 launches of 11 us each at C2 otherwise -- a 64-row tile of 4096 rows leaves them latency-bound.
static bool prenet_chain_fits(const taco_model* m) {
  const taco_hparams& hp = m->hp;
  if (!(m->bf3 && !m->bf3x6 && m->chain && m->force_cfg < 0 && !m->bf3_tn) || hp.enc_prenet_n != 2) return false;
  const ConvL& a = m->enc_prenet[0]; const ConvL& b = m->enc_prenet[1];
  return a.bh && b.bh && a.kw == 1 && b.kw == 1 && a.N == 256 && a.cin == hp.embedding_size && a.cin_pad16 <= 256 && (hp.embedding_size & 3) == 0 &&
         b.cin == 256 && b.N == hp.enc_prenet[1] && a.var_index >= 0 && b.var_index >= 0;
}
// riders (nullable): regions the launch's spare workgroups clear -- the words the persistent kernels of the same forward poll (forward_enqueue)
static int run_prenet_chain(const taco_model* m, hipStream_t st, const int* ids, int M, float* out, const ZeroRegions* riders) {
  const taco_hparams& hp = m->hp;
  ChainArgs a; memset(&a, 0, sizeof a);
  a.x = AP(m, m->emb); a.ldx = hp.embedding_size; a.Cin = hp.embedding_size; a.gather = ids;
  a.out = out; a.ldo = hp.enc_prenet[1]; a.rev_len = nullptr; a.rev_col0 = -1; a.M = M; a.T = M; a.krot = m->ff_rot;
  const int types[2] = {CH_DENSE, CH_XPROJ};
  for (int i = 0; i < 2; ++i) {
    const GemmVar& v = m->hvars[m->enc_prenet[i].var_index];
    ChainLayer& l = a.L[a.nlayers++];
    l.bh = v.bh; l.bl = v.bl; l.bh2 = v.bh2; l.bl2 = v.bl2; l.bias = v.bias; l.bias2 = v.bias2;
    l.type = types[i]; l.K16 = v.K16; l.NT = v.NT; l.N = v.N; l.act = ACT_RELU;
  }
  int nwg = cdiv(M, CH_BM);
  if (riders) {
    a.ntiles = nwg;
    for (int i = 0; i < 4; ++i) { a.zp[i] = riders->p[i]; a.znw[i] = riders->nw[i]; }
    nwg += std::min(192, std::max(16, 256 - nwg));
  }
  hipLaunchKernelGGL((k_pointwise_chain<256>), dim3(nwg), dim3(512), (size_t)2 * CH_BM * (256 + 8) * sizeof(unsigned short), st, a);
  HIPCHK(hipGetLastError());
  if (riders) register_cleared(*riders);      // the riders are in the stream now: the persistent kernels behind this launch need no fill of their own
  return 0;
}
// riders: see run_prenet_chain; only passed where prenet_chain_fits(m) (the caller clears the regions itself otherwise)
static int encoder_forward(const taco_model* m, hipStream_t st, const int* ids, const int* lengths, const int* speaker_id,
                           int B, int T, float* enc_out, const EncWs& w, bool spk_done, const ZeroRegions* riders = nullptr) {
  const taco_hparams& hp = m->hp;
  const int M = B * T;
  if (is_deepvoice(m) && !spk_done) TRY(spk_forward(m, st, speaker_id, B, w.spk));
  const float* cur = AP(m, m->emb); int curd = hp.embedding_size;
  if (prenet_chain_fits(m)) {
    TRY(run_prenet_chain(m, st, ids, M, w.pre[1], riders));
    return cbhg_forward(m, st, m->enc, w.pre[1], B, T, lengths, is_deepvoice(m) ? w.spk.vec[0] : nullptr,
                        is_deepvoice(m) ? w.spk.vec[1] : nullptr, enc_out, w.cb);
  }
  for (int i = 0; i < hp.enc_prenet_n; ++i) {  // embedding_lookup fused as a row gather (tacotron.py:38-39)
    GemmCall g; g.x = cur; g.ldx = curd; g.gather = (i == 0) ? ids : nullptr; g.M = M; g.act = ACT_RELU;
    g.out = w.pre[i]; g.ldo = hp.enc_prenet[i];
    TRY(run_gemm(m, st, &m->enc_prenet[i], 1, false, g));
    cur = w.pre[i]; curd = hp.enc_prenet[i];
  }
  return cbhg_forward(m, st, m->enc, cur, B, T, lengths, is_deepvoice(m) ? w.spk.vec[0] : nullptr,
                      is_deepvoice(m) ? w.spk.vec[1] : nullptr, enc_out, w.cb);
}

static int dx_rows_per_group(const taco_model* m, int B) {
  // the other presets (attention_size 128 / 512, three prenet layers) are instantiated for 4 and 8 rows per group only: smaller batches run
  // as padded groups of 4 (a step costs the same there: its time is the chain of exchanges, not the rows)
  const int lo = dx_reference_widths(m) ? 1 : 4;
  if (m->dx_rows == 1 || m->dx_rows == 2 || m->dx_rows == 4 || m->dx_rows == 8) { if (m->dx_rows * DX_NGROUP >= B && m->dx_rows >= lo) return m->dx_rows; }
  int RG = lo;
  while (RG < 8 && RG * DX_NGROUP < B) RG *= 2;
  return RG;
}
// ---- decoder (tacotron.py:120-214) ----
struct DecWs {
  float *keys, *zero, *ctx, *pz[4], *h_att, *rh, *u, *xc, *q, *align, *o[5], *hd[4], *Y, *escr;
  int* nz;
  unsigned long long* xbuf; size_t xbuf_bytes; unsigned* dxctl;   // persistent decoder: exchange granules, census words
  float* rowbias;                                                 // persistent decoder, 'simple': [B, DXRB_N, 256]
  SpkWs spk;
};
static void carve_dec(Carver& cv, const taco_model* m, int B, int T_in, int n, DecWs& w) {
  const taco_hparams& hp = m->hp;
  const int D = 2 * hp.enc_rnn_size, As = hp.attention_state_size, Hd = hp.dec_rnn_size;
  const int Hmax = std::max(As, Hd);
  w.keys = cv.f((size_t)B * T_in * hp.attention_size);
  w.zero = cv.f((size_t)B * hp.num_mels);
  const int S = simple_S(m);     // 'simple': the speaker embedding rides in the last S columns of ctx and of the prenet output
  // [h_att | ctx (| spk)] share one row (stride As + D + S): the concat projection's input is then ONE contiguous segment
  w.h_att = cv.f((size_t)B * (As + D + S)); w.ctx = w.h_att ? w.h_att + As : nullptr;
  for (int i = 0; i < hp.dec_prenet_n; ++i) w.pz[i] = cv.f((size_t)B * (hp.dec_prenet[i] + S));
  w.rh = cv.f((size_t)B * Hmax); w.u = cv.f((size_t)B * Hmax); w.xc = cv.f((size_t)B * 2 * Hmax);
  w.q = cv.f((size_t)B * hp.attention_size);
  w.align = cv.f((size_t)B * T_in);
  w.escr = cv.f((size_t)B * T_in);
  for (int i = 0; i <= hp.dec_layer_num; ++i) w.o[i] = cv.f((size_t)B * Hd);
  for (int i = 0; i < hp.dec_layer_num; ++i) w.hd[i] = cv.f((size_t)B * Hd);
  w.Y = nullptr;
  w.nz = cv.i((size_t)n * B);
  carve_spk(cv, m, B, w.spk);
  {  // persistent decoder (taco_decoder_xcd.h): sized for the largest rows-per-group, so a debug override cannot outgrow it
    w.xbuf_bytes = (size_t)DX_NGROUP * dx_xlayout(8, T_in).total * sizeof(unsigned long long);
    w.xbuf = (unsigned long long*)cv.raw(w.xbuf_bytes);
    w.dxctl = (unsigned*)cv.raw(256);
    w.rowbias = cv.f(is_simple(m) ? (size_t)B * DXRB_N * DX_W : 1);
  }
}
// persistent decoder: usable for this call?  (reference widths, no teacher forcing, the member's LDS fits)
static bool dx_usable(const taco_model* m, int B, int T_in, const float* manual, const float* teacher) {
  // 256 workgroups, one per CU, all resident at once: only on a whole MI355X (a partition of it -- CPX / DPX modes -- or a smaller
  // part would leave workgroups waiting for CUs held by workgroups that wait for them)
  (void)manual;   // manual alignments are a mode of the persistent kernel (the score phases are skipped)
  // teacher forcing needs the raw prenet rows in the registers where inference keeps the frame-projection composite: only the
  // training shadow model's pack has them (taco_model_finalize)
  if (!m->dx_mode || !m->dx_pack || B > 8 * DX_NGROUP || m->cu_count < DX_NGROUP * DX_GROUP) return false;
  // teacher forcing: the TAPE instantiation -- the training shadow model (pack in teacher form), or an inference model at the reference widths
  // (raw frame rows of prenet layer 1 kept beside the composite pack), with or without manual alignments
  if (teacher && !(m->tp || (m->dx_p1o_raw && m->hp.num_mels <= DX_P2))) return false;
  const int RG = dx_rows_per_group(m, B);
  if (m->hp.attention_size == 512 && RG > 4) return false;        // 128 score channels per member: no instantiation (query registers, the q / v slots)
  return dx_lds_floats(RG, T_in, m->tp != nullptr || teacher != nullptr, m->hp.attention_size) * sizeof(float) <= 160 * 1024;
}
template <int RG, bool TAPE, int AW = DX_W, int PD = 2>
static int dx_launch_rg(hipStream_t st, const DxArgs& a_in, size_t lds) {
  DxArgs a = a_in;
  if constexpr (!TAPE && AW == DX_W && PD == 2) {      // the stamped instantiation (tools/time_decoder.py): the plain decoder at the reference widths only
    if (a.trace && !a.manual) { hipLaunchKernelGGL((k_decoder_xcd<RG, false, false, AW, PD, true>), dim3(DX_NGROUP * DX_GROUP), dim3(DX_NT), lds, st, a); HIPCHK(hipGetLastError()); return 0; }
  }
  a.trace = nullptr;
  // manual alignments: their own instantiation (the score phases are compiled out); with teacher frames as well (round 6: teacher-forced decoding under
  // manual alignments on an inference model -- taco_decoder_forward with both -- used to fall to the launch-per-stage loop)
  if (a.manual) { hipLaunchKernelGGL((k_decoder_xcd<RG, TAPE, true, AW, PD>), dim3(DX_NGROUP * DX_GROUP), dim3(DX_NT), lds, st, a); HIPCHK(hipGetLastError()); return 0; }
  hipLaunchKernelGGL((k_decoder_xcd<RG, TAPE, false, AW, PD>), dim3(DX_NGROUP * DX_GROUP), dim3(DX_NT), lds, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}
// `tape`: null (inference), or the TAPE instantiation's extra arguments already filled in (teacher, tape pointers; training forward)
static int dx_launch(const taco_model* m, hipStream_t st, const float* enc_out, const int* speaker_id, const float* spk_rows, int B, int T_in, int n,
                     const float* manual, float* mel, float* align_out, float* dbg, int dbgw, const float* keys, int* nz,
                     unsigned long long* xbuf, unsigned* dxctl, float* rowbias, const float* h_att0, const float* h10, const float* h20,
                     const DxArgs* tape = nullptr) {
  ChipTurn turn(m->device, st);
#ifdef DX_DLY_RT
  {   // A/B build: the eleven first-poll sleeps (units of 64 clocks) from TACO_DX_DLY = "a,b,c,..." (sites 0..10 of taco_decoder_xcd.h); eager launches only
    int d[32]; const int dflt[11] = {DX_POLL_DELAY_B, DX_FIRST_POLL_DELAY, DX_FIRST_POLL_DELAY, DX_FIRST_POLL_DELAY, DX_POLL_DELAY_B, DX_FIRST_POLL_DELAY,
                                     DX_FIRST_POLL_DELAY, DX_POLL_DELAY_B, DX_FIRST_POLL_DELAY, DX_POLL_DELAY_B, DX_FIRST_POLL_DELAY};
    const int RGs = dx_rows_per_group(m, B);
    for (int i = 0; i < 32; ++i) d[i] = (i & 15) < 11 ? dx_site_delay(i & 15, dflt[i & 15], RGs) : 0;
    if (const char* e = getenv("TACO_DX_DLY")) { int i = 0; const char* p = e; while (*p && i < 11) { d[i] = d[16 + i] = atoi(p); ++i; while (*p && *p != ',') ++p; if (*p == ',') ++p; } }
    if (const char* e = getenv("TACO_DX_DLY_B")) { int i = 0; const char* p = e; while (*p && i < 11) { d[16 + i] = atoi(p); ++i; while (*p && *p != ',') ++p; if (*p == ',') ++p; } }      // waves 4-7
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_dx_dly), d, sizeof d));
  }
#endif
  const int RG = dx_rows_per_group(m, B);
  DxArgs a; memset(&a, 0, sizeof a);
  if (tape) a = *tape;
  a.manual = manual;
  if (is_simple(m)) {   // the speaker embedding's share of the attention GRU and GRU 1 pre-activations, once per launch
    hipLaunchKernelGGL(k_dx_rowbias, dim3(B), dim3(DX_W), 0, st, spk_rows ? spk_rows : AP(m, m->spk_emb), spk_rows ? (const int*)nullptr : speaker_id,
                       AP(m, m->dx_spkw), m->hp.speaker_embedding_size, rowbias);
    HIPCHK(hipGetLastError());
    a.rowbias = rowbias;
  }
  a.wpack = AP(m, m->dx_pack);
  a.qpack = AP(m, m->dx_qpack[RG == 1 ? 0 : RG == 2 ? 1 : RG == 4 ? 2 : 3]);
  a.b_p1_0 = AP(m, m->dx_b_p1_0); a.b_p1c = AP(m, m->dx_b_p1c); a.b_p2 = AP(m, m->dx_b_p2); a.b_p3 = AP(m, m->dx_b_p3); a.b_ag = AP(m, m->dx_b_ag); a.b_ac = AP(m, m->dx_b_ac);
  a.b_g1f = AP(m, m->dx_b_g1f); a.b_g1c = AP(m, m->dx_b_g1c); a.b_g2g = AP(m, m->dx_b_g2g); a.b_g2c = AP(m, m->dx_b_g2c);
  a.b_f = AP(m, m->dx_b_f);
  a.att_v = AP(m, m->att_v); a.att_b = AP(m, m->att_b); a.score_bias = AP(m, m->att_sb);
  a.keys = keys; a.values = enc_out; a.h_att0 = h_att0; a.h10 = h10; a.h20 = h20;
  a.mel = mel; a.hist = align_out; a.nz = nz; a.dbg = dbg; a.dbgw = dbgw;
  a.xbuf = xbuf; a.ctl = dxctl; a.err = m->d_err; a.trace = m->trace_on ? m->d_trace : nullptr;
  if (a.trace) { const char* e = getenv("TACO_TRACE_MEMBER"); a.trc_member = e ? atoi(e) : 0; e = getenv("TACO_TRACE_TID"); a.trc_tid = e ? atoi(e) : 0; }
  a.B = B; a.T_in = T_in; a.n = n; a.rM = m->hp.num_mels * m->hp.reduction_factor; a.att_type = m->hp.attention_type;
  a.mels = m->hp.num_mels;
  a.grp0 = 0; a.ngroups = cdiv(B, RG); a.force_wt = m->dx_mode == 2 ? 1 : 0;
  // every polled word starts from zero on every launch (tags are step numbers, the census counts arrivals)
  HIPCHK(clear_polled(xbuf, (size_t)((char*)dxctl - (char*)xbuf) + 256, st));   // carved back to back: one fill launch
  const size_t lds = dx_lds_floats(RG, T_in, tape != nullptr, m->hp.attention_size) * sizeof(float);
  if (!dx_reference_widths(m)) {      // the presets of hparams.py:71-117 that the reference ships switched off: (attention_size, prenet layers)
    const int aw = m->hp.attention_size, pd = dx_prenet_depth(m);
    if (tape) return fail(TACO_ERR_UNSUPPORTED, "the persistent training forward exists at the reference widths only");
    if (aw == 128 && pd == 2) return RG == 4 ? dx_launch_rg<4, false, 128, 2>(st, a, lds) : dx_launch_rg<8, false, 128, 2>(st, a, lds);      // "Single Speaker"
    if (aw == 256 && pd == 3) return RG == 4 ? dx_launch_rg<4, false, 256, 3>(st, a, lds) : dx_launch_rg<8, false, 256, 3>(st, a, lds);      // "Single Speaker with generalization"
    if (aw == 512 && pd == 3 && RG == 4) return dx_launch_rg<4, false, 512, 3>(st, a, lds);                                                   // "Deep Voice 2" (the first, disabled block)
    if (aw == 128 && pd == 3) return RG == 4 ? dx_launch_rg<4, false, 128, 3>(st, a, lds) : dx_launch_rg<8, false, 128, 3>(st, a, lds);
    if (aw == 512 && pd == 2 && RG == 4) return dx_launch_rg<4, false, 512, 2>(st, a, lds);
    return fail(TACO_ERR_STATE, "no persistent decoder instantiation for attention_size %d, %d prenet layers, %d rows per group", aw, pd, RG);
  }
  if (tape) {
    switch (RG) {
      case 1: return dx_launch_rg<1, true>(st, a, lds);
      case 2: return dx_launch_rg<2, true>(st, a, lds);
      case 4: return dx_launch_rg<4, true>(st, a, lds);
      default: return dx_launch_rg<8, true>(st, a, lds);
    }
  }
  switch (RG) {
    case 1: return dx_launch_rg<1, false>(st, a, lds);
    case 2: return dx_launch_rg<2, false>(st, a, lds);
    case 4: return dx_launch_rg<4, false>(st, a, lds);
    default: return dx_launch_rg<8, false>(st, a, lds);
  }
}
// persistent BPTT (taco_decoder_bwd_xcd.h): same placement rules as the forward kernel
static size_t dbx_xbuf_granules(int T_in) {
  size_t g = 0;
  for (int RG = 1; RG <= 8; RG *= 2) g = std::max(g, (size_t)db_xlayout(RG, T_in).total);
  return g;
}
static bool dbx_usable(const taco_model* m, int B, int T_in) {
  if (!m->dx_mode || !m->dbx_pack || !m->tp || B > 8 * DX_NGROUP || m->cu_count < DX_NGROUP * DX_GROUP) return false;
  return db_lds_floats(dx_rows_per_group(m, B), T_in) * sizeof(float) <= 160 * 1024;
}
static int dbx_launch(const taco_model* m, hipStream_t st, DbArgs a, int B, int T_in, int n, unsigned long long* xbuf, unsigned* dxctl) {
  ChipTurn turn(m->device, st);
#ifdef DX_DLY_RT
  {   // A/B build: the twelve first-poll sleeps of the backward loop from TACO_DB_DLY (sites 0..11 of taco_decoder_bwd_xcd.h)
    int d[16];
    for (int i = 0; i < 16; ++i) d[i] = i < 12 ? db_site_delay(i) : 0;
    if (const char* e = getenv("TACO_DB_DLY")) { int i = 0; const char* p = e; while (*p && i < 12) { d[i++] = atoi(p); while (*p && *p != ',') ++p; if (*p == ',') ++p; } }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_db_dly), d, sizeof d));
  }
#endif
  const int RG = dx_rows_per_group(m, B);
  a.wpack = AP(m, m->dbx_pack);
  a.att_v = AP(m, m->att_v); a.att_b = AP(m, m->att_b); a.score_bias = AP(m, m->att_sb);
  a.xbuf = xbuf; a.ctl = dxctl; a.err = m->d_err; a.trace = (m->trace_on && m->d_trace) ? m->d_trace + 2 * DX_TRACE_STEPS * DX_TRACE_SLOTS : nullptr;
  a.B = B; a.T_in = T_in; a.n = n; a.att_type = m->hp.attention_type;
  a.force_wt = m->dx_mode == 2 ? 1 : 0;
  HIPCHK(zero_async(xbuf, (size_t)((char*)dxctl - (char*)xbuf) + 256, st));
  const size_t lds = db_lds_floats(RG, T_in) * sizeof(float);
  const dim3 grid(DX_NGROUP * DX_GROUP), blk(DX_NT);
  if (a.trace && RG == 4) {      // the stamped instantiation (tools/trace_bptt.py): the C4 shard's geometry only
    hipLaunchKernelGGL((k_decoder_bwd_xcd<4, true>), grid, blk, lds, st, a);
    HIPCHK(hipGetLastError());
    return 0;
  }
  a.trace = nullptr;
  switch (RG) {
    case 1: hipLaunchKernelGGL((k_decoder_bwd_xcd<1>), grid, blk, lds, st, a); break;
    case 2: hipLaunchKernelGGL((k_decoder_bwd_xcd<2>), grid, blk, lds, st, a); break;
    case 4: hipLaunchKernelGGL((k_decoder_bwd_xcd<4>), grid, blk, lds, st, a); break;
    default: hipLaunchKernelGGL((k_decoder_bwd_xcd<8>), grid, blk, lds, st, a); break;
  }
  HIPCHK(hipGetLastError());
  return 0;
}
static int decoder_forward(const taco_model* m, hipStream_t st, const float* enc_out, const int* speaker_id, int B,
                           int T_in, int n, const float* manual, const float* teacher, float* mel, float* align_out,
                           int* stop_step, float* dbg, const DecWs& w, bool spk_ready, const SpkWs* spk_in,
                           const std::function<int(int)>* after_step = nullptr) {
  const taco_hparams& hp = m->hp;
  const int D = 2 * hp.enc_rnn_size, As = hp.attention_state_size, Hd = hp.dec_rnn_size, A = hp.attention_size;
  const int Mm = hp.num_mels, rM = hp.num_mels * hp.reduction_factor, L = hp.dec_layer_num;
  if (T_in > ATT_MAXT) return fail(TACO_ERR_UNSUPPORTED, "T_in %d > %d", T_in, ATT_MAXT);
  if ((A % 4) || (D % 4)) return fail(TACO_ERR_UNSUPPORTED, "attention_size and 2*enc_rnn_size must be multiples of 4");
  const SpkWs* spk = spk_in ? spk_in : &w.spk;
  if (is_deepvoice(m) && !spk_ready) TRY(spk_forward(m, st, speaker_id, B, *spk));
  if (is_simple(m) && !speaker_id) return fail(TACO_ERR_ARG, "speaker_id required for a multi-speaker model");
  // attention memory: keys = values . W_mem, no bias, no length mask (A.8)
  { GemmCall g; g.x = enc_out; g.ldx = D; g.M = B * T_in; g.out = w.keys; g.ldo = A;
    TRY(run_gemm(m, st, &m->memory_layer, 1, false, g)); }
  // initial state (rnn_wrappers.py:186-216; tacotron.py:183-197)
  auto fill = [&](const float* src, int lds, float* dst, int C) {
    hipLaunchKernelGGL(k_copy2d, dim3(cdiv(B * C, 256)), dim3(256), 0, st, src, lds, dst, C, B, C);
  };
  const bool dv = is_deepvoice(m);
  const int S = simple_S(m), ldc = As + D + S, np = hp.dec_prenet_n;   // ldc: row stride of [h_att | ctx | spk]
  const int Ilast = hp.dec_prenet[np - 1], ldz = Ilast + S;
  const int ldY = n * rM;   // mel buffer viewed as Y [B, n, r*num_mels] (tacotron.py:213-214 is a pure reshape)
  const int dbgw = As + D + L * Hd;
  HIPCHK(clear_polled(w.nz, (size_t)n * B * sizeof(int), st));
  if (dx_usable(m, B, T_in, manual, teacher) && !after_step) {
    // the whole loop as ONE persistent launch (taco_decoder_xcd.h), which builds its initial state itself (zeros or the deepvoice
    // vectors); the launch-per-stage loop below is the general path
    DxArgs ta; memset(&ta, 0, sizeof ta);
    if (teacher) { ta.teacher = teacher; ta.p1o_raw = m->tp ? nullptr : AP(m, m->dx_p1o_raw); }     // teacher-forced: the TAPE instantiation without a tape
    TRY(dx_launch(m, st, enc_out, speaker_id, nullptr, B, T_in, n, manual, mel, align_out, dbg, dbgw, w.keys, w.nz, w.xbuf, w.dxctl, w.rowbias,
                  dv ? spk->vec[2] : nullptr, dv ? spk->vec[3] : nullptr, dv ? spk->vec[4] : nullptr, teacher ? &ta : nullptr));
    if (stop_step) {
      hipLaunchKernelGGL(k_stop_step, dim3(1), dim3(1024), 0, st, w.nz, B, n, stop_step);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }
  fill(nullptr, 0, w.zero, Mm);
  hipLaunchKernelGGL(k_copy2d, dim3(cdiv(B * D, 256)), dim3(256), 0, st, (const float*)nullptr, 0, w.ctx, ldc, B, D);
  if (S) {  // speaker_embed = embedding_lookup(table, speaker_id) (tacotron.py:44-49), parked behind ctx and behind the prenet output
    hipLaunchKernelGGL(k_gather_rows, dim3(cdiv(B * S, 256)), dim3(256), 0, st, AP(m, m->spk_emb), speaker_id, B, S, spk->emb);
    hipLaunchKernelGGL(k_copy2d, dim3(cdiv(B * S, 256)), dim3(256), 0, st, spk->emb, S, w.ctx + D, ldc, B, S);
    hipLaunchKernelGGL(k_copy2d, dim3(cdiv(B * S, 256)), dim3(256), 0, st, spk->emb, S, w.pz[np - 1] + Ilast, ldz, B, S);
  }
  hipLaunchKernelGGL(k_copy2d, dim3(cdiv(B * As, 256)), dim3(256), 0, st, dv ? (const float*)spk->vec[2] : (const float*)nullptr, As, w.h_att, ldc, B, As);
  for (int i = 0; i < L; ++i) fill(dv ? spk->vec[3 + i] : nullptr, Hd, w.hd[i], Hd);
  hipLaunchKernelGGL(k_init_align, dim3(cdiv(B * T_in, 256)), dim3(256), 0, st, w.align, B, T_in, hp.attention_type == 2 ? 1 : 0);
  HIPCHK(hipGetLastError());
  for (int t = 0; t < n; ++t) {
    // frame fed to the prenet: zeros at t=0 (helpers.py:70-72), else last of the r frames (helpers.py:31)
    const float* frame; int ldf;
    if (t == 0) { frame = w.zero; ldf = Mm; }
    else if (teacher) { frame = teacher + (size_t)(t - 1) * Mm; ldf = n * Mm; }
    else { frame = mel + (size_t)(t - 1) * rM + (rM - Mm); ldf = ldY; }
    // DecoderPrenetWrapper (rnn_wrappers.py:249,367-378): prenet(concat(frame, previous context))
    const float* cur = nullptr; int curd = 0;
    // layer 1 of steps t >= 1 was already produced by step t-1's frame-projection launch (prenet1_next), unless frames are teacher-forced
    const bool fuse_p1 = m->fuse_prenet1 && !teacher && np > 1;
    for (int i = 0; i < hp.dec_prenet_n; ++i) {
      const int ldo = (i == np - 1) ? ldz : hp.dec_prenet[i];
      if (!(i == 0 && fuse_p1 && t > 0)) {
        SkJob j = (i == 0) ? sk_linear(m, m->dec_prenet[0], frame, ldf, Mm, w.ctx, ldc, ACT_RELU, w.pz[0], ldo)
                           : sk_linear(m, m->dec_prenet[i], cur, curd, curd, nullptr, 0, ACT_RELU, w.pz[i], ldo);
        TRY(run_skinny(st, B, &j, 1));
      }
      cur = w.pz[i]; curd = hp.dec_prenet[i];
    }
    // attention GRUCell (tacotron.py:127-130); 'simple': input = concat(prenet_out, speaker_embed) (rnn_wrappers.py:372-376)
    TRY(run_gru_cell(m, st, m->att_gru, B, cur, ldz, w.h_att, w.rh, w.u, w.xc, nullptr, ldc));
    // query + score + normaliser + context (rnn_wrappers.py:304-341)
    // the query mat-vec runs inside the attention kernel (one launch less) when its partials fit the kernel's LDS
    const bool fuse_q = (A % 4 == 0) && A <= ATT_MAXT && A / 4 <= 64 * ATT_NW;
    if (!fuse_q) { SkJob j = sk_linear(m, m->query, w.h_att, ldc, As, nullptr, 0, ACT_NONE, w.q, A);
      TRY(run_skinny(st, B, &j, 1)); }
    { AttnArgs a; memset(&a, 0, sizeof a);
      a.q = fuse_q ? nullptr : w.q; a.hq = w.h_att; a.ldhq = ldc; a.wq = AP(m, m->raw_wq); a.As = As; a.keys = w.keys; a.values = enc_out; a.v = AP(m, m->att_v); a.battn = AP(m, m->att_b);
      a.score_bias = AP(m, m->att_sb); a.manual = manual; a.align = w.align; a.hist = align_out; a.ctx = w.ctx; a.ldctx = ldc;
      a.T_in = T_in; a.A = A; a.D = D; a.type = hp.attention_type; a.step = t; a.n_steps = n;
      // few rows x many encoder positions (C5): two launches that spread every row over att_split workgroups
      const int att_split = (m->att_split >= 0) ? m->att_split : ((B <= 16 && T_in >= 256 && A % 4 == 0 && A <= 1024 && A / 4 <= 64 * ATS_NW) ? 4 : 0);
      if (att_split > 1 && fuse_q) {
        if (!manual) hipLaunchKernelGGL(k_att_scores, dim3(B, att_split), dim3(64 * ATS_NW), 0, st, a, w.escr, att_split);
        hipLaunchKernelGGL(k_att_context, dim3(B, att_split), dim3(64 * ATS_NW), 0, st, a, (const float*)w.escr, att_split);
      } else {
        hipLaunchKernelGGL(k_attention, dim3(B), dim3(64 * ATT_NW), 0, st, a);
      }
      HIPCHK(hipGetLastError()); }
    // ConcatOutputAndAttentionWrapper + OutputProjectionWrapper (rnn_wrappers.py:405-415; tacotron.py:166-170)
    // 'simple': + speaker_embed (rnn_wrappers.py:408-413).  The projection is linear, so by default it is folded into the x rows
    // of the first decoder GRU (gru1_fold): its gates launch reads [h_att | ctx | spk] directly and also emits o0 for the residual.
    const bool fold = m->fuse_concat && L > 0;
    if (!fold) { SkJob j = sk_linear(m, m->concat_proj, w.h_att, ldc, As, w.ctx, ldc, ACT_NONE, w.o[0], Hd);
      TRY(run_skinny(st, B, &j, 1)); }
    // residual GRU stack (tacotron.py:171-172)
    for (int i = 0; i < L; ++i) {
      if (i == 0 && fold) TRY(run_gru_cell(m, st, m->gru1_fold, B, w.h_att, ldc, w.hd[0], w.rh, w.u, w.xc, w.o[1], 0, 2 * Hd));
      else TRY(run_gru_cell(m, st, m->dec_gru[i], B, w.o[i], Hd, w.hd[i], w.rh, w.u, w.xc, w.o[i + 1]));
    }
    // frame projection to r frames (tacotron.py:178-179), written straight into the mel buffer
    { SkJob j[2];
      j[0] = sk_linear(m, m->frame_proj, w.o[L], Hd, Hd, nullptr, 0, ACT_NONE, mel + (size_t)t * rM, ldY);
      j[0].o2 = reinterpret_cast<float*>(w.nz + (size_t)t * B);
      int nj = 1;
      if (fuse_p1 && t + 1 < n) { j[1] = sk_linear(m, m->prenet1_next, w.o[L], Hd, Hd, w.ctx, ldc, ACT_RELU, w.pz[0], hp.dec_prenet[0]); nj = 2; }
      TRY(run_skinny(st, B, j, nj)); }
    if (after_step) TRY((*after_step)(t));
    if (dbg) {
      float* d = dbg + (size_t)t * B * dbgw;
      hipLaunchKernelGGL(k_copy2d, dim3(cdiv(B * As, 256)), dim3(256), 0, st, w.h_att, ldc, d, dbgw, B, As);
      hipLaunchKernelGGL(k_copy2d, dim3(cdiv(B * D, 256)), dim3(256), 0, st, w.ctx, ldc, d + As, dbgw, B, D);
      for (int i = 0; i < L; ++i)
        hipLaunchKernelGGL(k_copy2d, dim3(cdiv(B * Hd, 256)), dim3(256), 0, st, w.hd[i], Hd, d + As + D + i * Hd, dbgw, B, Hd);
      HIPCHK(hipGetLastError());
    }
  }
  if (stop_step) {
    hipLaunchKernelGGL(k_stop_step, dim3(1), dim3(1024), 0, st, w.nz, B, n, stop_step);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// ---- post-net + linear head (tacotron.py:219-235) ----
struct PostWs { CbhgWs cb; float* post_out; float* spk_emb; float* rowvec; };
static void carve_post(Carver& cv, const taco_model* m, int B, int T, PostWs& w) {
  carve_cbhg(cv, m->post, B, T, w.cb);
  w.post_out = cv.f((size_t)B * T * 2 * m->hp.post_rnn_size);
  w.spk_emb = cv.f((size_t)B * std::max(simple_S(m), 1));
  w.rowvec = cv.f((size_t)B * m->hp.num_freq);
}
static int postnet_tail(const taco_model* m, hipStream_t st, const int* speaker_id, int B, int T, float* linear,
                        float* post_out_user, const PostWs& w) {
  float* po = post_out_user ? post_out_user : w.post_out;
  TRY(bigru_scan(m, st, m->post, B, T, nullptr, nullptr, po, w.cb));
  GemmCall g; g.x = po; g.ldx = 2 * m->hp.post_rnn_size; g.M = B * T; g.out = linear; g.ldo = m->hp.num_freq;
  if (is_simple(m)) {
    // linear(concat(tiled speaker_embed, post)) (tacotron.py:226-235) = post . W[S:] + (speaker_embed . W[:S]) per batch row
    if (!speaker_id) return fail(TACO_ERR_ARG, "speaker_id required for a multi-speaker model");
    const int S = simple_S(m), F = m->hp.num_freq;
    hipLaunchKernelGGL(k_gather_rows, dim3(cdiv(B * S, 256)), dim3(256), 0, st, AP(m, m->spk_emb), speaker_id, B, S, w.spk_emb);
    HIPCHK(hipGetLastError());
    SkJob j = sk_linear(m, m->lin_spk, w.spk_emb, S, S, nullptr, 0, ACT_NONE, w.rowvec, F);
    TRY(run_skinny(st, B, &j, 1));
    g.T = T; g.rowvec = w.rowvec; g.ldrv = F;
  }
  return run_gemm(m, st, &m->linear, 1, false, g);
}
static int postnet_forward(const taco_model* m, hipStream_t st, const float* mel, const int* speaker_id, int B, int T,
                           float* linear, float* post_out_user, const PostWs& w) {
  FfProg pg;
  TRY(cbhg_ff_advance(m, st, m->post, mel, B, T, nullptr, nullptr, w.cb, pg, T, nullptr));
  return postnet_tail(m, st, speaker_id, B, T, linear, post_out_user, w);
}

struct FullWs { EncWs enc; DecWs dec; PostWs post; float* enc_out; };
static void carve_full(Carver& cv, const taco_model* m, int B, int T_in, int n, FullWs& w) {
  carve_enc(cv, m, B, T_in, w.enc);
  w.enc_out = cv.f((size_t)B * T_in * 2 * m->hp.enc_rnn_size);
  carve_dec(cv, m, B, T_in, n, w.dec);
  carve_post(cv, m, B, n * m->hp.reduction_factor, w.post);
}

static int check_common(const taco_model* m, int B, int T) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  if (!m->finalized) return fail(TACO_ERR_STATE, "model not finalized");
  if (B <= 0 || T <= 0) return fail(TACO_ERR_ARG, "bad batch/time %d/%d", B, T);
  if (B > 64) return fail(TACO_ERR_UNSUPPORTED, "batch %d > 64 rows per device is not supported: shard the batch", B);
  return 0;
}

// Last node of a forward: if a persistent kernel of this (or an earlier, still unacknowledged) forward gave up -- the device error
// word is sticky and makes every later persistent launch drain at once -- the forward's own stop word becomes -(error), so the
// caller that reads the stop step learns that THIS forward's outputs are invalid without a second transfer.
static int latch_errors(const taco_model* m, hipStream_t st, int32_t* stop) {
  if (!stop) return 0;
  hipLaunchKernelGGL(k_latch_errors, dim3(1), dim3(64), 0, st, (const unsigned*)m->d_err, stop);
  HIPCHK(hipGetLastError());
  return 0;
}

// One pass of at most 64 batch rows (what the persistent kernels place on one chip: 8 groups x 8 rows).
static int forward_pass(taco_model* m, hipStream_t st, const int32_t* ids, const int32_t* lengths, const int32_t* spk,
                        int B, int T_in, int n, const float* manual, float* mel, float* linear, float* align,
                        int32_t* stop, void* ws, size_t ws_bytes) {
  TRY(check_common(m, B, T_in));
  if (n <= 0) return fail(TACO_ERR_ARG, "n_steps must be positive");
  if (!ids || !lengths || !mel || !linear || !align || !ws) return fail(TACO_ERR_ARG, "null buffer");
  if (m->hp.num_speakers > 1 && !spk) return fail(TACO_ERR_ARG, "speaker_id required for a multi-speaker model");
  Carver cv(ws, ws_bytes);
  FullWs w;
  carve_full(cv, m, B, T_in, n, w);
  if (!cv.ok()) return fail(TACO_ERR_STATE, "workspace too small: need %zu bytes, have %zu", cv.off, ws_bytes);
  const int r = m->hp.reduction_factor, T_mel = n * r;
  const int CH = m->overlap > 16 ? m->overlap : 16;   // decoder steps per post-net chunk
  if (!m->overlap || (size_t)(n / CH + 4) > m->events.size()) {
    // every word a persistent kernel polls and the stop flags, cleared by ONE launch in front of the forward; the stop rule and the
    // error latch are ONE launch behind it (11 graph nodes at C2 with the clears riding in the prenet launch; 13 in round 4, 20 in round 3)
    ZeroRegions z; memset(&z, 0, sizeof z);
    z.p[0] = (uint32_t*)w.dec.xbuf; z.nw[0] = ((size_t)((char*)w.dec.dxctl - (char*)w.dec.xbuf) + 256) / 4;
    z.p[1] = (uint32_t*)w.dec.nz; z.nw[1] = (size_t)n * B;
    z.p[2] = (uint32_t*)w.post.cb.gxbuf; z.nw[2] = ((size_t)((char*)w.post.cb.gxctl - (char*)w.post.cb.gxbuf) + 256) / 4;
    z.p[3] = (uint32_t*)w.enc.cb.gxbuf; z.nw[3] = ((size_t)((char*)w.enc.cb.gxctl - (char*)w.enc.cb.gxbuf) + 256) / 4;     // (an encoder of width 256 scans on k_bigru_duo too)
    // (they ride in the encoder prenet's launch where that is the chain kernel -- the first launch of the forward, with CUs to spare)
    // The regions count as cleared (zero_async then skips its own fill) only from the point where the launch that clears them HAS been
    // enqueued: here for the fill kernel, inside run_prenet_chain for the riders -- never on the strength of a predicate evaluated twice.
    struct Guard { Guard() { g_cleared.cnt = 0; } ~Guard() { g_cleared.cnt = 0; } } guard;
    const bool ride = prenet_chain_fits(m);
    if (!ride) {
      hipLaunchKernelGGL(k_zero_fill_multi, dim3(256, 4), dim3(256), 0, st, z);
      HIPCHK(hipGetLastError());
      register_cleared(z);
    }
    TRY(encoder_forward(m, st, ids, lengths, spk, B, T_in, w.enc_out, w.enc, false, ride ? &z : nullptr));
    TRY(decoder_forward(m, st, w.enc_out, spk, B, T_in, n, manual, nullptr, mel, align, nullptr, nullptr, w.dec, true, &w.enc.spk));
    TRY(postnet_forward(m, st, mel, spk, B, T_mel, linear, nullptr, w.post));
    if (stop) {
      hipLaunchKernelGGL(k_stop_step, dim3(1), dim3(1024), 0, st, (const int*)w.dec.nz, B, n, stop, (const unsigned*)m->d_err);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }
  TRY(encoder_forward(m, st, ids, lengths, spk, B, T_in, w.enc_out, w.enc, false));
  // The decoder loop is a chain of tiny dependent launches that occupies < 1/5 of the CUs; the post-net's
  // feed-forward stages (conv bank, projections, highways, hoisted GRU projection: ~1.3 ms of fp32 MFMA work @C2)
  // need only frames that already exist plus a conv halo.  They run on a second stream, one chunk of CH steps
  // behind the decoder (fork/join by events; captured into the hipGraph as a parallel branch).
  hipStream_t s2 = m->side;
  size_t ev = 0;
  HIPCHK(hipEventRecord(m->events[ev], st));
  HIPCHK(hipStreamWaitEvent(s2, m->events[ev], 0));
  ++ev;
  FfProg pg;
  const std::function<int(int)> hook = [&](int t) -> int {
    if ((t + 1) % CH != 0 && t != n - 1) return 0;
    HIPCHK(hipEventRecord(m->events[ev], st));
    HIPCHK(hipStreamWaitEvent(s2, m->events[ev], 0));
    ++ev;
    return cbhg_ff_advance(m, s2, m->post, mel, B, T_mel, nullptr, nullptr, w.post.cb, pg, (t + 1) * r, nullptr);
  };
  TRY(decoder_forward(m, st, w.enc_out, spk, B, T_in, n, manual, nullptr, mel, align, stop, nullptr, w.dec, true, &w.enc.spk, &hook));
  HIPCHK(hipEventRecord(m->events[ev], s2));
  HIPCHK(hipStreamWaitEvent(st, m->events[ev], 0));
  TRY(postnet_tail(m, st, spk, B, T_mel, linear, nullptr, w.post));
  return latch_errors(m, st, stop);
}

// Any batch size (synthesizer.py:120-131 and eval.py:86-119 put no cap on it): more than 64 rows run as ceil(B / 64) passes of equal
// size over the SAME workspace, back to back on the caller's stream (batch rows are independent at inference: BatchNorm uses the
// moving statistics, modules.py:131).  The stop step of the batch (helpers.py:29 with dynamic_decode's all-rows-finished loop condition)
// is the maximum over the passes' stop steps -- a row's `finished` is sticky, so the loop would have ended when the LAST pass's rows
// were all done; a negative pass word (a persistent kernel gave up) wins.
struct PassPlan { int passes, rows; };
static PassPlan pass_plan(int B, int cap = 64) { PassPlan p; p.passes = (B + cap - 1) / cap; p.rows = p.passes ? (B + p.passes - 1) / p.passes : 0; return p; }
// Rows one pass of the decoder loop may hold: 64 (8 groups x 8 rows) -- but an inference model with attention_size 512 (hparams.py:71-82, the first
// "Deep Voice 2" block) has no persistent instantiation at 8 rows per group (128 score channels per member do not fit the query registers), so its
// passes hold at most 32 rows: two passes on the persistent decoder instead of one on the launch-per-stage loop, which is four times slower per step.
static int pass_cap(const taco_model* m) {
  if (m && !m->tp && m->hp.attention_size == 512 && m->dx_mode && m->dx_pack && m->cu_count >= DX_NGROUP * DX_GROUP) return 32;
  return 64;
}
__global__ void k_stop_combine(const int* pstop, int np, int* stop) {
  if (threadIdx.x == 0) {
    int worst = 0, err = 0;
    for (int i = 0; i < np; ++i) { const int v = pstop[i]; if (v < 0 && !err) err = v; worst = max(worst, v); }
    *stop = err ? err : worst;
  }
}
static size_t forward_workspace(const taco_model* m, int B, int T_in, int n, FullWs* w_out, int32_t** pstop, void* ws, size_t ws_bytes, bool* ok) {
  const PassPlan pp = pass_plan(B, pass_cap(m));
  Carver cv(ws, ws_bytes);
  FullWs w;
  carve_full(cv, m, pp.rows, T_in, n, w);
  const size_t pass_bytes = cv.off;
  int32_t* ps = pp.passes > 1 ? (int32_t*)cv.i((size_t)pp.passes) : nullptr;
  if (w_out) *w_out = w;
  if (pstop) *pstop = ps;
  if (ok) *ok = cv.ok();
  (void)pass_bytes;
  return cv.off;
}
static int forward_enqueue(taco_model* m, hipStream_t st, const int32_t* ids, const int32_t* lengths, const int32_t* spk,
                           int B, int T_in, int n, const float* manual, float* mel, float* linear, float* align,
                           int32_t* stop, void* ws, size_t ws_bytes) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  const int cap = pass_cap(m);
  if (B <= cap) return forward_pass(m, st, ids, lengths, spk, B, T_in, n, manual, mel, linear, align, stop, ws, ws_bytes);
  if (T_in <= 0 || n <= 0) return fail(TACO_ERR_ARG, "bad time %d / steps %d", T_in, n);
  if (!ids || !lengths || !mel || !linear || !align || !ws) return fail(TACO_ERR_ARG, "null buffer");
  const PassPlan pp = pass_plan(B, cap);
  int32_t* pstop = nullptr; bool ok = false;
  const size_t need = forward_workspace(m, B, T_in, n, nullptr, &pstop, ws, ws_bytes, &ok);
  if (!ok) return fail(TACO_ERR_STATE, "workspace too small: need %zu bytes, have %zu", need, ws_bytes);
  const size_t rM = (size_t)m->hp.reduction_factor * m->hp.num_mels, T_mel = (size_t)n * m->hp.reduction_factor;
  for (int p = 0; p < pp.passes; ++p) {
    const int b0 = p * pp.rows, rows = std::min(pp.rows, B - b0);
    if (rows <= 0) break;
    TRY(forward_pass(m, st, ids + (size_t)b0 * T_in, lengths + b0, spk ? spk + b0 : nullptr, rows, T_in, n,
                     manual ? manual + (size_t)b0 * n * T_in : nullptr, mel + (size_t)b0 * n * rM, linear + (size_t)b0 * T_mel * m->hp.num_freq,
                     align + (size_t)b0 * T_in * n, stop ? pstop + p : nullptr, ws, ws_bytes));
  }
  if (stop) {
    hipLaunchKernelGGL(k_stop_combine, dim3(1), dim3(64), 0, st, (const int*)pstop, pp.passes, stop);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
#include "taco_train.h"
#include "taco_audio.h"

extern "C" {

int taco_abi_version(void) { return TACO_ABI_VERSION; }
const char* taco_last_error(void) { return g_err.c_str(); }

int taco_model_create(const taco_hparams* hp, int device, taco_model** out) {
  if (!hp || !out) return fail(TACO_ERR_ARG, "null argument");
  if (hp->model_type < 0 || hp->model_type > 2) return fail(TACO_ERR_UNSUPPORTED, " [!] Unkown multi-speaker model type: %d", hp->model_type);
  if (hp->attention_type < 0 || hp->attention_type > 2) return fail(TACO_ERR_UNSUPPORTED, " [!] Unkown attention type: %d", hp->attention_type);
  if (hp->num_speakers > 1 && hp->model_type == 1 && hp->speaker_embedding_size == 1)
    return fail(TACO_ERR_UNSUPPORTED, "model_type 'simple' with speaker_embedding_size 1 leaves speaker_embed undefined in the reference (tacotron.py:44-49,82-86)");
  if (hp->num_speakers > 1 && hp->model_type == 1 && hp->speaker_embedding_size % 4)
    return fail(TACO_ERR_UNSUPPORTED, "model_type 'simple' needs speaker_embedding_size to be a multiple of 4");
  if (hp->num_speakers > 1 && hp->model_type == 0)
    return fail(TACO_ERR_UNSUPPORTED, " [!] Unkown multi-speaker model type: single (num_speakers > 1 needs simple or deepvoice)");
  if (hp->enc_prenet_n < 1 || hp->enc_prenet_n > 4 || hp->dec_prenet_n < 1 || hp->dec_prenet_n > 4 || hp->enc_proj_n < 1 ||
      hp->enc_proj_n > 4 || hp->post_proj_n < 1 || hp->post_proj_n > 4 || hp->dec_layer_num < 1 || hp->dec_layer_num > 4)
    return fail(TACO_ERR_ARG, "layer-count hparams out of range");
  if (hp->enc_proj[hp->enc_proj_n - 1] != hp->enc_prenet[hp->enc_prenet_n - 1])
    return fail(TACO_ERR_SHAPE, "encoder CBHG residual needs enc_proj_sizes[-1] == enc_prenet_sizes[-1]");
  if (hp->post_proj[hp->post_proj_n - 1] != hp->num_mels)
    return fail(TACO_ERR_SHAPE, "post CBHG residual needs post_proj_sizes[-1] == num_mels");
  if (hp->attention_state_size % 4 || hp->dec_rnn_size % 4 || hp->enc_rnn_size % 4 || hp->post_rnn_size % 4 || hp->num_mels % 4)
    return fail(TACO_ERR_UNSUPPORTED, "rnn sizes and num_mels must be multiples of 4");
  taco_model* m = new taco_model();
  m->hp = *hp;
  m->device = device;
  build_spec(m);
  cbhg_dims(m->enc, hp->enc_prenet[hp->enc_prenet_n - 1], hp->enc_bank_size, hp->enc_bank_channels, hp->enc_maxpool,
            hp->enc_highway_depth, hp->enc_rnn_size, hp->enc_proj, hp->enc_proj_n, hp->enc_proj_width);
  cbhg_dims(m->post, hp->num_mels, hp->post_bank_size, hp->post_bank_channels, hp->post_maxpool, hp->post_highway_depth,
            hp->post_rnn_size, hp->post_proj, hp->post_proj_n, hp->post_proj_width);
  *out = m;
  return 0;
}

int taco_model_num_weights(const taco_model* m) { return m ? (int)m->spec.size() : 0; }

int taco_model_weight_name(const taco_model* m, int i, char* buf, int buflen, int64_t* shape4, int* ndim) {
  if (!m || i < 0 || i >= (int)m->spec.size() || !buf) return fail(TACO_ERR_ARG, "bad index");
  snprintf(buf, buflen, "%s", m->spec[i].first.c_str());
  if (ndim) *ndim = (int)m->spec[i].second.size();
  if (shape4) for (size_t d = 0; d < m->spec[i].second.size() && d < 4; ++d) shape4[d] = m->spec[i].second[d];
  return 0;
}

int taco_model_set_weight(taco_model* m, const char* name, const float* host, const int64_t* shape, int ndim) {
  if (!m || !name || !host || ndim < 0 || (ndim > 0 && !shape)) return fail(TACO_ERR_ARG, "null argument");
  if (m->finalized) return fail(TACO_ERR_STATE, "model already finalized");
  for (auto& s : m->spec) {
    if (s.first != name) continue;
    if ((int)s.second.size() != ndim) return fail(TACO_ERR_SHAPE, "%s: rank %d, expected %zu", name, ndim, s.second.size());
    size_t n = 1;
    for (int d = 0; d < ndim; ++d) {
      if (shape[d] != s.second[d]) return fail(TACO_ERR_SHAPE, "%s: dim %d is %lld, expected %lld", name, d, (long long)shape[d], (long long)s.second[d]);
      n *= (size_t)shape[d];
    }
    HostTensor& t = m->raw[name];
    t.shape.assign(shape, shape + ndim);
    t.data.assign(host, host + n);
    t.set = true;
    return 0;
  }
  return fail(TACO_ERR_ARG, "unknown weight name '%s'", name);
}

int taco_model_finalize(taco_model* m) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  if (m->finalized) return 0;
  for (auto& s : m->spec)
    if (!m->raw.count(s.first) || !m->raw[s.first].set) return fail(TACO_ERR_STATE, "weight '%s' was never set", s.first.c_str());
  const taco_hparams& hp = m->hp;
  HIPCHK(hipSetDevice(m->device));
  HIPCHK(hipDeviceGetAttribute(&m->cu_count, hipDeviceAttributeMultiprocessorCount, m->device));
  m->harena.clear(); m->hvars.clear();
  m->emb = arena_put(m, T_(m, "embedding").data.data(), T_(m, "embedding").data.size());
  for (int i = 0; i < hp.enc_prenet_n; ++i) {
    const std::string n = "prenet/dense_" + std::to_string(i + 1);
    m->enc_prenet.push_back(make_conv(m, n, false, true, true));
  }
  make_cbhg(m, m->enc, "encoder_cbhg", hp.enc_prenet[hp.enc_prenet_n - 1], hp.enc_bank_size, hp.enc_bank_channels,
            hp.enc_maxpool, hp.enc_highway_depth, hp.enc_rnn_size, hp.enc_proj, hp.enc_proj_n, hp.enc_proj_width, true);
  // fp32-grade: the keys feed the alignment argmax, the one output held to "bit-identical".  Rounds 1-3: the exact-fp32 MFMA (k_gemm, +35 us per
  // C2 forward over the three-product split); round 4: the SIX-product split (operands split three ways, 2^-24 per product) on the bf16 pipe
  m->memory_layer = make_conv(m, "attention/memory_layer", false, false, true, true);
  make_cbhg(m, m->post, "post_cbhg", hp.num_mels, hp.post_bank_size, hp.post_bank_channels, hp.post_maxpool,
            hp.post_highway_depth, hp.post_rnn_size, hp.post_proj, hp.post_proj_n, hp.post_proj_width, true);
  if (hp.num_speakers > 1 && hp.model_type == 1) {
    // linear head input = concat(speaker_embed, post_outputs) (tacotron.py:226-235): rows [0,S) of the kernel
    // multiply the per-utterance embedding -> a per-batch-row vector; rows [S,..) stay a GEMM over the frames
    const int S = hp.speaker_embedding_size, F = hp.num_freq;
    HostTensor& k = m->raw["linear/kernel"];
    m->lin_spk = pack_w16(m, k.data.data(), F, 0, S, 0, F, nullptr);
    { HostTensor head; head.shape = {(int64_t)S, (int64_t)F}; head.data.assign(k.data.begin(), k.data.begin() + (size_t)S * F); head.set = true;
      m->raw["linear/kernel_spk"] = head; }     // kept for the training packs (transposed speaker rows)
    HostTensor rest; rest.shape = {k.shape[0] - S, (int64_t)F};
    rest.data.assign(k.data.begin() + (size_t)S * F, k.data.end());
    rest.set = true;
    m->raw["linear/kernel"] = rest;
    m->spk_emb = arena_put(m, T_(m, "speaker_embedding").data.data(), T_(m, "speaker_embedding").data.size());
  }
  m->linear = make_conv(m, "linear", false, true, true);
  // decoder (skinny packs)
  const int D = 2 * hp.enc_rnn_size, As = hp.attention_state_size, Hd = hp.dec_rnn_size, A = hp.attention_size;
  int d = hp.num_mels + D;
  for (int i = 0; i < hp.dec_prenet_n; ++i) {
    const std::string n = "decoder/prenet/dense_" + std::to_string(i + 1);
    m->dec_prenet.push_back(pack_w16(m, T_(m, n + "/kernel").data.data(), hp.dec_prenet[i], 0, d, 0, hp.dec_prenet[i], T_(m, n + "/bias").data.data()));
    m->skinny[n] = m->dec_prenet.back();
    d = hp.dec_prenet[i];
  }
  m->att_gru = make_grudec(m, "decoder/attention_gru", d + simple_S(m), As);   // simple: input = concat(prenet_out, speaker_embed)
  m->grus["decoder/attention_gru"] = m->att_gru;
  m->query = pack_w16(m, T_(m, "attention/query_layer/kernel").data.data(), A, 0, As, 0, A, nullptr);
  m->skinny["attention/query_layer"] = m->query;
  m->raw_wq = arena_put(m, T_(m, "attention/query_layer/kernel").data.data(), (size_t)As * A);
  m->concat_proj = pack_w16(m, T_(m, "decoder/concat_projection/kernel").data.data(), Hd, 0, As + D + simple_S(m), 0, Hd, T_(m, "decoder/concat_projection/bias").data.data());
  m->skinny["decoder/concat_projection"] = m->concat_proj;
  for (int i = 0; i < hp.dec_layer_num; ++i) {
    const std::string n = "decoder/gru_" + std::to_string(i + 1);
    m->dec_gru.push_back(make_grudec(m, n, Hd, Hd));
    m->grus[n] = m->dec_gru.back();
  }
  const int rM = hp.num_mels * hp.reduction_factor;
  m->frame_proj = pack_w16(m, T_(m, "decoder/frame_projection/kernel").data.data(), rM, 0, Hd, 0, rM, T_(m, "decoder/frame_projection/bias").data.data());
  m->skinny["decoder/frame_projection"] = m->frame_proj;
  std::vector<float> dx_Wc, dx_bc;   // kept for the persistent decoder's pack (dx_build_pack)
  {  // prenet layer 1 of the NEXT step, fed with this step's outputs (helpers.py:31 feeds back the last of the r frames):
     //   relu([frame, ctx] . W1 + b1), frame = o . Wf[:, rM-M:] + bf[rM-M:]   =>   relu([o, ctx] . [Wf_last . W1a ; W1b] + (b1 + bf_last . W1a))
     // computed in the same launch as the frame projection: one dependent launch less per decoder step.
    const int Mm = hp.num_mels, P0 = hp.dec_prenet[0];
    const auto& Wf = T_(m, "decoder/frame_projection/kernel").data; const auto& bf = T_(m, "decoder/frame_projection/bias").data;
    const auto& W1 = T_(m, "decoder/prenet/dense_1/kernel").data; const auto& b1 = T_(m, "decoder/prenet/dense_1/bias").data;
    std::vector<float>& Wc = dx_Wc; std::vector<float>& bc = dx_bc;
    Wc.assign((size_t)(Hd + D) * P0, 0.f); bc.assign(P0, 0.f);
    for (int k = 0; k < Hd; ++k)
      for (int q = 0; q < P0; ++q) {
        double acc = 0;
        for (int j = 0; j < Mm; ++j) acc += (double)Wf[(size_t)k * rM + (rM - Mm) + j] * (double)W1[(size_t)j * P0 + q];
        Wc[(size_t)k * P0 + q] = (float)acc;
      }
    for (int k = 0; k < D; ++k)
      for (int q = 0; q < P0; ++q) Wc[(size_t)(Hd + k) * P0 + q] = W1[(size_t)(Mm + k) * P0 + q];
    for (int q = 0; q < P0; ++q) {
      double acc = b1[q];
      for (int j = 0; j < Mm; ++j) acc += (double)bf[(rM - Mm) + j] * (double)W1[(size_t)j * P0 + q];
      bc[q] = (float)acc;
    }
    m->prenet1_next = pack_w16(m, Wc.data(), P0, 0, Hd + D, 0, P0, bc.data());
  }
  if (!m->tp && hp.dec_layer_num > 0) {
    // Decoder GRU 1 with the concat projection folded in.  o0 = z . Wc + bc is linear in z = [h_att | ctx (| spk)], so
    //   gates:      [o0, h] . Wg + bg = z . (Wc . Wg_x) + h . Wg_h + (bg + bc . Wg_x)
    //   candidate x: o0 . Wk_x       = z . (Wc . Wk_x) + bc . Wk_x
    // and o0 itself (the ResidualWrapper input, tacotron.py:172) comes out of the same launch as Hd extra columns: one dependent
    // launch less per decoder step.  Products are formed in double and rounded once.  (Not built for the training shadow model.)
    const int Z = As + D + simple_S(m), N4 = 4 * Hd;
    const auto& Wc = T_(m, "decoder/concat_projection/kernel").data; const auto& bcv = T_(m, "decoder/concat_projection/bias").data;
    const auto& gk = T_(m, "decoder/gru_1/gates/kernel").data; const auto& gb = T_(m, "decoder/gru_1/gates/bias").data;
    const auto& ck = T_(m, "decoder/gru_1/candidate/kernel").data; const auto& cb = T_(m, "decoder/gru_1/candidate/bias").data;
    std::vector<float> W((size_t)(Z + Hd) * N4, 0.f), bb(N4, 0.f);
    std::vector<double> row(3 * Hd);
    for (int z = 0; z <= Z; ++z) {      // row Z of this loop = the bias row (z . Wc replaced by bc)
      std::fill(row.begin(), row.end(), 0.0);
      for (int k = 0; k < Hd; ++k) {
        const double c = (z < Z) ? (double)Wc[(size_t)z * Hd + k] : (double)bcv[k];
        for (int j = 0; j < 2 * Hd; ++j) row[j] += c * (double)gk[(size_t)k * 2 * Hd + j];
        for (int j = 0; j < Hd; ++j) row[2 * Hd + j] += c * (double)ck[(size_t)k * Hd + j];
      }
      if (z < Z) {
        for (int j = 0; j < 3 * Hd; ++j) W[(size_t)z * N4 + j] = (float)row[j];
        for (int j = 0; j < Hd; ++j) W[(size_t)z * N4 + 3 * Hd + j] = Wc[(size_t)z * Hd + j];
      } else {
        for (int j = 0; j < 2 * Hd; ++j) bb[j] = (float)(row[j] + (double)gb[j]);
        for (int j = 0; j < Hd; ++j) { bb[2 * Hd + j] = (float)row[2 * Hd + j]; bb[3 * Hd + j] = bcv[j]; }
      }
    }
    for (int k = 0; k < Hd; ++k)
      for (int j = 0; j < 2 * Hd; ++j) W[(size_t)(Z + k) * N4 + j] = gk[(size_t)(Hd + k) * 2 * Hd + j];
    m->gru1_fold.I = Z; m->gru1_fold.H = Hd;
    m->gru1_fold.gx = pack_w16(m, W.data(), N4, 0, Z + Hd, 0, N4, bb.data());
    m->gru1_fold.ch = pack_w16(m, ck.data(), Hd, Hd, Hd, 0, Hd, cb.data());
    m->fuse_concat = 1;
    if (hp.dec_layer_num == 2) TRY(dx_build_pack(m, dx_Wc, dx_bc, W, bb));
  }
  if (m->tp && hp.dec_layer_num == 2 && dx_widths_ok(m)) {
    // Training shadow model: every "weight" here is 1 + the flat index of a parameter, and the arena doubles as the index map of
    // taco_train_refresh.  The persistent decoder's pack is built in TEACHER form: the next step's prenet layer 1 reads the
    // teacher's frame, so its registers hold the raw kernel rows (zero beyond num_mels) instead of the frame-projection composite;
    // the one composite that remains -- the concat projection folded into GRU 1 -- is computed on the device after every
    // optimizer step (k_dx_fold) into a buffer that the index map addresses as NP + 1 + i.
    const int Mm = hp.num_mels, P0 = hp.dec_prenet[0], H = Hd, Z = As + D + simple_S(m), N4 = 4 * H;
    const auto& W1 = T_(m, "decoder/prenet/dense_1/kernel").data;
    std::vector<float> Wc_t((size_t)(Hd + D) * P0, 0.f);
    for (int k = 0; k < Mm; ++k) for (int q = 0; q < P0; ++q) Wc_t[(size_t)k * P0 + q] = W1[(size_t)k * P0 + q];                 // frame rows (k < num_mels)
    for (int k = 0; k < D; ++k) for (int q = 0; q < P0; ++q) Wc_t[(size_t)(Hd + k) * P0 + q] = W1[(size_t)(Mm + k) * P0 + q];    // context rows
    const auto& Wcc = T_(m, "decoder/concat_projection/kernel").data; const auto& bcv = T_(m, "decoder/concat_projection/bias").data;
    const auto& gk = T_(m, "decoder/gru_1/gates/kernel").data;
    size_t NP = 0;
    for (auto& sp : m->spec) { size_t c = 1; for (int64_t d : sp.second) c *= (size_t)d; NP += c; }
    std::vector<float> Wf_t((size_t)(Z + Hd) * N4, 0.f), bf_t(N4, 0.f);
    for (int z = 0; z <= Z; ++z)
      for (int j = 0; j < 3 * H; ++j) {
        const float idx = (float)(NP + 1 + (size_t)z * 3 * H + j);
        if (z < Z) Wf_t[(size_t)z * N4 + j] = idx; else bf_t[j] = idx;
      }
    for (int z = 0; z < Z; ++z) for (int j = 0; j < H; ++j) Wf_t[(size_t)z * N4 + 3 * H + j] = Wcc[(size_t)z * Hd + j];
    for (int j = 0; j < H; ++j) bf_t[3 * H + j] = bcv[j];
    for (int k = 0; k < Hd; ++k) for (int j = 0; j < 2 * H; ++j) Wf_t[(size_t)(Z + k) * N4 + j] = gk[(size_t)(Hd + k) * 2 * H + j];
    m->dx_fold_n = (size_t)(Z + 1) * 3 * H;
    if (NP + m->dx_fold_n >= (1u << 24)) return fail(TACO_ERR_UNSUPPORTED, "parameter + fold indices exceed 2^24");
    TRY(dx_build_pack(m, Wc_t, T_(m, "decoder/prenet/dense_1/bias").data, Wf_t, bf_t));
    TRY(dbx_build_pack(m));
  }
  {  // attention vectors; bah_norm: v_hat = g * v / |v| (A.9)
    std::vector<float> v = T_(m, "attention/attention_v").data;
    if (hp.attention_type == 1) {
      double nrm = 0; for (float x : v) nrm += (double)x * x;
      const double sc = (double)T_(m, "attention/attention_g").data[0] / std::sqrt(nrm);
      for (float& x : v) x = (float)(x * sc);
      m->att_b = arena_put(m, T_(m, "attention/attention_b").data.data(), A);
    }
    m->att_v = arena_put(m, v.data(), A);
    if (hp.attention_type == 2) m->att_sb = arena_put(m, T_(m, "attention/attention_score_bias").data.data(), 1);
  }
  if (is_deepvoice(m)) {
    std::vector<std::string> names = {kSpkNames[0], kSpkNames[1], kSpkNames[2]};
    for (int i = 0; i < hp.dec_layer_num; ++i) names.push_back("decoder_rnn_init_" + std::to_string(i + 1));
    if (hp.speaker_embedding_size == 1) {
      for (auto& n : names) m->spk_table.push_back(arena_put(m, T_(m, "spk/" + n + "/table").data.data(), T_(m, "spk/" + n + "/table").data.size()));
    } else {
      m->spk_emb = arena_put(m, T_(m, "speaker_embedding").data.data(), T_(m, "speaker_embedding").data.size());
      for (auto& n : names) {
        const HostTensor& k = T_(m, "spk/" + n + "/kernel");
        m->spk_dense.push_back(pack_w16(m, k.data.data(), (int)k.shape[1], 0, (int)k.shape[0], 0, (int)k.shape[1], T_(m, "spk/" + n + "/bias").data.data()));
      }
    }
  }
  // GemmVar table (bank variants must be contiguous)
  for (int i = 0; i < hp.enc_prenet_n; ++i) add_var(m, m->enc_prenet[i], 0);
  for (Cbhg* c : {&m->enc, &m->post}) {
    for (size_t i = 0; i < c->bank.size(); ++i) add_var(m, c->bank[i], (c->bank[i].kw - 1) * c->C);
    for (auto& L : c->proj) add_var(m, L, 0);
    if (c->has_dense) add_var(m, c->dense, 0);
    for (auto& L : c->hw) add_var(m, L, 0);
    add_var(m, c->xproj, 0);
  }
  add_var(m, m->memory_layer, 0);
  add_var(m, m->linear, 0);
  // stand-alone copies for the op-level entry points (coff 0)
  for (auto& kv : m->convs) add_var(m, kv.second, 0);
  for (int i = 0; i < hp.enc_prenet_n; ++i) m->convs["prenet/dense_" + std::to_string(i + 1)] = m->enc_prenet[i];
  m->convs["attention/memory_layer"] = m->memory_layer;
  m->convs["linear"] = m->linear;
  if (m->tp) TRY(build_train_packs(m));
  // upload
  HIPCHK(hipMalloc((void**)&m->darena, m->harena.size() * sizeof(float)));
  HIPCHK(hipMemcpy(m->darena, m->harena.data(), m->harena.size() * sizeof(float), hipMemcpyHostToDevice));
  for (auto& v : m->hvars) {
    v.wp = AP(m, (size_t)v.wp); v.wp2 = AP(m, (size_t)v.wp2); v.bias = AP(m, (size_t)v.bias); v.bias2 = AP(m, (size_t)v.bias2);
    v.bn_scale = AP(m, (size_t)v.bn_scale); v.bn_shift = AP(m, (size_t)v.bn_shift);
    v.bh = (const unsigned short*)AP(m, (size_t)v.bh); v.bl = (const unsigned short*)AP(m, (size_t)v.bl);
    v.bh2 = (const unsigned short*)AP(m, (size_t)v.bh2); v.bl2 = (const unsigned short*)AP(m, (size_t)v.bl2);
    v.bl3 = (const unsigned short*)AP(m, (size_t)v.bl3); v.bl3_2 = (const unsigned short*)AP(m, (size_t)v.bl3_2);
  }
  // persistent kernels carve up to the full 160 KiB of LDS
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cbhg_front<2, 80, 80, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cbhg_front<1, 144, 128, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_res<256, 64, 24, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_res<128, 32, 0, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_res<256, 64, 24, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_res<128, 32, 0, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_resu<256, 4, 16, 4, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_resw<16, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_resu<256, 4, 12, 5, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_resu<256, 2, 32, 10, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_rows<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_rows<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_rows<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_rows<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pointwise_chain<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head_sweep<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head_sweep<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pointwise_chain<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_xcd<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_xcd<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_xcd<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_xcd<8, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_xcd<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_xcd<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bigru_xcd<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<8, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define DX_ATTR_T(RG) \
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<RG, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<RG, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  DX_ATTR_T(1) DX_ATTR_T(2) DX_ATTR_T(4) DX_ATTR_T(8)      // teacher frames (with and without manual alignments)
#undef DX_ATTR_T
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<1, false, false, DX_W, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      // the stamped instantiations
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<2, false, false, DX_W, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<4, false, false, DX_W, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<8, false, false, DX_W, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define DX_ATTR(RG, AW, PD) \
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<RG, false, false, AW, PD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decoder_xcd<RG, false, true, AW, PD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  DX_ATTR(4, 128, 2) DX_ATTR(8, 128, 2) DX_ATTR(4, 256, 3) DX_ATTR(8, 256, 3) DX_ATTR(4, 512, 3) DX_ATTR(4, 128, 3) DX_ATTR(8, 128, 3) DX_ATTR(4, 512, 2)
#undef DX_ATTR
  HIPCHK(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
  m->events.resize(192);
  for (auto& e : m->events) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(hipMalloc((void**)&m->d_err, 256));
  HIPCHK(hipMemset(m->d_err, 0, 256));
  m->arena_n = m->harena.size();
  m->harena.clear(); m->harena.shrink_to_fit();
  m->raw.clear();
  m->finalized = true;
  return 0;
}

void taco_model_destroy(taco_model* m) {
  if (!m) return;
  if (m->darena) (void)hipFree(m->darena);
  if (m->d_err) (void)hipFree(m->d_err);
  if (m->d_trace) (void)hipFree(m->d_trace);
  for (auto& e : m->events) (void)hipEventDestroy(e);
  if (m->side) (void)hipStreamDestroy(m->side);
  delete m;
}

int taco_model_device_errors(taco_model* m, int* out) {
  if (!m || !m->finalized || !out) return fail(TACO_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->device));
  unsigned v = 0;
  HIPCHK(hipMemcpy(&v, m->d_err, sizeof v, hipMemcpyDeviceToHost));
  if (v) HIPCHK(hipMemset(m->d_err, 0, 256));
  *out = (int)v;
  return 0;
}

// test hook: raise the sticky device error word by hand (what a persistent kernel does when its bounded spin expires)
int taco_debug_raise_device_error(taco_model* m, int value) {
  if (!m || !m->finalized) return fail(TACO_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->device));
  const unsigned v = (unsigned)value;
  HIPCHK(hipMemcpy(m->d_err, &v, sizeof v, hipMemcpyHostToDevice));
  return 0;
}

int taco_debug_set_bf3(taco_model* m, int on, int tile_n) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  m->bf3 = (on & 1) != 0; m->bf3_tn = tile_n;
  m->chain = (on & 4) ? 0 : 1;     // on = 5: split-bf16 GEMMs with one launch per point-wise layer (A/B of taco_chain.h)
  m->front = (on & 8) ? 0 : 1;     // on = 9: conv bank and proj_1 as two k_gemm_bf3 launches (A/B of taco_front.h)
  m->front_entry = (on & 16) ? 0 : 1;   // on = 17: fused front, but k_front_combine and proj_2 as launches of their own (A/B of the chain's fused entry)
  m->head_sweep = (on & 32) ? 0 : 1;    // on = 33: the linear head on k_gemm_bf3's 64 x 256 tiles (A/B of taco_head.h)
  // on = 65: EVERY feed-forward layer on the six-product instantiation of k_gemm_bf3 (operands split three ways: fp32-grade products on the bf16
  // pipe), one launch per layer -- the fused front / chain / head kernels are three-product kernels and stay out of this mode
  if (!m->tp) m->bf3x6 = (on & 64) ? 1 : 0;
  return 0;
}

int taco_attention_trim(void* hip_stream, const float* d_alignments, const int32_t* d_seq_len, int B, int T_in, int n_steps,
                        int reduction_factor, int32_t* d_spec_end) {
  if (!d_alignments || !d_seq_len || !d_spec_end || B <= 0 || T_in <= 0 || n_steps <= 0) return fail(TACO_ERR_ARG, "bad argument");
  if ((size_t)n_steps * sizeof(int) > 64 * 1024) return fail(TACO_ERR_UNSUPPORTED, "more than 16384 decoder steps");
  hipLaunchKernelGGL(k_attention_trim, dim3(B), dim3(64), (size_t)n_steps * sizeof(int), (hipStream_t)hip_stream, d_alignments, d_seq_len, T_in,
                     n_steps, reduction_factor, d_spec_end);
  HIPCHK(hipGetLastError());
  return 0;
}

int taco_stop_steps(void* hip_stream, const float* d_mel, int B, int n_steps, int width, int rows_per_group, int32_t* d_stop) {
  if (!d_mel || !d_stop || B <= 0 || n_steps <= 0 || width <= 0 || rows_per_group <= 0 || B % rows_per_group)
    return fail(TACO_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)hip_stream;
  HIPCHK(zero_async(d_stop, (size_t)(B / rows_per_group) * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_stop_groups, dim3(B), dim3(256), 0, st, d_mel, n_steps, width, rows_per_group, d_stop);
  HIPCHK(hipGetLastError());
  return 0;
}

// on: every tile of k_pointwise_chain walks a layer's K from step 0, so a row's fp32 accumulation order does not depend on the tile -- i.e. on
// the batch position or the shard -- it lands in: outputs are bit-invariant under batch permutation at any size.  off (default): the
// workgroups of an XCD start at different steps (they do not all ask the L2 for the same weight slices at once): results
// stay reproducible run to run, and equal under permutation to fp32 rounding, bitwise only while the launch has fewer than 8 tiles.
int taco_model_set_batch_invariant(taco_model* m, int on) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  m->ff_rot = on ? 0 : 1;
  return 0;
}
int taco_debug_set_att_split(taco_model* m, int slices) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  m->att_split = slices;
  return 0;
}
int taco_debug_set_fuse_prenet(taco_model* m, int on) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  m->fuse_prenet1 = on ? 1 : 0;
  return 0;
}
#ifdef TACO_TRACE
int taco_debug_read_trace(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(taco_trace), 64 * sizeof(long long)) == hipSuccess ? 0 : -1; }
int taco_debug_read_trace_front(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(taco_trace_front), 64 * sizeof(long long)) == hipSuccess ? 0 : -1; }
int taco_debug_read_trace_head(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(taco_trace_head), 16 * sizeof(long long)) == hipSuccess ? 0 : -1; }
int taco_debug_read_trace_chain(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(taco_trace_chain), 64 * sizeof(long long)) == hipSuccess ? 0 : -1; }
#endif
int taco_debug_set_fuse_concat(taco_model* m, int on) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  if (on && !m->gru1_fold.H) return fail(TACO_ERR_STATE, "the folded GRU pack was not built for this model");
  m->fuse_concat = on ? 1 : 0;
  return 0;
}
int taco_debug_set_overlap(taco_model* m, int on) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  m->overlap = on;
  return 0;
}

int taco_debug_set_persistent(taco_model* m, int on) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  m->persist = on;
  return 0;
}

int taco_debug_set_decoder_persist(taco_model* m, int mode, int rows_per_group) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  if (mode < 0 || mode > 2) return fail(TACO_ERR_ARG, "mode must be 0 (launch per stage), 1 (persistent) or 2 (persistent, write-through exchanges)");
  m->dx_mode = mode; m->dx_rows = rows_per_group;
  return 0;
}
int taco_debug_decoder_info(taco_model* m, int* out16) {
  if (!m || !m->finalized || !out16) return fail(TACO_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->device));
  unsigned v[16];
  HIPCHK(hipMemcpy(v, m->d_err + 8, sizeof v, hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i) out16[i] = (int)v[i];
  { unsigned b = 0; HIPCHK(hipMemcpy(&b, m->d_err + 40, sizeof b, hipMemcpyDeviceToHost)); out16[9] = m->last_bptt ? (int)b : 0; }
  out16[14] = m->cu_count;
  out16[15] = m->dx_pack ? 1 : 0;
  return 0;
}
// Which engine a call of this shape would run on, and why not the persistent one when it would not (nothing is launched).
int taco_model_engine_plan(taco_model* m, int B, int T_in, int T_mel, int manual, char* out, int out_len) {
  if (!m || !m->finalized || !out || out_len < 64) return fail(TACO_ERR_ARG, "bad argument");
  std::string s;
  if (B > pass_cap(m) && !m->tp) {      // taco_forward_infer / taco_plan_create serve any batch as passes of at most 64 rows (32: attention_size 512); the plan below is a pass's
    const PassPlan pp = pass_plan(B, pass_cap(m));
    s = std::to_string(B) + " rows = " + std::to_string(pp.passes) + " passes of " + std::to_string(pp.rows) + " (the last one " + std::to_string(B - (pp.passes - 1) * pp.rows) + "); per pass: ";
    B = pp.rows;
  }
  auto why_common = [&](int rows) -> std::string {
    if (!m->dx_mode) return "switched off (taco_debug_set_decoder_persist 0)";
    if (m->cu_count < DX_NGROUP * DX_GROUP) return "the device exposes " + std::to_string(m->cu_count) + " compute units (a partition of an MI355X, or another part): the whole-chip kernels need 256";
    if (rows > 8 * DX_NGROUP) return std::to_string(rows) + " rows > 64 per launch";
    return "";
  };
  {  // decoder loop
    std::string why = why_common(B);
    if (why.empty() && !m->dx_pack) why = "widths outside the presets (256-wide decoder cells and memory, attention_size 128 / 256 / 512, prenet 256-128[-64], 2 decoder layers, r * num_mels <= 512)";
    const int RG = dx_rows_per_group(m, B);
    if (why.empty() && m->hp.attention_size == 512 && RG > 4) why = "attention_size 512 with more than 32 rows (no instantiation at 8 rows per group)";
    if (why.empty() && dx_lds_floats(RG, T_in, m->tp != nullptr, m->hp.attention_size) * sizeof(float) > 160 * 1024)
      why = "T_in = " + std::to_string(T_in) + " does not fit a member's LDS at " + std::to_string(RG) + " rows per group";
    if (why.empty()) s += "decoder loop: persistent k_decoder_xcd<" + std::to_string(RG) + (dx_reference_widths(m) ? std::string("") : ", attention " + std::to_string(m->hp.attention_size) + ", " +
                          std::to_string(dx_prenet_depth(m)) + " prenet layers") + "> (" + std::string(manual ? "manual alignments" : "computed alignments") + ", " +
                          std::string(m->dx_mode == 2 ? "write-through exchanges forced" : "XCD-local exchanges when the census finds 32 workgroups per XCD") + ")";
    else s += "decoder loop: one launch per stage -- " + why;
  }
  {  // post-net scan
    const Cbhg& c = m->post;
    std::string why = why_common(B);
    if (why.empty() && m->persist != 1 && m->persist != 8 && m->persist != 9 && m->persist != 10 && m->persist != 11) why = "taco_debug_set_persistent(" + std::to_string(m->persist) + ")";
    if (why.empty() && (c.rnn != GX_H || !c.gd_pack)) why = "post_rnn_size " + std::to_string(c.rnn) + " != 256";
    if (why.empty() && T_mel < 2) why = "fewer than 2 frames";
    int RG = 1;
    while (RG * DX_NGROUP < B) RG *= 2;
    const int upw = why.empty() ? oct_upw(m, c, B, T_mel) : 0;
    if (upw) s += "; post-net scan: persistent k_bigru_oct<" + std::to_string(upw) + "> (one row per cluster of " + std::to_string(DX_GROUP / upw) + " CUs)";
    else if (why.empty()) s += "; post-net scan: persistent " + std::string(m->persist == 8 || m->persist == 9 ? "k_bigru_xcd<" : "k_bigru_duo<") + std::to_string(RG) + ">";
    else if (c.rnn == 128 && m->persist == 1) s += "; post-net scan: k_bigru_quad (post_rnn_size 128: a row and direction per workgroup, its weights resident, no exchange between workgroups)";
    else s += "; post-net scan: resident per-row kernels -- " + why;
  }
  s += m->enc.rnn == 128 && m->persist == 1 ? "; encoder scan: k_bigru_quad (a row and direction per workgroup, K split inside a quad of lanes); feed-forward: "
                                            : "; encoder scan: k_bigru_res (rows resident per workgroup); feed-forward: ";
  s += m->bf3 ? "split-bf16 MFMA (k_gemm_bf3 / k_pointwise_chain)" : "exact-fp32 MFMA (k_gemm)";
  if (!m->tp) s += prenet_chain_fits(m) ? "; encoder prenet: one k_pointwise_chain launch (embedding rows gathered, both layers; the forward's zero fills ride in it)"
                                        : "; encoder prenet: one GEMM launch per layer";
  else {        // the training step's backward scans (taco_train.h: cbhg_backward)
    const Cbhg& c = m->post;
    s += std::string("; backward scans: post-net ") + (duo_usable(m, c, B, T_mel) && c.gb_pack ? (oct_bwd_usable(m, c, B, T_mel) ? "k_bigru_oct_bwd (one row per cluster of 8 CUs)" : "k_bigru_duo_bwd")
                                                                                                : "k_bigru_rows_bwd") +
         ", encoder " + (m->enc.rnn == 128 ? "k_bigru_resb (recurrent kernels in registers)" : "k_bigru_rows_bwd");
  }
  snprintf(out, (size_t)out_len, "%s", s.c_str());
  return 0;
}
int taco_debug_decoder_trace(taco_model* m, int enable, long long* out) {
  if (!m || !m->finalized) return fail(TACO_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->device));
  const size_t half = (size_t)DX_TRACE_STEPS * DX_TRACE_SLOTS * sizeof(long long);
  if ((enable & 1) && !m->d_trace) { HIPCHK(hipMalloc((void**)&m->d_trace, 3 * half)); HIPCHK(hipMemset(m->d_trace, 0, 3 * half)); }
  // enable bit 1 selects the scan's third of the buffer for the read-back, bit 2 the persistent BPTT's
  if (out && m->d_trace) HIPCHK(hipMemcpy(out, (const char*)m->d_trace + ((enable & 4) ? 2 * half : (enable & 2) ? half : 0), half, hipMemcpyDeviceToHost));
  m->trace_on = enable & 1;      // the buffer itself stays until taco_model_destroy: a captured plan may still carry its address
  return 0;
}

int taco_debug_set_front(taco_model* m, int delay_clocks, int prio) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  m->front_delay = delay_clocks; m->front_prio = prio;     // k_cbhg_front: start delay and s_setprio level of the second K half (tools/time_front.py)
  return 0;
}
int taco_debug_set_chip_turns(int on) { g_chip_turns = on ? 1 : 0; return 0; }
int taco_debug_set_skip_scans(taco_model* m, int on) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  m->skip_scans = on ? 1 : 0;
  return 0;
}

int taco_debug_force_gemm_config(taco_model* m, int cfg) {
  if (!m) return fail(TACO_ERR_ARG, "null model");
  m->force_cfg = cfg;
  return 0;
}

size_t taco_workspace_bytes(const taco_model* m, int B, int T_in, int n_steps) {
  if (!m || B <= 0) return 0;
  return forward_workspace(m, B, T_in, n_steps, nullptr, nullptr, nullptr, 0, nullptr);      // (more than 64 rows: the workspace of ONE pass + the passes' stop words)
}

size_t taco_stage_workspace_bytes(const taco_model* m, int B, int T) {
  if (!m) return 0;
  // large enough for any single stage at (B, T): encoder at T_in=T, decoder with T_in=T and n_steps=T, post-net at T_mel=T
  Carver a(nullptr, 0), b(nullptr, 0), c(nullptr, 0);
  const int R = B > 64 ? pass_plan(B).rows : B;      // more than 64 rows: passes over one pass's workspace (+ the passes' stop words)
  const int Rd = B > pass_cap(m) ? pass_plan(B, pass_cap(m)).rows : B;
  EncWs e; carve_enc(a, m, R, T, e);
  DecWs d; carve_dec(b, m, Rd, T, T, d);
  PostWs p; carve_post(c, m, R, T, p);
  return std::max(a.off, std::max(b.off, c.off)) + (B > pass_cap(m) ? 256 : 0);
}

int taco_forward_infer(taco_model* m, void* hip_stream, const int32_t* d_inputs, const int32_t* d_input_lengths,
                       const int32_t* d_speaker_id, int B, int T_in, int n_steps, const float* d_manual_alignments,
                       float* d_mel, float* d_linear, float* d_alignments, int32_t* d_stop_step, void* d_workspace,
                       size_t workspace_bytes) {
  if (m) HIPCHK(hipSetDevice(m->device));
  return forward_enqueue(m, (hipStream_t)hip_stream, d_inputs, d_input_lengths, d_speaker_id, B, T_in, n_steps,
                         d_manual_alignments, d_mel, d_linear, d_alignments, d_stop_step, d_workspace, workspace_bytes);
}

struct taco_plan {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  size_t nodes = 0;
  int device = 0;
  bool whole_chip = false;      // the graph contains a whole-chip persistent kernel: a replay takes the device's turn (ChipTurn)
};

int taco_plan_create(taco_model* m, const int32_t* d_inputs, const int32_t* d_input_lengths, const int32_t* d_speaker_id,
                     int B, int T_in, int n_steps, const float* d_manual_alignments, float* d_mel, float* d_linear,
                     float* d_alignments, int32_t* d_stop_step, void* d_workspace, size_t workspace_bytes, taco_plan** out) {
  if (!m || !out) return fail(TACO_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(m->device));
  hipStream_t cs;
  HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { (void)hipStreamDestroy(cs); return fail(TACO_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e)); }
  g_captured_whole_chip = false;
  int rc = forward_enqueue(m, cs, d_inputs, d_input_lengths, d_speaker_id, B, T_in, n_steps, d_manual_alignments, d_mel,
                           d_linear, d_alignments, d_stop_step, d_workspace, workspace_bytes);
  const bool whole_chip = g_captured_whole_chip;
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(cs, &g);
  (void)hipStreamDestroy(cs);
  if (rc != 0) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) return fail(TACO_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  taco_plan* p = new taco_plan();
  p->graph = g; p->device = m->device; p->whole_chip = whole_chip;
  e = hipGraphInstantiate(&p->exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); delete p; return fail(TACO_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
  (void)hipGraphGetNodes(g, nullptr, &p->nodes);
  *out = p;
  return 0;
}

int taco_plan_launch(taco_plan* p, void* hip_stream) {
  if (!p || !p->exec) return fail(TACO_ERR_ARG, "null plan");
  if (p->whole_chip) {
    ChipTurn turn(p->device, (hipStream_t)hip_stream);
    HIPCHK(hipGraphLaunch(p->exec, (hipStream_t)hip_stream));
    return 0;
  }
  HIPCHK(hipGraphLaunch(p->exec, (hipStream_t)hip_stream));
  return 0;
}
int taco_plan_num_nodes(const taco_plan* p) { return p ? (int)p->nodes : 0; }
int taco_plan_whole_chip(const taco_plan* p) { return (p && p->whole_chip) ? 1 : 0; }
void taco_plan_destroy(taco_plan* p) {
  if (!p) return;
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  delete p;
}

int taco_encoder_forward(taco_model* m, void* hip_stream, const int32_t* d_inputs, const int32_t* d_input_lengths,
                         const int32_t* d_speaker_id, int B, int T_in, float* d_encoder_out, void* d_workspace,
                         size_t workspace_bytes) {
  if (m && m->finalized && B > 64 && T_in > 0 && d_inputs && d_input_lengths && d_encoder_out) {      // passes of at most 64 rows (forward_enqueue)
    const PassPlan pp = pass_plan(B);
    for (int b0 = 0; b0 < B; b0 += pp.rows)
      TRY(taco_encoder_forward(m, hip_stream, d_inputs + (size_t)b0 * T_in, d_input_lengths + b0, d_speaker_id ? d_speaker_id + b0 : nullptr, std::min(pp.rows, B - b0),
                               T_in, d_encoder_out + (size_t)b0 * T_in * 2 * m->hp.enc_rnn_size, d_workspace, workspace_bytes));
    return 0;
  }
  TRY(check_common(m, B, T_in));
  if (!d_inputs || !d_input_lengths || !d_encoder_out || !d_workspace) return fail(TACO_ERR_ARG, "null buffer");
  if (m->hp.num_speakers > 1 && !d_speaker_id) return fail(TACO_ERR_ARG, "speaker_id required for a multi-speaker model");
  HIPCHK(hipSetDevice(m->device));
  Carver cv(d_workspace, workspace_bytes);
  EncWs w; carve_enc(cv, m, B, T_in, w);
  if (!cv.ok()) return fail(TACO_ERR_STATE, "workspace too small: need %zu bytes", cv.off);
  return encoder_forward(m, (hipStream_t)hip_stream, d_inputs, d_input_lengths, d_speaker_id, B, T_in, d_encoder_out, w, false);
}

int taco_decoder_forward(taco_model* m, void* hip_stream, const float* d_encoder_out, const int32_t* d_speaker_id, int B,
                         int T_in, int n_steps, const float* d_manual_alignments, const float* d_teacher_frames,
                         float* d_mel, float* d_alignments, int32_t* d_stop_step, float* d_dbg_states, void* d_workspace,
                         size_t workspace_bytes) {
  if (m && m->finalized && B > pass_cap(m) && !(d_dbg_states && B <= 64) && T_in > 0 && n_steps > 0 && d_encoder_out && d_mel && d_alignments && d_workspace) {      // passes of at most 64 (32) rows
    if (d_dbg_states) return fail(TACO_ERR_UNSUPPORTED, "the per-step state dump is laid out [step][row]: at most 64 rows per call");
    const PassPlan pp = pass_plan(B, pass_cap(m));
    const size_t rM = (size_t)m->hp.reduction_factor * m->hp.num_mels, D = (size_t)2 * m->hp.enc_rnn_size;
    if (workspace_bytes < 256) return fail(TACO_ERR_STATE, "workspace too small");
    int32_t* pstop = (int32_t*)((char*)d_workspace + ((workspace_bytes - (size_t)pp.passes * sizeof(int32_t)) & ~(size_t)255));      // the tail of the caller's buffer
    const size_t pass_bytes = (size_t)((char*)pstop - (char*)d_workspace);
    int p = 0;
    for (int b0 = 0; b0 < B; b0 += pp.rows, ++p)
      TRY(taco_decoder_forward(m, hip_stream, d_encoder_out + (size_t)b0 * T_in * D, d_speaker_id ? d_speaker_id + b0 : nullptr, std::min(pp.rows, B - b0), T_in, n_steps,
                               d_manual_alignments ? d_manual_alignments + (size_t)b0 * n_steps * T_in : nullptr,
                               d_teacher_frames ? d_teacher_frames + (size_t)b0 * n_steps * m->hp.num_mels : nullptr, d_mel + (size_t)b0 * n_steps * rM,
                               d_alignments + (size_t)b0 * T_in * n_steps, d_stop_step ? pstop + p : nullptr, nullptr, d_workspace, pass_bytes));
    if (d_stop_step) {
      hipLaunchKernelGGL(k_stop_combine, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, (const int*)pstop, pp.passes, d_stop_step);
      HIPCHK(hipGetLastError());
    }
    return 0;
  }
  TRY(check_common(m, B, T_in));
  if (n_steps <= 0 || !d_encoder_out || !d_mel || !d_alignments || !d_workspace) return fail(TACO_ERR_ARG, "bad argument");
  if (m->hp.num_speakers > 1 && !d_speaker_id) return fail(TACO_ERR_ARG, "speaker_id required for a multi-speaker model");
  HIPCHK(hipSetDevice(m->device));
  Carver cv(d_workspace, workspace_bytes);
  DecWs w; carve_dec(cv, m, B, T_in, n_steps, w);
  if (!cv.ok()) return fail(TACO_ERR_STATE, "workspace too small: need %zu bytes", cv.off);
  return decoder_forward(m, (hipStream_t)hip_stream, d_encoder_out, d_speaker_id, B, T_in, n_steps, d_manual_alignments,
                         d_teacher_frames, d_mel, d_alignments, d_stop_step, d_dbg_states, w, false, nullptr);
}

int taco_postnet_forward(taco_model* m, void* hip_stream, const float* d_mel, const int32_t* d_speaker_id, int B, int T_mel,
                         float* d_linear, float* d_post_out, void* d_workspace, size_t workspace_bytes) {
  if (m && m->finalized && B > 64 && T_mel > 0 && d_mel && d_linear) {      // passes of at most 64 rows
    const PassPlan pp = pass_plan(B);
    for (int b0 = 0; b0 < B; b0 += pp.rows)
      TRY(taco_postnet_forward(m, hip_stream, d_mel + (size_t)b0 * T_mel * m->hp.num_mels, d_speaker_id ? d_speaker_id + b0 : nullptr, std::min(pp.rows, B - b0), T_mel,
                               d_linear + (size_t)b0 * T_mel * m->hp.num_freq, d_post_out ? d_post_out + (size_t)b0 * T_mel * 2 * m->hp.post_rnn_size : nullptr,
                               d_workspace, workspace_bytes));
    return 0;
  }
  TRY(check_common(m, B, T_mel));
  if (!d_mel || !d_linear || !d_workspace) return fail(TACO_ERR_ARG, "null buffer");
  HIPCHK(hipSetDevice(m->device));
  Carver cv(d_workspace, workspace_bytes);
  PostWs w; carve_post(cv, m, B, T_mel, w);
  if (!cv.ok()) return fail(TACO_ERR_STATE, "workspace too small: need %zu bytes", cv.off);
  return postnet_forward(m, (hipStream_t)hip_stream, d_mel, d_speaker_id, B, T_mel, d_linear, d_post_out, w);
}

static const ConvL* find_conv(taco_model* m, const char* layer) {
  auto it = m->convs.find(layer ? layer : "");
  return it == m->convs.end() ? nullptr : &it->second;
}

int taco_conv1d_bn_f32(taco_model* m, void* hip_stream, const char* layer, const float* d_x, int B, int T, int act,
                       int maxpool_width, float* d_out) {
  if (!m || !m->finalized) return fail(TACO_ERR_STATE, "model not finalized");
  const ConvL* L = find_conv(m, layer);
  if (!L) return fail(TACO_ERR_ARG, "unknown conv layer '%s'", layer ? layer : "(null)");
  if (!d_x || !d_out || B <= 0 || T <= 0) return fail(TACO_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->device));
  GemmCall g; g.x = d_x; g.ldx = L->cin; g.M = B * T; g.T = T; g.act = act; g.mpw = maxpool_width < 1 ? 1 : maxpool_width;
  g.out = d_out; g.ldo = L->N;
  return run_gemm(m, (hipStream_t)hip_stream, L, 1, false, g);
}

int taco_dense_f32(taco_model* m, void* hip_stream, const char* layer, const float* d_x, int rows, int act, float* d_out) {
  if (!m || !m->finalized) return fail(TACO_ERR_STATE, "model not finalized");
  if (!d_x || !d_out || rows <= 0) return fail(TACO_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->device));
  auto sk = m->skinny.find(layer ? layer : "");
  if (sk != m->skinny.end()) {
    SkJob j = sk_linear(m, sk->second, d_x, sk->second.K, sk->second.K, nullptr, 0, act, d_out, sk->second.N);
    return run_skinny((hipStream_t)hip_stream, rows, &j, 1);
  }
  const ConvL* L = find_conv(m, layer);
  if (!L || L->kw != 1) return fail(TACO_ERR_ARG, "unknown dense layer '%s'", layer ? layer : "(null)");
  GemmCall g; g.x = d_x; g.ldx = L->cin; g.M = rows; g.act = act; g.out = d_out; g.ldo = L->N;
  return run_gemm(m, (hipStream_t)hip_stream, L, 1, false, g);
}

int taco_highway_f32(taco_model* m, void* hip_stream, const char* layer, const float* d_x, int rows, float* d_out) {
  if (!m || !m->finalized) return fail(TACO_ERR_STATE, "model not finalized");
  const ConvL* L = find_conv(m, layer);
  if (!L || !L->wp2) return fail(TACO_ERR_ARG, "unknown highway layer '%s'", layer ? layer : "(null)");
  if (!d_x || !d_out || rows <= 0 || d_x == d_out) return fail(TACO_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->device));
  GemmCall g; g.x = d_x; g.ldx = L->cin; g.M = rows; g.out = d_out; g.ldo = L->N;
  return run_gemm(m, (hipStream_t)hip_stream, L, 1, true, g);
}

int taco_bigru_f32(taco_model* m, void* hip_stream, const char* scope, const float* d_x, const int32_t* d_lengths,
                   const float* d_init_state, int B, int T, float* d_out, void* d_workspace, size_t workspace_bytes) {
  TRY(check_common(m, B, T));
  const std::string sc = scope ? scope : "";
  const Cbhg* c = sc == "encoder_cbhg" ? &m->enc : (sc == "post_cbhg" ? &m->post : nullptr);
  if (!c) return fail(TACO_ERR_ARG, "unknown BiGRU scope '%s'", sc.c_str());
  if (!d_x || !d_out || !d_workspace) return fail(TACO_ERR_ARG, "null buffer");
  HIPCHK(hipSetDevice(m->device));
  hipStream_t st = (hipStream_t)hip_stream;
  Carver cv(d_workspace, workspace_bytes);
  CbhgWs w; carve_cbhg(cv, *c, B, T, w);
  if (!cv.ok()) return fail(TACO_ERR_STATE, "workspace too small: need %zu bytes", cv.off);
  { const int H = c->rnn;   // hoisted input projection (backward columns stored time-reversed), then the scan
    GemmCall xp; xp.x = d_x; xp.ldx = H; xp.M = B * T; xp.T = T; xp.out = w.xproj; xp.ldo = 6 * H;
    xp.rev_len = d_lengths; xp.rev_col0 = 3 * H;
    TRY(run_gemm(m, st, &c->xproj, 1, false, xp)); }
  return bigru_scan(m, st, *c, B, T, d_lengths, d_init_state, d_out, w);
}

int taco_attention_step_f32(taco_model* m, void* hip_stream, const float* d_cell_output, const float* d_keys,
                            const float* d_values, const float* d_prev_alignments, int B, int T_in, float* d_alignments,
                            float* d_context, void* d_workspace, size_t workspace_bytes) {
  TRY(check_common(m, B, T_in));
  if (!d_cell_output || !d_keys || !d_values || !d_prev_alignments || !d_alignments || !d_context || !d_workspace)
    return fail(TACO_ERR_ARG, "null buffer");
  if (T_in > ATT_MAXT) return fail(TACO_ERR_UNSUPPORTED, "T_in %d > %d", T_in, ATT_MAXT);
  HIPCHK(hipSetDevice(m->device));
  hipStream_t st = (hipStream_t)hip_stream;
  const taco_hparams& hp = m->hp;
  const int A = hp.attention_size, D = 2 * hp.enc_rnn_size, As = hp.attention_state_size;
  Carver cv(d_workspace, workspace_bytes);
  float* q = cv.f((size_t)B * A);
  if (!cv.ok()) return fail(TACO_ERR_STATE, "workspace too small");
  SkJob j = sk_linear(m, m->query, d_cell_output, As, As, nullptr, 0, ACT_NONE, q, A);
  TRY(run_skinny(st, B, &j, 1));
  if (d_alignments != d_prev_alignments)
    HIPCHK(hipMemcpyAsync(d_alignments, d_prev_alignments, (size_t)B * T_in * sizeof(float), hipMemcpyDeviceToDevice, st));
  AttnArgs a; memset(&a, 0, sizeof a);
  a.q = q; a.keys = d_keys; a.values = d_values; a.v = AP(m, m->att_v); a.battn = AP(m, m->att_b); a.score_bias = AP(m, m->att_sb);
  a.align = d_alignments; a.ctx = d_context; a.ldctx = D; a.T_in = T_in; a.A = A; a.D = D; a.type = hp.attention_type; a.n_steps = 1;
  hipLaunchKernelGGL(k_attention, dim3(B), dim3(64 * ATT_NW), 0, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

/* One GRUCell step of a decoder GRU by name ("decoder/attention_gru", "decoder/gru_1", ...):
 * d_h [R,H] is updated in place; d_out_res (nullable) = h' + x (ResidualWrapper).  Workspace: 3*R*H floats. */
int taco_gru_cell_f32(taco_model* m, void* hip_stream, const char* name, const float* d_x, float* d_h, int R,
                      float* d_out_res, void* d_workspace, size_t workspace_bytes) {
  if (!m || !m->finalized) return fail(TACO_ERR_STATE, "model not finalized");
  auto it = m->grus.find(name ? name : "");
  if (it == m->grus.end()) return fail(TACO_ERR_ARG, "unknown GRU '%s'", name ? name : "(null)");
  if (!d_x || !d_h || !d_workspace || R <= 0) return fail(TACO_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(m->device));
  const GruDec& g = it->second;
  Carver cv(d_workspace, workspace_bytes);
  float* rh = cv.f((size_t)R * g.H); float* u = cv.f((size_t)R * g.H); float* xc = cv.f((size_t)R * g.H);
  if (!cv.ok()) return fail(TACO_ERR_STATE, "workspace too small");
  if (d_out_res && g.I != g.H) return fail(TACO_ERR_SHAPE, "residual needs input size == state size");
  return run_gru_cell(m, (hipStream_t)hip_stream, g, R, d_x, g.I, d_h, rh, u, xc, d_out_res);
}

/* add_loss (tacotron.py:274-302) on device tensors.  d_losses[4] = loss, mel_loss, linear_loss, loss_without_coeff.
 * Workspace: 2 * 1024 * 4 doubles. */
int taco_loss_f32(void* hip_stream, const float* d_mel_out, const float* d_mel_tgt, const float* d_lin_out,
                  const float* d_lin_tgt, const float* d_loss_coeff, int B, int T, int num_mels, int num_freq,
                  int prioritize_loss, int sample_rate, float* d_losses, void* d_workspace, size_t workspace_bytes) {
  if (!d_mel_out || !d_mel_tgt || !d_lin_out || !d_lin_tgt || !d_losses || !d_workspace || B <= 0 || T <= 0)
    return fail(TACO_ERR_ARG, "bad argument");
  if (workspace_bytes < (size_t)2 * TR_MAXBLK * 4 * sizeof(double)) return fail(TACO_ERR_STATE, "workspace too small");
  hipStream_t st = (hipStream_t)hip_stream;
  double* pm = (double*)d_workspace; double* pl = pm + (size_t)TR_MAXBLK * 4;
  const int rows = B * T;
  int lo = 0, hi = 0;
  if (prioritize_loss) {   // tacotron.py:283-285
    hi = (int)(5000 / (sample_rate * 0.5) * num_freq);
    lo = (int)(165 / (sample_rate * 0.5) * num_freq);
  }
  const int nbm = std::min(TR_MAXBLK, cdiv(rows * num_mels, TR_NT * 8)), nbl = std::min(TR_MAXBLK, cdiv(rows * num_freq, TR_NT * 8));
  hipLaunchKernelGGL(k_l1_partial, dim3(nbm), dim3(TR_NT), 0, st, d_mel_out, d_mel_tgt, d_loss_coeff, rows, T, num_mels, 0, 0, pm);
  hipLaunchKernelGGL(k_l1_partial, dim3(nbl), dim3(TR_NT), 0, st, d_lin_out, d_lin_tgt, d_loss_coeff, rows, T, num_freq, lo, hi, pl);
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(TR_NT), 0, st, pm, nbm, pl, nbl, (double)rows * num_mels, (double)rows * num_freq,
                     (double)rows * std::max(hi - lo, 1), prioritize_loss, d_losses);
  HIPCHK(hipGetLastError());
  return 0;
}

/* learning rate of tacotron.py:313-325 for `global_step` (0-based count of completed updates). */
float taco_learning_rate(long long global_step, float initial_learning_rate, int decay_learning_rate_mode, int is_randomly_initialized) {
  const double step = (double)(global_step + 1);
  if (decay_learning_rate_mode == 0) {
    const double w = is_randomly_initialized ? 4000.0 : 40000.0;
    return (float)(initial_learning_rate * std::sqrt(w) * std::min(step * std::pow(w, -1.5), std::pow(step, -0.5)));
  }
  return (float)(initial_learning_rate * std::pow(0.95, step / 3000.0));
}

/* add_optimizer's update (tacotron.py:327-336) on flat buffers of n floats: clip_by_global_norm(clip_norm), then
 * tf.train.AdamOptimizer(lr, beta1, beta2) in TF form.  `global_step` = completed updates so far (Adam's t = +1).
 * d_gnorm_out (nullable) receives the global gradient norm.  Workspace: 1024 doubles. */
int taco_adam_step_f32(void* hip_stream, float* d_params, const float* d_grads, float* d_m, float* d_v, size_t n,
                       long long global_step, float learning_rate, float beta1, float beta2, float epsilon, float clip_norm,
                       float* d_gnorm_out, void* d_workspace, size_t workspace_bytes) {
  if (!d_params || !d_grads || !d_m || !d_v || !d_workspace || n == 0) return fail(TACO_ERR_ARG, "bad argument");
  if (workspace_bytes < (size_t)TR_MAXBLK * sizeof(double)) return fail(TACO_ERR_STATE, "workspace too small");
  if ((reinterpret_cast<uintptr_t>(d_grads) & 15) != 0) return fail(TACO_ERR_ARG, "gradient buffer must be 16-byte aligned");
  hipStream_t st = (hipStream_t)hip_stream;
  double* part = (double*)d_workspace;
  const int nblk = (int)std::min<size_t>(TR_MAXBLK, (n + TR_NT * 16 - 1) / (TR_NT * 16));
  const double t = (double)(global_step + 1);
  const float lr_t = (float)((double)learning_rate * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t)));
  hipLaunchKernelGGL(k_sumsq_partial, dim3(nblk), dim3(TR_NT), 0, st, d_grads, n, part);
  hipLaunchKernelGGL(k_adam, dim3(std::max(nblk, 1) * 2), dim3(TR_NT), 0, st, d_params, d_grads, d_m, d_v, n, part, nblk, lr_t, beta1,
                     beta2, epsilon, clip_norm, d_gnorm_out);
  HIPCHK(hipGetLastError());
  return 0;
}

#include "taco_train_api.h"
#include "taco_audio_api.h"

}  // extern "C"
