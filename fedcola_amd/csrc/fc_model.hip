// Host-side driver of the client step: parameter layout (reference state_dict order), workspace carving, and the
// forward / backward / step launch sequences.  Everything is enqueued on the caller's stream; nothing allocates.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fedcola_hip.h"
#include "fc_kernels.h"

// ---------------------------------------------------------------- error string
static thread_local char g_err[1024] = "";
void fc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* fc_last_error(void) { return g_err; }
extern "C" int fc_abi_version(void) { return FC_ABI_VERSION; }

// ---------------------------------------------------------------- layout
struct LinearP {
  int64_t w = -1, b = -1, scale = -1, aux = -1;
  int out = 0, in = 0;
  int seg_w = -1;
  bool shared = false;   // colearn_param == 'attn': this tower USES another tower's linear; its weight gradients go to the side buffer
};
struct BlockP {
  int64_t n1w, n1b, n2w, n2b;
  LinearP qkv, proj, fc1, fc2;
};
struct TowerP {
  bool present = false;
  int64_t pos = -1, cls = -1, pw = -1, pb = -1;                    // img
  int64_t word = -1, tpos = -1, ttype = -1, lnw = -1, lnb = -1;    // txt
  std::vector<BlockP> blocks;
  int64_t head_w = -1, head_b = -1;
  int task = 0, ncls = 0;
};
// memo of the last forward (the autograd engine calls fc_backward from another thread, so this lives in the handle)
struct LastFwd { const void* ws = nullptr; int B = 0, n_txt = 0, feat_out = 0; const float* droppath = nullptr; const int64_t* ids = nullptr; };
struct fc_model {
  mutable LastFwd last;
  mutable int last_ntxt = 0;
  mutable std::vector<FcTnProblem> probs_host;   // last uploaded grouped-GEMM table (+ where it lives on the device)
  mutable const void* probs_dev = nullptr;
  mutable std::vector<FcLnReduce> ln_host;        // same for the grouped LayerNorm-gradient reduction
  mutable std::vector<char> prox_host;            // same for the FedProx chunk table
  mutable const void* prox_dev = nullptr;
  mutable std::vector<char> clip_host;            // ... and for the gradient-clipping chunk table
  mutable const void* clip_dev = nullptr;
  mutable const void* ln_dev = nullptr;
  // device tables of the grouped launches (weight-gradient problems, LayerNorm reductions): owned by the handle, so that a cached
  // table can never be overwritten behind the cache's back (a caller-owned workspace can be freed and come back at the same
  // address with other contents)
  mutable void* tables_dev = nullptr;
  mutable size_t tables_bytes = 0;
  // fused optimizer (fc_client_step): which segments the weight-gradient epilogue steps, and the chunk table of everything else
  mutable std::vector<std::pair<int64_t, int64_t>> zero_runs;   // fc_client_step: (offset, count) runs of the gradient buffer that must be zeroed
  mutable int cover_B = -1, cover_ntxt = -1;
  mutable std::vector<char> fused_host, rest_host;
  mutable bool rest_valid = false;                // rest_host / rest_chunks describe fused_host
  mutable int rest_chunks = 0;
  // the two towers are independent until the loss: the text tower runs on a side stream, forked/joined with events
  mutable hipStream_t side = nullptr;
  mutable hipStream_t cap = nullptr;              // private stream layer graphs are captured on (never executes anything)
  mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // weight gradients are launched in chunks (every few layers) on their own stream, under the rest of the backward
  mutable hipStream_t dws = nullptr;
  mutable hipEvent_t ev_dw_in = nullptr, ev_dw_in2[3] = {nullptr, nullptr, nullptr}, ev_dw_out = nullptr, ev_dw_prev = nullptr;
  mutable hipEvent_t ev_dw_last = nullptr, ev_zero = nullptr;      // the last chunk when it runs on a chain's stream; the step's zero-fills on the text stream
  // second micro-batch of the image tower
  mutable hipStream_t mbs[3] = {nullptr, nullptr, nullptr};      // micro-batch chains 1..3 (chain 0 runs on the caller's stream)
  mutable hipEvent_t ev_mb_join[3] = {nullptr, nullptr, nullptr};
  ~fc_model() {
    for (auto& kv : step_graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (adamw_dyn) (void)hipFree(adamw_dyn);
    if (tables_dev) (void)hipFree(tables_dev);
    if (shared_dev) (void)hipFree(shared_dev);
    if (reparam_dev) (void)hipFree(reparam_dev);
    if (ev_dw_in) (void)hipEventDestroy(ev_dw_in);
    for (int k = 0; k < 3; ++k) {
      if (ev_dw_in2[k]) (void)hipEventDestroy(ev_dw_in2[k]);
      if (ev_mb_join[k]) (void)hipEventDestroy(ev_mb_join[k]);
    }
    if (ev_dw_out) (void)hipEventDestroy(ev_dw_out);
    if (ev_dw_prev) (void)hipEventDestroy(ev_dw_prev);
    if (ev_dw_last) (void)hipEventDestroy(ev_dw_last);
    if (ev_zero) (void)hipEventDestroy(ev_zero);
    if (cap) (void)hipStreamDestroy(cap);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
  }
  fc_model_cfg cfg;
  std::vector<fc_segment> segs;
  int64_t total = 0;
  TowerP tw[2];
  int64_t normw = -1, normb = -1;
  // colearn_param == 'attn': [shared_lo, shared_hi) = span of the flat buffer that holds the shared Attention linears (the second
  // tower writes its dW / db into a workspace buffer of that span, added to the gradients once both towers are done)
  int64_t shared_lo = 0, shared_hi = 0;
  std::vector<FcProxChunk> shared_chunks;
  mutable void* shared_dev = nullptr;
  // re-param linears: device table for the grouped fold / gradient-routing launches (rebuilt when a trainable flag changes)
  mutable std::vector<FcReparam> reparam_host;
  mutable void* reparam_dev = nullptr;
  int dt;  // FC_F32 / FC_BF16 activation + compute-weight type
  bool need_wc;
  // fused MLP (fc_mlp.hip): the bf16 mode of a 384-wide model reads fc1 / fc2 from streams packed in MFMA-fragment order, kept behind the
  // compute weights in the same buffer: per tower and layer a forward and a backward stream of 2 * D * Hd bf16 each
  bool mlp_fused = false;
  bool mlp_fused_ok = false;      // the shape is covered (bf16, D = 384, Hd % 128 == 0); mlp_fused: ... and switched on (fc_model_set_option)
  mutable bool step_graph = false;
  // whole-step HIP graphs (fc_client_step): one executable graph per set of buffer addresses and batch shape; the three AdamW constants that
  // change from step to step live in `adamw_dyn` (device) and are rewritten ahead of every replay
  struct StepKey {      // everything a captured step bakes into its kernel arguments (lr, step and the bias corrections travel through adamw_dyn)
    const void* p[12]; size_t ws_bytes; int B, n_txt; float beta1, beta2, eps, weight_decay;
    bool operator<(const StepKey& o) const { return memcmp(this, &o, sizeof(StepKey)) < 0; }
  };
  struct StepEntry { int seen = 0; hipGraphExec_t exec = nullptr; bool declined = false; };
  mutable long step_graph_hits = 0;      // steps that ran as a hipGraphLaunch (tests: a capture that silently declines must be visible)
  mutable std::map<StepKey, StepEntry> step_graphs;
  mutable float* adamw_dyn = nullptr;
#ifdef FC_PROBES      // the fused MLP and the whole-step graph are tools-build experiments (profiles/r05: exact, not faster in the step)
  size_t mlp_stream_elems() const { return fc_mlp_pack_elems(cfg.dim, cfg.mlp_hidden); }
  size_t mlp_pack_base() const { return ((size_t)total * fc_esize(dt) + 255) / 256 * 256; }      // byte offset of the first stream inside wc
  // streams exist only while the option is on, and only for the towers the model has
  size_t mlp_pack_bytes() const { return mlp_fused ? (size_t)((tw[0].present ? 1 : 0) + (tw[1].present ? 1 : 0)) * cfg.depth * 2 * mlp_stream_elems() * sizeof(bf16_t) : 0; }
  const bf16_t* mlp_stream(const void* wc, int tower, int layer, int bwd) const {
    const int slot = (tower == 1 && tw[0].present) ? 1 : 0;
    return (const bf16_t*)((const char*)wc + mlp_pack_base()) + ((size_t)(slot * cfg.depth + layer) * 2 + bwd) * mlp_stream_elems();
  }
#endif
  int64_t add(const std::string& name, std::vector<int64_t> shape, int trainable = 1) {
    fc_segment s;
    memset(&s, 0, sizeof(s));
    snprintf(s.name, sizeof(s.name), "%s", name.c_str());
    s.offset = total;
    s.numel = 1;
    s.ndim = (int)shape.size();
    for (size_t i = 0; i < shape.size(); ++i) { s.shape[i] = shape[i]; s.numel *= shape[i]; }
    s.trainable = trainable;
    segs.push_back(s);
    total += (s.numel + 63) / 64 * 64;
    return s.offset;
  }
};

static void add_linear(fc_model* m, LinearP& L, const std::string& pre, int out, int in, bool reparam, bool aux_trained) {
  L.out = out; L.in = in;
  L.seg_w = (int)m->segs.size();
  L.w = m->add(pre + ".weight", {out, in});
  L.b = m->add(pre + ".bias", {out});
  if (reparam) {
    L.scale = m->add(pre + ".cross_modal_scale", {1});
    L.aux = m->add(pre + ".aux_weight", {out, in}, aux_trained ? 1 : 0);
  }
}

extern "C" int fc_model_create(const fc_model_cfg* c, fc_model_t** out) {
  FC_REQUIRE(c && out, "fc_model_create: null argument");
  FC_REQUIRE(c->has_img || c->has_txt, "fc_model_create: no modality");
  FC_REQUIRE(c->dim > 0 && c->heads > 0 && c->dim % c->heads == 0, "fc_model_create: dim %d not divisible by heads %d", c->dim, c->heads);
  FC_REQUIRE(c->depth > 0 && c->mlp_hidden > 0, "fc_model_create: bad depth/mlp_hidden");
  FC_REQUIRE(!(c->aux_attn_only && c->aux_mlp_only), "Both aux_attn_only and aux_mlp_only cannot be True.");  // mome.py:779
  FC_REQUIRE(c->precision == FC_PREC_FP32 || c->precision == FC_PREC_BF16, "fc_model_create: bad precision");
  if (c->has_img) FC_REQUIRE(c->img_size % c->patch == 0 && c->in_chans > 0, "fc_model_create: bad image geometry");
  fc_model* m = new fc_model();
  m->cfg = *c;
  m->dt = c->precision == FC_PREC_BF16 ? FC_BF16 : FC_F32;
  const int D = c->dim;
  bool uni = !(c->has_img && c->has_txt);
  bool aux = c->with_aux && uni;                       // mome.py:768
  bool aux_attn = aux && !c->aux_mlp_only, aux_mlp = aux && !c->aux_attn_only;
  m->need_wc = (m->dt == FC_BF16) || aux;
#ifdef FC_PROBES
  {
    static const int on = fc_knob("FC_MLP_FUSED", 0), graph = fc_knob("FC_STEP_GRAPH", 0);
    m->mlp_fused_ok = m->dt == FC_BF16 && fc_mlp_fused_ok(c->dim, c->mlp_hidden);
    m->mlp_fused = on && m->mlp_fused_ok;
    m->step_graph = graph != 0;
  }
#endif
  int present[2] = {c->has_img, c->has_txt};
  // embeddings first (mome.py:709-723)
  for (int i = 0; i < 2; ++i) {
    TowerP& t = m->tw[i];
    t.present = present[i];
    if (!t.present) continue;
    std::string e = "embeddings." + std::to_string(i);
    if (i == 0) {
      int np = (c->img_size / c->patch) * (c->img_size / c->patch);
      t.pos = m->add(e + ".pos_embed", {1, np + 1, D});
      t.cls = m->add(e + ".cls_token", {1, 1, D});
      t.pw = m->add(e + ".embed.proj.weight", {D, c->in_chans, c->patch, c->patch});
      t.pb = m->add(e + ".embed.proj.bias", {D});
      t.task = c->task_img; t.ncls = c->num_classes_img;
    } else {
      e += ".text_embeddings";
      t.word = m->add(e + ".word_embeddings.weight", {c->vocab, D});
      t.tpos = m->add(e + ".position_embeddings.weight", {c->max_text_len, D});
      t.ttype = m->add(e + ".token_type_embeddings.weight", {2, D});
      t.lnw = m->add(e + ".LayerNorm.weight", {D});
      t.lnb = m->add(e + ".LayerNorm.bias", {D});
      t.task = c->task_txt; t.ncls = c->num_classes_txt;
    }
  }
  // blocks (mome.py:729-750)
  for (int i = 0; i < 2; ++i) {
    TowerP& t = m->tw[i];
    if (!t.present) continue;
    t.blocks.resize(c->depth);
    for (int l = 0; l < c->depth; ++l) {
      std::string p = "blockses." + std::to_string(i) + "." + std::to_string(l);
      BlockP& b = t.blocks[l];
      b.n1w = m->add(p + ".norm1.weight", {D});
      b.n1b = m->add(p + ".norm1.bias", {D});
      if (i == 1 && c->colearn_attn && m->tw[0].present) {   // mome.py:836-840: block.attn = self.blockses[main_idx][j].attn
        b.qkv = m->tw[0].blocks[l].qkv;
        b.proj = m->tw[0].blocks[l].proj;
        b.qkv.shared = b.proj.shared = true;
      } else {
        add_linear(m, b.qkv, p + ".attn.qkv", 3 * D, D, aux_attn, c->aux_trained);
        add_linear(m, b.proj, p + ".attn.proj", D, D, aux_attn, c->aux_trained);
      }
      b.n2w = m->add(p + ".norm2.weight", {D});
      b.n2b = m->add(p + ".norm2.bias", {D});
      add_linear(m, b.fc1, p + ".mlp.fc1", c->mlp_hidden, D, aux_mlp, c->aux_trained);
      add_linear(m, b.fc2, p + ".mlp.fc2", D, c->mlp_hidden, aux_mlp, c->aux_trained);
    }
  }
  m->normw = m->add("norm.weight", {D});
  m->normb = m->add("norm.bias", {D});
  if (c->colearn_attn && c->has_img && c->has_txt) {
    m->shared_lo = m->tw[0].blocks[0].qkv.w;
    for (const BlockP& b : m->tw[0].blocks)
      for (const LinearP* L : {&b.qkv, &b.proj})
        for (int64_t off : {L->w, L->b}) {
          const int64_t n = off == L->w ? (int64_t)L->out * L->in : L->out;
          for (int64_t o = 0; o < n; o += FC_PROX_CHUNK) m->shared_chunks.push_back(FcProxChunk{off + o, (int32_t)std::min<int64_t>(FC_PROX_CHUNK, n - o), 0});
          m->shared_hi = std::max<int64_t>(m->shared_hi, (off + n + 63) / 64 * 64);
        }
  }
  for (int i = 0; i < 2; ++i) {
    TowerP& t = m->tw[i];
    if (!t.present || t.task != FC_TASK_CLS || t.ncls <= 0) continue;
    std::string h = "heads." + std::to_string(i) + ".head";
    t.head_w = m->add(h + ".weight", {t.ncls, D});
    t.head_b = m->add(h + ".bias", {t.ncls});
  }
  *out = m;
  return 0;
}
extern "C" void fc_model_destroy(fc_model_t* m) { delete m; }
extern "C" int64_t fc_model_num_params(const fc_model_t* m) { return m->total; }
extern "C" int32_t fc_model_num_segments(const fc_model_t* m) { return (int32_t)m->segs.size(); }
extern "C" int fc_model_segment(const fc_model_t* m, int32_t i, fc_segment* out) {
  FC_REQUIRE(i >= 0 && i < (int)m->segs.size(), "fc_model_segment: index %d out of range", i);
  *out = m->segs[i];
  return 0;
}
extern "C" int fc_model_set_trainable(fc_model_t* m, int32_t seg, int32_t trainable) {
  for (auto& kv : m->step_graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);      // a captured step bakes the optimizer's coverage in
  m->step_graphs.clear();
  FC_REQUIRE(seg >= 0 && seg < (int)m->segs.size(), "fc_model_set_trainable: index %d out of range", seg);
  m->segs[seg].trainable = trainable;
  return 0;
}
#ifdef FC_PROBES
// Run-time switches of a handle, tools build only (include/fedcola_hip.h: FC_OPT_*): the round-5 experiments that are exact but do not pay in the step.
extern "C" long fc_dbg_step_graph_hits(const fc_model_t* m) { return m->step_graph_hits; }
extern "C" int fc_model_set_option(fc_model_t* m, int32_t option, int32_t value) {
  switch (option) {
    case FC_OPT_MLP_FUSED:
      FC_REQUIRE(!value || m->mlp_fused_ok, "FC_OPT_MLP_FUSED: the fused MLP covers the bf16 mode of 384-wide models with mlp_hidden %% 128 == 0");
      // switching it on needs the packed streams: they are (re)written by the next fc_prepare_weights / optimizer step; the caller must
      // call fc_prepare_weights before the next forward
      m->mlp_fused = value != 0;
      for (auto& kv : m->step_graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      m->step_graphs.clear();
      return 0;
    case FC_OPT_STEP_GRAPH:
      m->step_graph = value != 0;
      return 0;
    case FC_OPT_GEMM_FORM:
      FC_REQUIRE(value == 0 || value == 64 || value == 3 || value == 4, "FC_OPT_GEMM_FORM: 0 (128-row tiles), 64 (64-row tiles) or 3 / 4 (64-row tiles, 3- / 4-stage ring)");
      fc_gemm_set_form(value);
      return 0;
  }
  FC_REQUIRE(false, "fc_model_set_option: unknown option %d", option);
}
#endif
extern "C" size_t fc_compute_weights_bytes(const fc_model_t* m) {
  if (!m->need_wc) return 0;
#ifdef FC_PROBES
  if (m->mlp_fused) return m->mlp_pack_base() + m->mlp_pack_bytes();
#endif
  return (size_t)m->total * fc_esize(m->dt);
}

// ---------------------------------------------------------------- workspace
struct LayerWs {
  float *mean1, *rstd1, *mean2, *rstd2, *lse;
  void *h1, *qkv, *o, *xmid, *h2, *u, *gact;
  // backward: every layer keeps its own dY tensors so that all weight gradients can be computed in one grouped launch
  void *gxmid, *gdm, *gda, *gdu, *gdqkv;
};
struct TowerWs {
  int M = 0, N = 0;
  int dp_off = 0;          // first sample of this (micro-batch) view in the drop-path table
  size_t ln_stride = 0;    // elements between the partial sets of consecutive LayerNorm instances (sized for the full batch)
  void *patches = nullptr, *dtok = nullptr;
  float *emb_mean = nullptr, *emb_rstd = nullptr;
  std::vector<void*> x;
  std::vector<void*> gx;   // gradient w.r.t. x[l]
  std::vector<LayerWs> L;
  float *f, *hmean, *hrstd, *nrm, *out, *logits, *dlogits, *df;
  void *dh, *dO;          // backward temporaries (per tower: the towers run concurrently)
  float *delta;
  fc_ln_part_t *ln_partial;   // [2*depth][blocks][2*D] LayerNorm-backward partial sums (fp64)
};
struct Ws {
  TowerWs t[2];
  float *loss_scratch, *dout[2];
  FcTnProblem* probs;   // device array for the grouped weight-gradient launch
  FcLnReduce* lntab;    // device array for the grouped LayerNorm-gradient reduction
  int max_probs, max_ln;
  int B, n_txt, feat_out;
  int dp_stride = 0;       // drop-path table: samples per row
  const float* droppath;
  const int64_t* ids;
  float* shared_g = nullptr;       // colearn_param == 'attn': the second tower's gradients of the shared Attention linears
  size_t bytes;
};
struct Bump {
  char* base;
  size_t off = 0;
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return p;
  }
};
struct WsHeader {  // persisted at the start of the workspace so fc_backward knows what the forward did
  int32_t B, n_txt, feat_out, magic;
  const float* droppath;
  const int64_t* ids;
};

static void carve(const fc_model* m, int B, int n_txt, void* base, Ws& w) {
  const fc_model_cfg& c = m->cfg;
  Bump bp{(char*)base};
  bp.take(sizeof(WsHeader));
  size_t es = fc_esize(m->dt);
  const int D = c.dim, Hd = c.mlp_hidden;
  int maxM = 0;
  for (int i = 0; i < 2; ++i) {
    TowerWs& t = w.t[i];
    if (!m->tw[i].present) continue;
    t.N = i == 0 ? (c.img_size / c.patch) * (c.img_size / c.patch) + 1 : n_txt;
    t.M = B * t.N;
    if (t.M > maxM) maxM = t.M;
    if (i == 0) {
      size_t kp = (size_t)c.in_chans * c.patch * c.patch;
      t.patches = bp.take((size_t)B * (t.N - 1) * kp * es);
      t.dtok = bp.take((size_t)B * (t.N - 1) * D * es);
    } else {
      t.emb_mean = (float*)bp.take(sizeof(float) * t.M);
      t.emb_rstd = (float*)bp.take(sizeof(float) * t.M);
    }
    t.x.resize(c.depth + 1);
    t.gx.resize(c.depth + 1);
    t.L.resize(c.depth);
    for (int l = 0; l <= c.depth; ++l) t.x[l] = bp.take((size_t)t.M * D * es);
    for (int l = 0; l <= c.depth; ++l) t.gx[l] = bp.take((size_t)t.M * D * es);
    for (int l = 0; l < c.depth; ++l) {
      LayerWs& L = t.L[l];
      L.mean1 = (float*)bp.take(sizeof(float) * t.M);
      L.rstd1 = (float*)bp.take(sizeof(float) * t.M);
      L.mean2 = (float*)bp.take(sizeof(float) * t.M);
      L.rstd2 = (float*)bp.take(sizeof(float) * t.M);
      L.lse = (float*)bp.take(sizeof(float) * (size_t)B * c.heads * t.N);
      L.h1 = bp.take((size_t)t.M * D * es);
      L.qkv = bp.take((size_t)t.M * 3 * D * es);
      L.o = bp.take((size_t)t.M * D * es);
      L.xmid = bp.take((size_t)t.M * D * es);
      L.h2 = bp.take((size_t)t.M * D * es);
      L.u = bp.take((size_t)t.M * Hd * es);
      L.gact = bp.take((size_t)t.M * Hd * es);
      L.gxmid = bp.take((size_t)t.M * D * es);
      L.gdm = bp.take((size_t)t.M * D * es);
      L.gda = bp.take((size_t)t.M * D * es);
      L.gdu = bp.take((size_t)t.M * Hd * es);
      L.gdqkv = bp.take((size_t)t.M * 3 * D * es);
    }
    int nc = m->tw[i].ncls > 0 ? m->tw[i].ncls : 1;
    t.f = (float*)bp.take(sizeof(float) * (size_t)B * D);
    t.hmean = (float*)bp.take(sizeof(float) * B);
    t.hrstd = (float*)bp.take(sizeof(float) * B);
    t.nrm = (float*)bp.take(sizeof(float) * B);
    t.out = (float*)bp.take(sizeof(float) * (size_t)B * D);
    t.logits = (float*)bp.take(sizeof(float) * (size_t)B * nc);
    t.dlogits = (float*)bp.take(sizeof(float) * (size_t)B * nc);
    t.df = (float*)bp.take(sizeof(float) * (size_t)B * D);
    w.dout[i] = (float*)bp.take(sizeof(float) * (size_t)B * (D > nc ? D : nc));
    t.dh = bp.take((size_t)t.M * D * es);
    t.dO = bp.take((size_t)t.M * D * es);
    t.delta = (float*)bp.take(sizeof(float) * (size_t)B * c.heads * t.N);
    t.ln_stride = ((size_t)fc_layernorm_bwd_partial_blocks(t.M) + 1) * 2 * D;                  // (+1: a slice may round up once more)
    t.ln_partial = (fc_ln_part_t*)bp.take(sizeof(fc_ln_part_t) * (size_t)3 * 2 * c.depth * t.ln_stride);   // x3: micro-batch chains
  }
  w.max_probs = 2 * (4 * c.depth + 1);
  w.probs = (FcTnProblem*)bp.take(sizeof(FcTnProblem) * w.max_probs);
  w.max_ln = 12 * c.depth;
  w.lntab = (FcLnReduce*)bp.take(sizeof(FcLnReduce) * w.max_ln);
  w.loss_scratch = (float*)bp.take(sizeof(float) * (2 * (size_t)B * B + 2 * B + 64));
  w.shared_g = m->shared_hi > m->shared_lo ? (float*)bp.take(sizeof(float) * (size_t)(m->shared_hi - m->shared_lo)) : nullptr;
  w.bytes = bp.off;
  w.B = B; w.n_txt = n_txt;
  w.dp_stride = B;
}

// View of the workspace restricted to samples [b0, b0+bn) of tower i (micro-batch `mb`): every per-row pointer is advanced,
// so the unchanged forward / backward code runs on the slice.  The image tower is processed as two such slices on two
// streams: the chains are independent until the loss, and their latency-bound kernels fill each other's gaps.
static Ws slice_ws(const fc_model* m, const Ws& w, int i, int b0, int bn, int mb, bool both_towers_view = true) {
  Ws v = w;
  const fc_model_cfg& c = m->cfg;
  const size_t es = fc_esize(m->dt);
  TowerWs& t = v.t[i];
  const TowerWs& f = w.t[i];
  const int N = f.N, D = c.dim, Hd = c.mlp_hidden, H = c.heads;
  const size_t r0 = (size_t)b0 * N;
  auto adv = [&](void* p, size_t bytes_per_row) -> void* { return p ? (void*)((char*)p + r0 * bytes_per_row) : nullptr; };
  t.M = bn * N;
  if (i == 0) {
    const size_t kp = (size_t)c.in_chans * c.patch * c.patch;
    t.patches = f.patches ? (void*)((char*)f.patches + (size_t)b0 * (N - 1) * kp * es) : nullptr;
    t.dtok = f.dtok ? (void*)((char*)f.dtok + (size_t)b0 * (N - 1) * D * es) : nullptr;
  } else {
    t.emb_mean = f.emb_mean + r0;
    t.emb_rstd = f.emb_rstd + r0;
  }
  for (size_t l = 0; l < f.x.size(); ++l) { t.x[l] = adv(f.x[l], D * es); t.gx[l] = adv(f.gx[l], D * es); }
  for (size_t l = 0; l < f.L.size(); ++l) {
    const LayerWs& a = f.L[l];
    LayerWs& b = t.L[l];
    b.mean1 = a.mean1 + r0; b.rstd1 = a.rstd1 + r0; b.mean2 = a.mean2 + r0; b.rstd2 = a.rstd2 + r0;
    b.lse = a.lse + (size_t)b0 * H * N;
    b.h1 = adv(a.h1, D * es); b.o = adv(a.o, D * es); b.xmid = adv(a.xmid, D * es); b.h2 = adv(a.h2, D * es);
    b.gxmid = adv(a.gxmid, D * es); b.gdm = adv(a.gdm, D * es); b.gda = adv(a.gda, D * es);
    b.qkv = adv(a.qkv, 3 * D * es); b.gdqkv = adv(a.gdqkv, 3 * D * es);
    b.u = adv(a.u, Hd * es); b.gact = adv(a.gact, Hd * es); b.gdu = adv(a.gdu, Hd * es);
  }
  const int nc = m->tw[i].ncls > 0 ? m->tw[i].ncls : 1;
  t.f = f.f + (size_t)b0 * D; t.hmean = f.hmean + b0; t.hrstd = f.hrstd + b0; t.nrm = f.nrm + b0; t.out = f.out + (size_t)b0 * D;
  t.logits = f.logits + (size_t)b0 * nc; t.dlogits = f.dlogits + (size_t)b0 * nc; t.df = f.df + (size_t)b0 * D;
  t.dh = adv(f.dh, D * es); t.dO = adv(f.dO, D * es);
  t.delta = f.delta + (size_t)b0 * H * N;
  t.ln_partial = f.ln_partial + (size_t)mb * 2 * c.depth * f.ln_stride;
  t.dp_off = f.dp_off + b0;
  if (both_towers_view) v.B = bn;
  return v;
}
// first sample of micro-batch k of n.  Two chains split the batch 57 : 43 (FC_MB_FIRST, percent taken by the chain on the caller's
// stream), not in halves: two equal chains run the same kernel sequence in lockstep and meet in the same phases (both staging-bound GEMMs,
// then both LayerNorms ...); unequal ones drift apart.  Measured at B = 64 (ms/step, one box): 32/32 4.92, 33/31 4.93, 34/30 4.89,
// 35/29 4.84, 36/28 4.84, 37/27 4.85, 38/26 4.89; the other way round 30/34 5.00, 28/36 4.91.
static int mb_begin(int B, int k, int n) {
  static const int first = fc_knob("FC_MB_FIRST", 57);
  if (n == 2 && k == 1 && first > 0 && first < 100) { int b = (int)((long)B * first / 100); return b < 1 ? 1 : (b > B - 1 ? B - 1 : b); }
  return (int)((long)B * k / n);
}
static int microbatches(const fc_model* m, int B) {
  static int req = fc_knob("FC_MICROBATCH", 2);
  if (m->dt != FC_BF16 || req < 2 || B < 16) return 1;
  int n = req > 2 ? 2 : req;      // two chains at most: the LayerNorm-gradient reduction takes two partial sets per tensor
  if (B < 8 * n) n = 2;
  while (n > 1 && !m->mbs[n - 2]) --n;      // streams are created for the configured count only
  return n;
}

extern "C" size_t fc_workspace_bytes(const fc_model_t* m, int32_t B, int32_t n_txt) {
  Ws w;
  carve(m, B, n_txt, nullptr, w);
  return w.bytes;
}
extern "C" size_t fc_contrastive_scratch_floats(int32_t B) { return 2 * (size_t)B * B + 2 * (size_t)B + 64; }

// ---------------------------------------------------------------- weights
template <typename F>
static int for_each_linear(const fc_model* m, F f) {
  for (int i = 0; i < 2; ++i) {
    if (!m->tw[i].present) continue;
    for (const BlockP& b : m->tw[i].blocks) {
      FC_TRY(f(b.qkv)); FC_TRY(f(b.proj)); FC_TRY(f(b.fc1)); FC_TRY(f(b.fc2));
    }
  }
  return 0;
}

// device table of the re-param linears (cached in the handle; a changed trainable flag rebuilds it: one device synchronisation)
static int reparam_table(const fc_model* m, const FcReparam** tab, int* n) {
  std::vector<FcReparam> t;
  (void)for_each_linear(m, [&](const LinearP& L) -> int {
    if (L.aux >= 0) t.push_back(FcReparam{L.w, L.aux, L.scale, (int64_t)L.out * L.in, m->segs[L.seg_w + 3].trainable ? 1 : 0, 0});
    return 0;
  });
  *n = (int)t.size();
  *tab = nullptr;
  if (t.empty()) return 0;
  if (!m->reparam_dev || m->reparam_host.size() != t.size() || memcmp(m->reparam_host.data(), t.data(), t.size() * sizeof(FcReparam)) != 0) {
    FC_CHECK_HIP(hipDeviceSynchronize());
    if (m->reparam_dev) FC_CHECK_HIP(hipFree(m->reparam_dev));
    FC_CHECK_HIP(hipMalloc(&m->reparam_dev, t.size() * sizeof(FcReparam)));
    FC_CHECK_HIP(hipMemcpy(m->reparam_dev, t.data(), t.size() * sizeof(FcReparam), hipMemcpyHostToDevice));
    m->reparam_host = t;
  }
  *tab = (const FcReparam*)m->reparam_dev;
  return 0;
}
static int cached_table(const void* host, size_t bytes, const void** out);
// the fused MLP's weight streams of every layer of both towers, from the bf16 compute weights in `wc` (one launch; tools build)
static int mlp_pack_all(const fc_model* m, void* wc, hipStream_t s) {
#ifndef FC_PROBES
  return 0;
#else
  if (!m->mlp_fused) return 0;      // (a handle that switches the fused MLP on calls fc_prepare_weights next)
  std::vector<FcMlpPackJob> jobs;
  const bf16_t* W = (const bf16_t*)wc;
  for (int i = 0; i < 2; ++i) {
    if (!m->tw[i].present) continue;
    for (int l = 0; l < m->cfg.depth; ++l) {
      const BlockP& b = m->tw[i].blocks[l];
      jobs.push_back(FcMlpPackJob{W + b.fc1.w, W + b.fc2.w, (bf16_t*)m->mlp_stream(wc, i, l, 0), (bf16_t*)m->mlp_stream(wc, i, l, 1)});
    }
  }
  const void* tab = nullptr;
  FC_TRY(cached_table(jobs.data(), jobs.size() * sizeof(FcMlpPackJob), &tab));
  return fc_mlp_pack(tab, (int)jobs.size(), m->cfg.dim, m->cfg.mlp_hidden, s);
#endif
}
extern "C" int fc_prepare_weights(const fc_model_t* m, const float* params, void* wc, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!m->need_wc) return 0;
  FC_REQUIRE(wc && wc != (const void*)params, "fc_prepare_weights: a separate compute-weight buffer is required");
  FC_TRY(fc_cast(m->dt, params, wc, (size_t)m->total, s));
  const FcReparam* tab; int n;
  FC_TRY(reparam_table(m, &tab, &n));
  FC_TRY(fc_reparam_fold_grouped(m->dt, tab, n, params, wc, s));      // W + s*A of every re-param linear in one launch
  return mlp_pack_all(m, wc, s);
}

extern "C" int fc_upload_fold(const fc_model_t* m, const float* params, float* dst, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (dst != params) FC_CHECK_HIP(hipMemcpyAsync(dst, params, sizeof(float) * (size_t)m->total, hipMemcpyDeviceToDevice, s));
  return for_each_linear(m, [&](const LinearP& L) -> int {
    if (L.aux < 0) return 0;
    return fc_reparam_fold(FC_F32, params + L.w, params + L.aux, params + L.scale, dst + L.w, (size_t)L.out * L.in, s);
  });
}

// ---------------------------------------------------------------- GEMM dispatch (MFMA fast path when available)
struct Ctx {
  const fc_model* m;
  const float* params;
  const char* wc;  // compute weights (type m->dt), same element offsets as params
  hipStream_t s;
  int dt;
  size_t es;
  std::vector<FcTnProblem>* defer = nullptr;   // non-null: weight/bias gradients are queued for the grouped launch
  struct DwState* dw = nullptr;                // chunked early launches of the queued problems
  bool no_wgrad = false;                       // micro-batch slice: the full-batch weight gradients are queued by the driver
  int n_more = 0;                              // flush_dw also orders the chunk after the other micro-batch chains' streams
  std::vector<FcLnReduce>* lnq = nullptr;      // non-null: LayerNorm dgamma/dbeta partials are queued likewise
  int ln_accumulate = 1;                       // the grouped reduction adds to the gradient buffer (0: plain store, buffer not zeroed)
  float* gshared = nullptr;                    // colearn 'attn': base such that gshared + L.w is the side buffer's slot of a shared linear
  const FcAdamW* fopt = nullptr;               // non-null: the grouped weight-gradient launches also take the AdamW step of what they produce
  std::vector<char>* fused_seg = nullptr;      // ... and the segments they cover are flagged here
  const hipStream_t* more_s = nullptr;         // the n_more streams flush_dw also waits for (default: m->mbs[k])
  struct LayerRec* rec = nullptr;              // non-null while a layer is being captured into a graph: host-side notes are recorded too
  // queue the reduction of one LayerNorm backward's partial rows.  Two micro-batch chains of one tower share dg / db: one entry, two
  // partial sets (the reduction uses no atomics).  Host-only: a replayed layer graph repeats its notes (LayerRec).
  int note_ln(fc_ln_part_t* partial, float* dg, float* db, int M, int D) const;
  int ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* g, const void* res, void* dx, float* dg,
             float* db, int M, int D, fc_ln_part_t* partial, void* dx_scaled = nullptr, const float* rowscale = nullptr, int rps = 1) const {
    int r = fc_layernorm_bwd(dt, dy, x, mean, rstd, g, res, dx, dg, db, M, D, s, lnq ? partial : nullptr, dx_scaled, rowscale, rps);
    if (r == 1) return note_ln(partial, dg, db, M, D);
    if (r == 0 && rec) mark_ungraphable();     // the atomic fallback wrote dg / db itself: fine eagerly, but keep such layers out of graphs
    return r;
  }
  void mark_ungraphable() const;
  const void* W(int64_t off) const { return wc + (size_t)off * es; }
  // Y[M,N] = X[M,K] . W[N,K]^T
  int gemm_fwd(const void* X, const void* Wt, void* Y, int M, int N, int K, const GemmEpi& e) const {
    if (dt == FC_BF16) {
      int r = fc_gemm_mfma(FC_GEMM_NT, FC_BF16, (const bf16_t*)X, K, (const bf16_t*)Wt, K, Y, N, M, N, K, e, s);
      if (r <= 0) return r;
    } else if (dt == FC_F32) {
      int r = fc_gemm_x3(FC_GEMM_NT, (const float*)X, K, (const float*)Wt, K, (float*)Y, N, M, N, K, e, s);
      if (r <= 0) return r;
    }
    return fc_gemm_generic(dt, dt, dt, X, K, 1, Wt, 1, K, Y, N, M, N, K, e, s);
  }
  // dX[M,K] = dY[M,N] . W[N,K]
  int gemm_dx(const void* dY, const void* Wt, void* dX, int M, int N, int K, const GemmEpi& e) const {
    if (dt == FC_BF16) {
      int r = fc_gemm_mfma(FC_GEMM_NN, FC_BF16, (const bf16_t*)dY, N, (const bf16_t*)Wt, K, dX, K, M, K, N, e, s);
      if (r <= 0) return r;
    } else if (dt == FC_F32) {
      int r = fc_gemm_x3(FC_GEMM_NN, (const float*)dY, N, (const float*)Wt, K, (float*)dX, K, M, K, N, e, s);
      if (r <= 0) return r;
    }
    return fc_gemm_generic(dt, dt, dt, dY, N, 1, Wt, K, 1, dX, K, M, K, N, e, s);
  }
  // dW[N,K] = dY[M,N]^T . X[M,K]   (fp32 out)
  int gemm_dw(const void* dY, const void* X, float* dW, int M, int N, int K) const {
    GemmEpi e;
    e.out_zeroed = 1;  // the flat gradient buffer is zero-filled before every backward
    if (dt == FC_BF16) {
      int r = fc_gemm_mfma(FC_GEMM_TN, FC_F32, (const bf16_t*)dY, N, (const bf16_t*)X, K, dW, K, N, K, M, e, s);
      if (r <= 0) return r;
    } else if (dt == FC_F32) {
      int r = fc_gemm_x3(FC_GEMM_TN, (const float*)dY, N, (const float*)X, K, dW, K, N, K, M, e, s);
      if (r <= 0) return r;
    }
    return fc_gemm_generic(dt, dt, FC_F32, dY, 1, N, X, K, 1, dW, K, N, K, M, e, s);
  }
  int attn_fwd(const void* qkv, void* o, float* lse, int B, int N) const {
    int H = m->cfg.heads, d = m->cfg.dim / H;
    float scale = 1.0f / sqrtf((float)d);
    if (dt == FC_BF16) {
      int r = fc_attn_fwd_mfma((const bf16_t*)qkv, (bf16_t*)o, lse, B, N, H, d, scale, s);
      if (r <= 0) return r;
    } else if (dt == FC_F32) {
      int r = fc_attn_f32_fwd((const float*)qkv, (float*)o, lse, B, N, H, d, scale, s);
      if (r <= 0) return r;
    }
    return fc_attn_fwd_generic(dt, qkv, o, lse, B, N, H, d, scale, s);
  }
  int attn_bwd(const void* qkv, const void* o, const void* dO, const float* lse, float* delta, void* dqkv, int B, int N) const {
    int H = m->cfg.heads, d = m->cfg.dim / H;
    float scale = 1.0f / sqrtf((float)d);
    if (dt == FC_BF16) {
      int r = fc_attn_bwd_mfma((const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dO, lse, delta, (bf16_t*)dqkv, B, N, H, d, scale, s);
      if (r <= 0) return r;
    } else if (dt == FC_F32) {
      int r = fc_attn_f32_bwd((const float*)qkv, (const float*)o, (const float*)dO, lse, delta, (float*)dqkv, B, N, H, d, scale, s);
      if (r <= 0) return r;
    }
    return fc_attn_bwd_generic(dt, qkv, o, dO, lse, delta, dqkv, B, N, H, d, scale, s);
  }
};

struct LnNote { fc_ln_part_t* partial; float* dg; float* db; int M, D; };
struct LayerRec { std::vector<LnNote> ln; bool ungraphable = false; };
void Ctx::mark_ungraphable() const { if (rec) rec->ungraphable = true; }
int Ctx::note_ln(fc_ln_part_t* partial, float* dg, float* db, int M, int D) const {
  if (rec) rec->ln.push_back(LnNote{partial, dg, db, M, D});
  for (FcLnReduce& e : *lnq)
    if (e.dg == dg) {
      if (!e.partial2) { e.partial2 = partial; e.nblocks2 = fc_layernorm_bwd_partial_blocks(M); return 0; }
      FC_REQUIRE(!e.partial3, "internal: more than three partial sets for one LayerNorm gradient");
      e.partial3 = partial; e.nblocks3 = fc_layernorm_bwd_partial_blocks(M);
      return 0;
    }
  lnq->push_back(FcLnReduce{partial, nullptr, dg, db, fc_layernorm_bwd_partial_blocks(M), 0, D, ln_accumulate, nullptr, 0, dt == FC_F32 ? 1 : 0});
  return 0;
}

static const float* dp_ptr(const fc_model* m, const Ws& w, int tower, int layer, int branch) {
  if (!w.droppath) return nullptr;
  return w.droppath + (((size_t)tower * m->cfg.depth + layer) * 2 + branch) * w.dp_stride + w.t[tower].dp_off;
}

// ---------------------------------------------------------------- forward (mome.py:881-922)
static int tower_embed_fwd(const Ctx& c, Ws& w, int i, const float* img, const int64_t* ids) {
  const fc_model* m = c.m;
  const fc_model_cfg& cf = m->cfg;
  const TowerP& tp = m->tw[i];
  TowerWs& t = w.t[i];
  const int D = cf.dim, N = t.N, B = t.M / N;
  const float* P = c.params;
  if (i == 0) {  // ImageEmbedding.forward mome.py:597-611
    int np = N - 1, kp = cf.in_chans * cf.patch * cf.patch;
    FC_TRY(fc_patchify(c.dt, img, t.patches, B, cf.in_chans, cf.img_size, cf.patch, c.s));
    GemmEpi e;
    e.bias = P + tp.pb;
    e.patch_rows = np;
    e.pos = P + tp.pos;
    FC_TRY(c.gemm_fwd(t.patches, c.W(tp.pw), t.x[0], B * np, D, kp, e));
    FC_TRY(fc_cls_rows(c.dt, P + tp.cls, P + tp.pos, t.x[0], B, N, D, c.s));
  } else {  // TextEmbedding.forward mome.py:632-639
    FC_REQUIRE(N <= cf.max_text_len, "text length %d exceeds max_text_len %d", N, cf.max_text_len);
    FC_TRY(fc_txt_embed_fwd(c.dt, ids, P + tp.word, P + tp.tpos, P + tp.ttype, P + tp.lnw, P + tp.lnb, t.x[0], t.emb_mean, t.emb_rstd, B, N, D,
                            cf.vocab, 1e-12f, c.s));
  }
  return 0;
}
static int tower_head_fwd(const Ctx& c, Ws& w, int i, int feat_out, float* out) {
  const fc_model* m = c.m;
  const fc_model_cfg& cf = m->cfg;
  const TowerP& tp = m->tw[i];
  TowerWs& t = w.t[i];
  const int D = cf.dim, N = t.N, B = t.M / N;
  const float* P = c.params;
  int normalize = feat_out || tp.task == FC_TASK_RTV;
  FC_TRY(fc_head_fwd(c.dt, t.x[cf.depth], P + m->normw, P + m->normb, t.f, t.hmean, t.hrstd, t.nrm, t.out, B, N, D, 1e-6f, normalize, c.s));
  if (normalize) {
    if (out) FC_CHECK_HIP(hipMemcpyAsync(out, t.out, sizeof(float) * (size_t)B * D, hipMemcpyDeviceToDevice, c.s));
  } else {
    FC_REQUIRE(tp.task == FC_TASK_CLS && tp.head_w >= 0, "tower %d has no head for feat_out=0", i);
    GemmEpi e;
    e.bias = P + tp.head_b;
    FC_TRY(fc_gemm_generic(FC_F32, FC_F32, FC_F32, t.f, D, 1, P + tp.head_w, 1, D, t.logits, tp.ncls, B, tp.ncls, D, e, c.s));
    if (out) FC_CHECK_HIP(hipMemcpyAsync(out, t.logits, sizeof(float) * (size_t)B * tp.ncls, hipMemcpyDeviceToDevice, c.s));
  }
  return 0;
}
// ---- chain schedule: the towers advance layer by layer in GROUPED launches on ONE stream (LayerNorm, the four linears and their dX
// products each take the image and the text rows in one launch: one read of a weight matrix per linear instead of three, 345 instead
// of 150 tiles for the N = 384 GEMMs, no hardware-queue lottery); the weight gradients still run as grouped chunks on their own stream.
struct LnFwdD { const void* x; const float* g; const float* b; void* y; float* mean; float* rstd; int M; };
static int ln_fwd_multi(const Ctx& c, const LnFwdD* d, int n, int D, float eps) {
  bool ok = fc_layernorm_grouped_ok(D) && n <= 2;
  for (int i = 0; i < n; ++i) ok = ok && !(((uintptr_t)d[i].x | (uintptr_t)d[i].y | (uintptr_t)d[i].g | (uintptr_t)d[i].b) & 15);
  if (ok && !FC_ABLATED("ln")) {
    FcLnFwdArgs a{};
    for (int i = 0; i < n; ++i) a.p[i] = FcLnFwdP{d[i].x, d[i].y, d[i].g, d[i].b, d[i].mean, d[i].rstd, d[i].M, 0};
    a.nprob = n; a.D = D; a.eps = eps;
    return fc_layernorm_fwd_grouped(c.dt, a, c.s);
  }
  for (int i = 0; i < n; ++i) FC_TRY(fc_layernorm_fwd(c.dt, d[i].x, d[i].g, d[i].b, d[i].y, d[i].mean, d[i].rstd, d[i].M, D, eps, c.s));
  return 0;
}
struct LnBwdD {
  const void* dy; const void* x; const float* mean; const float* rstd; const float* g; const void* res; void* dx; float* dg; float* db; int M;
  fc_ln_part_t* partial; void* dx_scaled; const float* rowscale; int rps;
};
static int ln_bwd_multi(const Ctx& c, const LnBwdD* d, int n, int D) {
  bool ok = fc_layernorm_grouped_ok(D) && n <= 2 && c.lnq;
  for (int i = 0; i < n; ++i)
    ok = ok && d[i].partial &&
         !(((uintptr_t)d[i].dy | (uintptr_t)d[i].x | (uintptr_t)d[i].res | (uintptr_t)d[i].dx | (uintptr_t)d[i].g | (uintptr_t)d[i].partial | (uintptr_t)d[i].dx_scaled) & 15);
  if (ok && !FC_ABLATED("ln")) {
    FcLnBwdArgs a{};
    for (int i = 0; i < n; ++i) {
      a.p[i] = FcLnBwdP{d[i].dy, d[i].x, d[i].mean, d[i].rstd, d[i].g, d[i].res, d[i].dx, d[i].dx_scaled, d[i].rowscale, d[i].partial, d[i].rps, d[i].M, 0, 0};
      FC_TRY(c.note_ln(d[i].partial, d[i].dg, d[i].db, d[i].M, D));
    }
    a.nprob = n; a.D = D;
    return fc_layernorm_bwd_grouped(c.dt, a, c.s);
  }
  for (int i = 0; i < n; ++i)
    FC_TRY(c.ln_bwd(d[i].dy, d[i].x, d[i].mean, d[i].rstd, d[i].g, d[i].res, d[i].dx, d[i].dg, d[i].db, d[i].M, D, d[i].partial, d[i].dx_scaled,
                    d[i].rowscale, d[i].rps));
  return 0;
}
// kind NT: C[M,N] = A[M,K] . W[N,K]^T (forward linear); kind NN: C[M,N] = A[M,K] . W[K,N] (dX = dY . W)
struct GemmD { const void* A; const void* W; void* C; int M; GemmEpi e; };
static int gemm_multi(const Ctx& c, int kind, const GemmD* d, int n, int N, int K) {
  if (c.dt == FC_BF16 && n >= 1 && n <= 2) {
    GemmGroup g{};
    for (int i = 0; i < n; ++i)
      g.p[i] = GemmProb{(const bf16_t*)d[i].A, (const bf16_t*)d[i].W, d[i].C, (long)K, kind == FC_GEMM_NT ? (long)K : (long)N, (long)N, d[i].M, d[i].e};
    g.N = N; g.K = K;
    int r = fc_gemm_mfma_grouped(kind, FC_BF16, g, n, c.s);
    if (r <= 0) return r;
  }
  for (int i = 0; i < n; ++i) {
    if (kind == FC_GEMM_NT) FC_TRY(c.gemm_fwd(d[i].A, d[i].W, d[i].C, d[i].M, N, K, d[i].e));
    else FC_TRY(c.gemm_dx(d[i].A, d[i].W, d[i].C, d[i].M, K, N, d[i].e));
  }
  return 0;
}

struct TowerList { int n = 0; int idx[2] = {0, 0}; };
static int chain_layer_forward(const Ctx& c, Ws& w, const TowerList& T, int l) {   // Block.forward mome.py:225-228, the listed towers per launch
  const fc_model* m = c.m;
  const fc_model_cfg& cf = m->cfg;
  const int D = cf.dim, Hd = cf.mlp_hidden, nt = T.n;
  const float* P = c.params;
  const int* tw = T.idx;
  LnFwdD ln[2];
  GemmD gd[2];
  for (int q = 0; q < nt; ++q) {
    const BlockP& b = m->tw[tw[q]].blocks[l]; TowerWs& t = w.t[tw[q]]; LayerWs& L = t.L[l];
    ln[q] = LnFwdD{t.x[l], P + b.n1w, P + b.n1b, L.h1, L.mean1, L.rstd1, t.M};
  }
  FC_TRY(ln_fwd_multi(c, ln, nt, D, 1e-5f));
  for (int q = 0; q < nt; ++q) {
    const BlockP& b = m->tw[tw[q]].blocks[l]; TowerWs& t = w.t[tw[q]]; LayerWs& L = t.L[l];
    gd[q] = GemmD{L.h1, c.W(b.qkv.w), L.qkv, t.M, GemmEpi()};
    gd[q].e.bias = P + b.qkv.b;
  }
  FC_TRY(gemm_multi(c, FC_GEMM_NT, gd, nt, 3 * D, D));
  for (int q = 0; q < nt; ++q) {
    TowerWs& t = w.t[tw[q]]; LayerWs& L = t.L[l];
    FC_TRY(c.attn_fwd(L.qkv, L.o, L.lse, t.M / t.N, t.N));
  }
  for (int q = 0; q < nt; ++q) {
    const BlockP& b = m->tw[tw[q]].blocks[l]; TowerWs& t = w.t[tw[q]]; LayerWs& L = t.L[l];
    gd[q] = GemmD{L.o, c.W(b.proj.w), L.xmid, t.M, GemmEpi()};
    gd[q].e.bias = P + b.proj.b; gd[q].e.res = t.x[l]; gd[q].e.rowscale = dp_ptr(m, w, tw[q], l, 0); gd[q].e.rows_per_sample = t.N;
  }
  FC_TRY(gemm_multi(c, FC_GEMM_NT, gd, nt, D, D));
  for (int q = 0; q < nt; ++q) {
    const BlockP& b = m->tw[tw[q]].blocks[l]; TowerWs& t = w.t[tw[q]]; LayerWs& L = t.L[l];
    ln[q] = LnFwdD{L.xmid, P + b.n2w, P + b.n2b, L.h2, L.mean2, L.rstd2, t.M};
  }
  FC_TRY(ln_fwd_multi(c, ln, nt, D, 1e-5f));
#ifdef FC_PROBES
  if (m->mlp_fused) {      // fc1 -> GELU -> fc2 + residual in one launch per tower (fc_mlp.hip)
    bool all = true;
    for (int q = 0; q < nt && all; ++q) {
      const BlockP& b = m->tw[tw[q]].blocks[l]; TowerWs& t = w.t[tw[q]]; LayerWs& L = t.L[l];
      const int r = fc_mlp_fused(0, L.h2, m->mlp_stream(c.wc, tw[q], l, 0), P + b.fc1.b, P + b.fc2.b, L.gact, L.u, L.xmid, dp_ptr(m, w, tw[q], l, 1), t.N,
                                 t.x[l + 1], t.M, D, Hd, c.s);
      if (r < 0) return r;
      FC_REQUIRE(r == 0 || q == 0, "internal: the fused MLP covered one tower of a layer and declined the other");
      all = r == 0;
    }
    if (all) return 0;
  }
#endif
  for (int q = 0; q < nt; ++q) {
    const BlockP& b = m->tw[tw[q]].blocks[l]; TowerWs& t = w.t[tw[q]]; LayerWs& L = t.L[l];
    gd[q] = GemmD{L.h2, c.W(b.fc1.w), L.gact, t.M, GemmEpi()};
    gd[q].e.bias = P + b.fc1.b; gd[q].e.preact = L.u; gd[q].e.gelu_saved_grad = (c.dt == FC_BF16);   // bf16: L.u holds gelu'(u)
  }
  FC_TRY(gemm_multi(c, FC_GEMM_NT, gd, nt, Hd, D));
  for (int q = 0; q < nt; ++q) {
    const BlockP& b = m->tw[tw[q]].blocks[l]; TowerWs& t = w.t[tw[q]]; LayerWs& L = t.L[l];
    gd[q] = GemmD{L.gact, c.W(b.fc2.w), t.x[l + 1], t.M, GemmEpi()};
    gd[q].e.bias = P + b.fc2.b; gd[q].e.res = L.xmid; gd[q].e.rowscale = dp_ptr(m, w, tw[q], l, 1); gd[q].e.rows_per_sample = t.N;
  }
  return gemm_multi(c, FC_GEMM_NT, gd, nt, D, Hd);
}
// Schedules (FC_SCHEDULE): "streams" (default) = two image micro-batch chains and the text tower on three streams + the weight-gradient
// stream; "chain" = ONE chain of grouped launches (every LayerNorm / GEMM launch takes all image and text rows); "chain2" = two chains of
// grouped launches (the first part of the image batch | the rest of it TOGETHER WITH the text tower).  Measured on the ViT-S step
// (round 3, one box, ms/step): streams 4.81, chain2 5.22, chain 5.31 -- DESIGN.md section 7.
enum { SCHED_CHAIN2 = 0, SCHED_CHAIN = 1, SCHED_STREAMS = 2 };
static int schedule() {
  static const int v = [] {
    const char* e = fc_knob_str("FC_SCHEDULE");
    if (e && strcmp(e, "chain2") == 0) return (int)SCHED_CHAIN2;
    if (e && strcmp(e, "chain") == 0) return (int)SCHED_CHAIN;
    return (int)SCHED_STREAMS;
  }();
  return v;
}
static bool chain_schedule() { return schedule() != SCHED_STREAMS; }

#ifdef FC_PROBES
// tools build (FC_STEP_PHASES=1): when each internal stream finished its part of the forward / backward, before the joins
static hipEvent_t g_stream_ev[64 * 8];
static bool g_stream_on = false;
static int g_stream_step = 0;
#define FC_STREAM_EV(i, st) do { if (g_stream_on) (void)hipEventRecord(g_stream_ev[(g_stream_step & 63) * 8 + (i)], (st)); } while (0)
#else
#define FC_STREAM_EV(i, st) do {} while (0)
#endif
// ---- immutable device tables (weight-gradient problem lists, LayerNorm reduction lists, optimizer chunk lists), cached per process by
// CONTENT.  A FedavgClient builds a new handle every round (download() = deepcopy), normally over the same device addresses: per-handle
// tables meant an allocation, an upload from pageable memory (which blocks the host until the stream gets there: 2.6 ms of the first
// step of every round) and, on replacement, a device synchronisation.  A table that is found here costs nothing; a new one costs one
// allocation + one synchronous copy, once per process.
// Lifetime rule: a pointer returned here is valid until the caller's NEXT cached_table() call at the latest (the cache may start over
// inside any call, behind a device synchronisation, so kernels already enqueued with an old table have finished before it is freed).
// Nobody keeps such a pointer across calls -- a handle that needs its table again asks again (a hit is one map lookup).
static std::mutex g_table_mu;
static int cached_table(const void* host, size_t bytes, const void** out) {
  static std::map<std::string, void*> cache;
  static const size_t limit = (size_t)fc_knob("FC_TABLE_CACHE_MAX", 1024);     // (tools build: a small limit forces the start-over path)
  std::lock_guard<std::mutex> lock(g_table_mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::string key((const char*)&dev, sizeof(dev));
  key.append((const char*)host, bytes);
  auto it = cache.find(key);
  if (it != cache.end()) { *out = it->second; return 0; }
  if (cache.size() >= limit) {                    // addresses kept changing: start over.  The tables of the generation BEFORE this one are
    static std::vector<void*> retired;            // freed now, behind a device synchronisation; this generation's stay allocated until the
    FC_CHECK_HIP(hipDeviceSynchronize());         // next start-over, so a thread that has just been handed one can still launch with it
    for (void* q : retired) (void)hipFree(q);
    retired.clear();
    for (auto& kv : cache) retired.push_back(kv.second);
    cache.clear();
  }
  void* d = nullptr;
  FC_CHECK_HIP(hipMalloc(&d, bytes));
  FC_CHECK_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
  cache[key] = d;
  *out = d;
  return 0;
}

// ---- per-layer HIP graphs.  The ViT-S step is ~600 kernel launches on four streams; an eager launch costs ~4.3 us of host time
// (tools/graph_launch_probe.hip: 4.0-4.7 us per kernel, 12 us per launch of a captured 7-kernel graph, which also dispatches at
// 2.1-2.7 us per dependent kernel on the GPU instead of 4), so the host needed 4.1 ms to enqueue a 4.6-ms step and FedavgClient's loop
// (loader + Python around it) was host-bound.  One layer of one chain -- 7 dependent kernels on one stream, no event inside -- is
// captured once into a graph (on a private stream, thread-local capture mode: nothing executes, other threads are unaffected) and
// replayed with one hipGraphLaunch.  Kernel arguments are baked in, so the key holds everything they derive from: model configuration,
// parameter / compute-weight / workspace / gradient addresses, batch cut, text length, drop-path on/off.  The cache is per process (a
// FedavgClient builds a new handle every round, usually at the same addresses).  First encounter of a key: eager (function attributes,
// lazy tables); second: capture; then replay.  Host-side notes of a layer (LayerNorm reductions to queue) are recorded with the graph.
struct GraphKey {
  uint64_t cfg;
  const void *ws, *wc, *params, *grads;
  int32_t dev, dir, tower, layer, b0, bn, B, n_txt, flags;
  bool operator<(const GraphKey& o) const { return memcmp(this, &o, sizeof(*this)) < 0; }
};
struct LayerGraph {
  hipGraphExec_t exec = nullptr;
  int seen = 0;
  bool eager_only = false;
  std::vector<LnNote> ln;
};
static std::map<GraphKey, LayerGraph>& graph_cache() { static std::map<GraphKey, LayerGraph> c; return c; }
static bool graphs_enabled() { static const bool v = fc_knob("FC_GRAPHS", 0) != 0; return v; }
static uint64_t cfg_hash(const fc_model* m) {
  uint64_t h = 1469598103934665603ull;
  const unsigned char* b = (const unsigned char*)&m->cfg;
  for (size_t i = 0; i < sizeof(m->cfg); ++i) h = (h ^ b[i]) * 1099511628211ull;
  return (h ^ (uint64_t)m->total) * 1099511628211ull;
}
static GraphKey graph_key(const Ctx& c, const Ws& w, int dir, int tower, int layer, const float* grads) {
  GraphKey k;
  memset(&k, 0, sizeof(k));
  k.cfg = cfg_hash(c.m);
  k.ws = w.t[tower].x.empty() ? nullptr : w.t[tower].x[0];      // the slice's first row: fixes workspace base and batch cut
  k.wc = c.wc; k.params = c.params; k.grads = grads;
  (void)hipGetDevice(&k.dev);
  k.dir = dir; k.tower = tower; k.layer = layer; k.b0 = w.t[tower].dp_off; k.bn = w.t[tower].M; k.B = w.dp_stride; k.n_txt = w.n_txt;
  k.flags = (w.droppath ? 1 : 0) | (c.no_wgrad ? 2 : 0) | (c.ln_accumulate ? 4 : 0) | (c.lnq ? 8 : 0);
  return k;
}
// body(ctx) enqueues the layer on ctx.s
template <typename F>
static int run_layer(const Ctx& c, const GraphKey& key, F body) {
  const fc_model* m = c.m;
  // drop-path tables are the caller's (a new address every step): such steps run eagerly
  if (!graphs_enabled() || c.dt != FC_BF16 || (key.flags & 1) || FC_ABLATED("graph")) return body(c);
  std::map<GraphKey, LayerGraph>& cache = graph_cache();
  if (cache.size() > 4096) {                      // addresses kept changing: start over
    for (auto& kv : cache)
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    cache.clear();
  }
  LayerGraph& g = cache[key];
  if (g.exec) {
    FC_CHECK_HIP(hipGraphLaunch(g.exec, c.s));
    for (const LnNote& n : g.ln) FC_TRY(c.note_ln(n.partial, n.dg, n.db, n.M, n.D));
    return 0;
  }
  if (g.eager_only || g.seen++ == 0) return body(c);
  if (!m->cap) FC_CHECK_HIP(hipStreamCreateWithFlags(&m->cap, hipStreamNonBlocking));
  LayerRec rec;
  Ctx cc = c;
  cc.s = m->cap;
  cc.rec = &rec;
  std::vector<FcLnReduce> ln_saved;
  if (c.lnq) ln_saved = *c.lnq;
  if (hipStreamBeginCapture(m->cap, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    g.eager_only = true;
    return body(c);
  }
  const int r = body(cc);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(m->cap, &graph);
  bool ok = r == 0 && e == hipSuccess && graph && !rec.ungraphable;
  if (ok && hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) { g.exec = nullptr; ok = false; }
  if (graph) (void)hipGraphDestroy(graph);
  if (!ok) {
    (void)hipGetLastError();
    g.eager_only = true;
    if (c.lnq) *c.lnq = ln_saved;                 // the capture's notes are repeated by the eager run
    if (r != 0) return r;
    return body(c);
  }
  g.ln = rec.ln;                                  // (already applied once, by the capture)
  FC_CHECK_HIP(hipGraphLaunch(g.exec, c.s));
  return 0;
}

struct ChainPlan {           // how the batch is cut into chains: chain 0 = image samples [0, b0) on the caller's stream; chain 1 = image
  int nchains = 1, b0 = 0;   // samples [b0, B) and the whole text tower on the second stream (nchains == 1: everything on the caller's)
};
static int ensure_side(const fc_model* m, hipStream_t caller);
static ChainPlan chain_plan(const fc_model* m, int B, bool run_img, bool run_txt) {
  ChainPlan p;
  static const int first = fc_knob("FC_MB_FIRST", 0);
  if (schedule() != SCHED_CHAIN2 || !run_img || m->dt != FC_BF16 || B < 16 || !m->mbs[0]) return p;
  // equal rows per chain with the text tower (B x n_txt rows) on the second one; image only: 57 : 43 (equal chains run in lockstep and
  // collide phase by phase, round 2)
  int pct = first > 0 && first < 100 ? first : 57;
  p.b0 = (int)((long)B * pct / 100);
  if (run_txt && !(first > 0 && first < 100)) {
    const int n_img = (m->cfg.img_size / m->cfg.patch) * (m->cfg.img_size / m->cfg.patch) + 1;
    const long n_txt = m->last_ntxt > 0 ? m->last_ntxt : 32;
    p.b0 = (int)(((long)B * (n_img + n_txt) + n_img) / (2L * n_img));
  }
  if (p.b0 < 1 || p.b0 > B - 1) return p;
  p.nchains = 2;
  return p;
}
static int chains_forward(const Ctx& c, Ws& w, const float* img, const int64_t* ids, int feat_out, float* out_img, float* out_txt) {
  const fc_model* m = c.m;
  const fc_model_cfg& cf = m->cfg;
  const bool has_img = m->tw[0].present, has_txt = m->tw[1].present && !FC_ABLATED("txt");
  FC_TRY(ensure_side(m, c.s));
  m->last_ntxt = w.n_txt;
  const ChainPlan pl = chain_plan(m, w.B, has_img, has_txt);
  if (pl.nchains == 1) {
    TowerList T;
    if (has_img) T.idx[T.n++] = 0;
    if (has_txt) T.idx[T.n++] = 1;
    for (int q = 0; q < T.n; ++q) FC_TRY(tower_embed_fwd(c, w, T.idx[q], img, ids));
    for (int l = 0; l < cf.depth; ++l) FC_TRY(chain_layer_forward(c, w, T, l));
    for (int q = 0; q < T.n; ++q) FC_TRY(tower_head_fwd(c, w, T.idx[q], feat_out, T.idx[q] == 0 ? out_img : out_txt));
    return 0;
  }
  const int b0 = pl.b0, B = w.B;
  const size_t ipx = (size_t)cf.in_chans * cf.img_size * cf.img_size;
  const size_t ow = (size_t)((feat_out || m->tw[0].task == FC_TASK_RTV) ? cf.dim : m->tw[0].ncls);
  Ws wa = slice_ws(m, w, 0, 0, b0, 0, false), wb = slice_ws(m, w, 0, b0, B - b0, 1, false);
  Ctx ca = c, cb = c;
  cb.s = m->mbs[0];
  TowerList TA, TB;
  TA.idx[TA.n++] = 0;
  TB.idx[TB.n++] = 0;
  if (has_txt) TB.idx[TB.n++] = 1;
  FC_CHECK_HIP(hipEventRecord(m->ev_fork, c.s));
  FC_CHECK_HIP(hipStreamWaitEvent(cb.s, m->ev_fork, 0));
  FC_TRY(tower_embed_fwd(ca, wa, 0, img, nullptr));
  FC_TRY(tower_embed_fwd(cb, wb, 0, img + (size_t)b0 * ipx, nullptr));
  if (has_txt) FC_TRY(tower_embed_fwd(cb, wb, 1, nullptr, ids));
  for (int l = 0; l < cf.depth; ++l) {
    FC_TRY(chain_layer_forward(ca, wa, TA, l));
    FC_TRY(chain_layer_forward(cb, wb, TB, l));
  }
  FC_TRY(tower_head_fwd(ca, wa, 0, feat_out, out_img));
  FC_TRY(tower_head_fwd(cb, wb, 0, feat_out, out_img ? out_img + (size_t)b0 * ow : nullptr));
  if (has_txt) FC_TRY(tower_head_fwd(cb, wb, 1, feat_out, out_txt));
  FC_STREAM_EV(1, cb.s); FC_STREAM_EV(2, c.s);
  FC_CHECK_HIP(hipEventRecord(m->ev_mb_join[0], cb.s));
  FC_CHECK_HIP(hipStreamWaitEvent(c.s, m->ev_mb_join[0], 0));
  return 0;
}

// ---- the library's internal streams: ONE set per device for the whole process, chosen by measurement.
// HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order; two streams that share a
// queue serialise.  Which queue a new stream gets depends on every stream the process created before (torch's pool, other
// handles), and a collision costs a third of the throughput (ViT-S step 5.5 ms -> 7.2 ms).  So candidates are created and TESTED:
// a stream is accepted only if a spin kernel on it overlaps with spin kernels on the caller's stream and on every stream
// accepted so far.  Streams beyond the distinct queues available are accepted untested.
__global__ void k_spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
static bool streams_overlap(hipStream_t a, hipStream_t b) {
  hipEvent_t e0, e1, eb;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess)
    return true;
  const long long ticks = 8000;      // 80 us at the 100 MHz wall clock
  (void)hipStreamSynchronize(a);
  (void)hipStreamSynchronize(b);
  (void)hipEventRecord(e0, a);
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks);
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks);
  (void)hipEventRecord(eb, b);
  (void)hipStreamWaitEvent(a, eb, 0);
  (void)hipEventRecord(e1, a);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(eb);
  return ms < 0.130f;               // concurrent: ~0.08 ms, serialised: ~0.16 ms
}
struct StreamSet { hipStream_t s[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; int n = 0; };
static StreamSet& device_streams(hipStream_t caller, int want) {
  static StreamSet sets[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  StreamSet& S = sets[dev & 15];
  static const bool calibrate = fc_knob("FC_STREAM_CALIBRATE", 1) != 0;
  int tested_ok = 0;
  while (S.n < want) {
    hipStream_t pick = nullptr;
    std::vector<hipStream_t> rejected;
    for (int attempt = 0; attempt < 8 && !pick; ++attempt) {
      hipStream_t c = nullptr;
      if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
      bool ok = true;
      static const int hwq = getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4;
      if (calibrate && S.n < hwq - 1) {    // caller + (queues - 1) streams can be distinct; later ones cannot
        ok = streams_overlap(caller, c);
        for (int i = 0; ok && i < S.n; ++i) ok = streams_overlap(S.s[i], c);
      }
      if (ok) pick = c; else rejected.push_back(c);
    }
    if (!pick) {                            // no collision-free queue found: take a fresh stream as it comes
      if (hipStreamCreateWithFlags(&pick, hipStreamNonBlocking) != hipSuccess) break;
    } else {
      ++tested_ok;
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    S.s[S.n++] = pick;
  }
  (void)tested_ok;
  return S;
}
static int ensure_side(const fc_model* m, hipStream_t caller) {
  if (!m->dws) {
    if (chain_schedule()) {     // weight gradients, the second chain, the batch prefetcher's copy stream (fc_model_side_stream): with the
      StreamSet& S = device_streams(caller, 3);   // caller's, the four hardware queues HIP drives
      FC_REQUIRE(S.n >= 3, "could not create the internal HIP streams");
      m->dws = S.s[0];
      m->mbs[0] = schedule() == SCHED_CHAIN2 ? S.s[1] : nullptr;
      m->side = S.s[2];
    } else {
    static int req = fc_knob("FC_MICROBATCH", 2);
    const int nmb = (req > 2 ? 2 : (req < 1 ? 1 : req)) - 1;          // extra image chains
    StreamSet& S = device_streams(caller, 2 + (nmb > 1 ? nmb : 1));
    FC_REQUIRE(S.n >= 3, "could not create the internal HIP streams");
    // order of preference for collision-free queues: weight gradients, first extra image chain, text tower
    m->dws = S.s[0];
    m->mbs[0] = nmb >= 1 ? S.s[1] : nullptr;
    m->side = S.s[2];
    for (int k = 1; k < 3; ++k) m->mbs[k] = (k < nmb && 2 + k < S.n) ? S.s[2 + k] : nullptr;
    }
    FC_CHECK_HIP(hipEventCreateWithFlags(&m->ev_dw_in, hipEventDisableTiming));
    for (int k = 0; k < 3; ++k) {
      FC_CHECK_HIP(hipEventCreateWithFlags(&m->ev_dw_in2[k], hipEventDisableTiming));
      FC_CHECK_HIP(hipEventCreateWithFlags(&m->ev_mb_join[k], hipEventDisableTiming));
    }
    FC_CHECK_HIP(hipEventCreateWithFlags(&m->ev_dw_out, hipEventDisableTiming));
    FC_CHECK_HIP(hipEventCreateWithFlags(&m->ev_dw_prev, hipEventDisableTiming));
    FC_CHECK_HIP(hipEventCreateWithFlags(&m->ev_dw_last, hipEventDisableTiming));
    FC_CHECK_HIP(hipEventCreateWithFlags(&m->ev_zero, hipEventDisableTiming));
    FC_CHECK_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    FC_CHECK_HIP(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
  }
  return 0;
}
extern "C" void* fc_model_side_stream(const fc_model_t* m) {
  if (!m || ensure_side(m, nullptr) != 0) return nullptr;
  return (void*)m->side;
}
static int fork_side(const fc_model* m, hipStream_t s) {   // side (text tower) and micro-batch streams start after `s`
  FC_TRY(ensure_side(m, s));
  FC_CHECK_HIP(hipEventRecord(m->ev_fork, s));
  FC_CHECK_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
  for (int k = 0; k < 3 && m->mbs[k]; ++k) FC_CHECK_HIP(hipStreamWaitEvent(m->mbs[k], m->ev_fork, 0));
  return 0;
}
static int join_side(const fc_model* m, hipStream_t s) {
  FC_CHECK_HIP(hipEventRecord(m->ev_join, m->side));
  FC_CHECK_HIP(hipStreamWaitEvent(s, m->ev_join, 0));
  for (int k = 0; k < 3 && m->mbs[k]; ++k) {
    FC_CHECK_HIP(hipEventRecord(m->ev_mb_join[k], m->mbs[k]));
    FC_CHECK_HIP(hipStreamWaitEvent(s, m->ev_mb_join[k], 0));
  }
  return 0;
}

static int check_ws(const fc_model* m, int B, int n_txt, void* workspace, size_t bytes, Ws& w) {
  FC_REQUIRE(B > 0, "batch must be positive");
  FC_REQUIRE(workspace != nullptr, "workspace is null");
  carve(m, B, n_txt, workspace, w);
  FC_REQUIRE(w.bytes <= bytes, "workspace too small: need %zu bytes, got %zu", w.bytes, bytes);
  return 0;
}

// the internal streams / events / tables belong to the CURRENT device: it must be the one that holds the caller's buffers
static int check_device(const void* p, const char* what) {
  hipPointerAttribute_t at;
  int cur = -1;
  if (p && hipPointerGetAttributes(&at, p) == hipSuccess && hipGetDevice(&cur) == hipSuccess)
    FC_REQUIRE(at.device == cur, "%s: the buffers live on device %d but the current device is %d (hipSetDevice / torch.cuda.device first)", what,
               at.device, cur);
  else
    (void)hipGetLastError();
  return 0;
}
static int ensure_tables(const fc_model* m, const Ws& w, hipStream_t s, FcTnProblem** probs, FcLnReduce** lntab) {
  (void)m; (void)w; (void)s;        // the device tables come from cached_table() at the point of use
  *probs = nullptr;
  *lntab = nullptr;
  return 0;
}

// ---- "streams" schedule (default): every tower, and every micro-batch slice of the image tower, is a CHAIN on its own stream; the
// chains advance layer by layer (host enqueue order: layer-major, so no chain waits for another one's launches), each layer of each
// chain being one replayed graph (run_layer).  Forward: three image slices (the weight-gradient stream is idle then) + the text tower;
// backward: three image slices (the third on the text tower's stream) + the text tower on the weight-gradient stream, the weight gradients
// of both towers queued per layer over the FULL batch by the driver and flushed every few layers to the weight-gradient stream behind
// the chains that produced them; the last image chunk goes to a chain's stream that has fallen idle by then.
struct ChainDef { hipStream_t s; Ws w; int tower; int b0; hipEvent_t join; };
static int build_chains(const fc_model* m, const Ws& w, hipStream_t s, bool fwd, bool run0, bool run1, bool deferred, ChainDef* ch) {
  int n = 0;
  const int B = w.B;
  if (run0) {
    int nimg = (m->dt == FC_BF16 && (fwd || deferred)) ? microbatches(m, B) : 1;
    static const int fwd_chains = fc_knob("FC_FWD_CHAINS", 3);
    static const int bwd_chains = fc_knob("FC_BWD_CHAINS", 3);
    // 4.79 -> 4.71 ms per ViT-S step (round 3); the 768-wide model is faster with two forward chains (12.63 vs 12.84 ms)
    if (fwd && nimg == 2 && fwd_chains == 3 && B >= 24 && m->dws && m->cfg.dim <= 512) nimg = 3;
    // Three image chains in the backward as well (thirds of the batch; FC_BWD_CHAINS=2, tools build: two chains cut 57 : 43): the third
    // on the text tower's stream, the text tower on the weight-gradient stream, chunk by chunk between the image chunks (backward_impl).
    // 4.51 -> 4.44 ms per ViT-S img+txt step on one box, 4.58 -> 4.49 on another (profiles/r03/bwd3.txt).  Only beside a text tower and
    // for narrow models: image-only ViT-S clients (4.41 -> 4.45 ms), ViT-Tiny (2.69 -> 2.81) and the 768-wide img+txt model (12.54 ->
    // 12.68) are faster with two backward chains.
    const bool bwd3 = !fwd && nimg == 2 && bwd_chains == 3 && B >= 24 && m->dws && m->side && deferred && run1 && m->cfg.dim <= 512;
    if (bwd3) nimg = 3;
    hipStream_t st[3] = {s, m->mbs[0], bwd3 ? m->side : m->dws};
    hipEvent_t ev[3] = {nullptr, m->ev_mb_join[0], bwd3 ? m->ev_join : m->ev_dw_prev};
    // cut points of the three forward chains in percent of the batch (FC_FWD_CUTS = "a,b", tools build; default thirds)
    static const int fcut_a = [] { const char* e = fc_knob_str("FC_FWD_CUTS"); return e ? atoi(e) : 0; }();
    static const int fcut_b = [] { const char* e = fc_knob_str("FC_FWD_CUTS"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 0; }();
    static const int bcut_a = [] { const char* e = fc_knob_str("FC_BWD_CUTS"); return e ? atoi(e) : 0; }();
    static const int bcut_b = [] { const char* e = fc_knob_str("FC_BWD_CUTS"); const char* c = e ? strchr(e, ',') : nullptr; return c ? atoi(c + 1) : 0; }();
    const int cut_a = fwd ? fcut_a : bcut_a, cut_b = fwd ? fcut_b : bcut_b;
    auto cut3 = [&](int k) { return k <= 0 ? 0 : k >= 3 ? B : (cut_a > 0 && cut_b > cut_a && cut_b < 100) ? std::max(k, std::min(B - 3 + k, (B * (k == 1 ? cut_a : cut_b) + 50) / 100)) : B * k / 3; };
    for (int k = 0; k < nimg; ++k) {
      const int b0 = nimg == 3 ? cut3(k) : mb_begin(B, k, nimg), b1 = nimg == 3 ? cut3(k + 1) : mb_begin(B, k + 1, nimg);
      ch[n++] = ChainDef{st[k], nimg == 1 ? w : slice_ws(m, w, 0, b0, b1 - b0, fwd ? (k < 2 ? k : 1) : k, false), 0, b0, ev[k]};
    }
    if (run1 && bwd3) {                        // the text tower on the weight-gradient stream
      ch[n++] = ChainDef{m->dws, w, 1, 0, m->ev_mb_join[2]};
      return n;
    }
  }
  if (run1) {
    const bool own = run0 && m->side;          // beside an image tower: its own stream
    ch[n++] = ChainDef{own ? m->side : s, w, 1, 0, own ? m->ev_join : nullptr};
  }
  return n;
}
static int chains_fork(const fc_model* m, hipStream_t s, const ChainDef* ch, int n) {
  bool any = false;
  for (int k = 0; k < n; ++k) any = any || ch[k].s != s;
  if (!any) return 0;
  FC_CHECK_HIP(hipEventRecord(m->ev_fork, s));
  for (int k = 0; k < n; ++k)
    if (ch[k].s != s) FC_CHECK_HIP(hipStreamWaitEvent(ch[k].s, m->ev_fork, 0));
  return 0;
}
static int chains_join(hipStream_t s, const ChainDef* ch, int n) {
  for (int k = 0; k < n; ++k)
    if (ch[k].s != s) {
      FC_CHECK_HIP(hipEventRecord(ch[k].join, ch[k].s));
      FC_CHECK_HIP(hipStreamWaitEvent(s, ch[k].join, 0));
    }
  return 0;
}
static int streams_forward(const Ctx& c, Ws& w, const float* img, const int64_t* ids, int feat_out, float* out_img, float* out_txt) {
  const fc_model* m = c.m;
  const fc_model_cfg& cf = m->cfg;
  const bool run0 = m->tw[0].present, run1 = m->tw[1].present && !FC_ABLATED("txt");
  if (c.dt == FC_BF16 || (run0 && run1)) FC_TRY(ensure_side(m, c.s));
  ChainDef ch[4];
  const int n = build_chains(m, w, c.s, true, run0, run1, true, ch);
  FC_TRY(chains_fork(m, c.s, ch, n));
  const size_t ipx = (size_t)cf.in_chans * cf.img_size * cf.img_size;
  const size_t ow = (size_t)((feat_out || m->tw[0].task == FC_TASK_RTV) ? cf.dim : m->tw[0].ncls);
  Ctx cx[4];
  for (int k = 0; k < n; ++k) {
    cx[k] = c;
    cx[k].s = ch[k].s;
    FC_TRY(tower_embed_fwd(cx[k], ch[k].w, ch[k].tower, ch[k].tower == 0 ? img + (size_t)ch[k].b0 * ipx : nullptr, ids));
  }
  auto layer = [&](int k, int l) -> int {
    TowerList T;
    T.idx[T.n++] = ch[k].tower;
    Ws& wk = ch[k].w;
    return run_layer(cx[k], graph_key(cx[k], wk, 0, ch[k].tower, l, nullptr), [&](const Ctx& q) { return chain_layer_forward(q, wk, T, l); });
  };
  // host enqueue order: chain by chain, the text tower first (FC_FWD_ORDER=1, tools build: layer by layer across the chains)
  static const bool layer_major = fc_knob("FC_FWD_ORDER", 0) != 0;
  if (layer_major) {
    for (int l = 0; l < cf.depth; ++l)
      for (int k = 0; k < n; ++k) FC_TRY(layer(k, l));
  } else {
    for (int q = 0; q < n; ++q) {
      const int k = (q + n - 1) % n;                   // the last chain (text, when there is one) first
      for (int l = 0; l < cf.depth; ++l) FC_TRY(layer(k, l));
    }
  }
  for (int k = 0; k < n; ++k) {
    float* out = ch[k].tower == 0 ? (out_img ? out_img + (size_t)ch[k].b0 * ow : nullptr) : out_txt;
    FC_TRY(tower_head_fwd(cx[k], ch[k].w, ch[k].tower, feat_out, out));
  }
  FC_STREAM_EV(0, m->side); FC_STREAM_EV(1, m->mbs[0]); FC_STREAM_EV(2, c.s);
  return chains_join(c.s, ch, n);
}

static int forward_impl(const fc_model* m, const float* params, const void* wc, const float* img, const int64_t* ids, int B, int n_txt,
                        int feat_out, const float* droppath, void* workspace, size_t wbytes, float* out_img, float* out_txt, hipStream_t s, Ws& w) {
  FC_REQUIRE(params, "params is null");
  FC_TRY(check_device(params, "fc_forward"));
  if (m->need_wc) FC_REQUIRE(wc && wc != (const void*)params, "this configuration needs a compute-weight buffer (fc_prepare_weights)");
  // 'None modality should have None input.' (mome.py:890)
  FC_REQUIRE(m->tw[0].present == (img != nullptr), "None modality should have None input. (img)");
  FC_REQUIRE(m->tw[1].present == (ids != nullptr), "None modality should have None input. (txt)");
  FC_TRY(check_ws(m, B, m->tw[1].present ? n_txt : 0, workspace, wbytes, w));
  w.feat_out = feat_out; w.droppath = droppath; w.ids = ids;
  Ctx c{m, params, m->need_wc ? (const char*)wc : (const char*)params, s, m->dt, fc_esize(m->dt)};
  if (chain_schedule()) return chains_forward(c, w, img, ids, feat_out, out_img, out_txt);
  return streams_forward(c, w, img, ids, feat_out, out_img, out_txt);
}


extern "C" int fc_forward(const fc_model_t* m, const float* params, const void* wc, const float* img, const int64_t* ids, int32_t B,
                          int32_t n_txt, int32_t feat_out, const float* droppath, void* workspace, size_t workspace_bytes, float* out_img,
                          float* out_txt, void* stream) {
  Ws w;
  FC_TRY(forward_impl(m, params, wc, img, ids, B, n_txt, feat_out, droppath, workspace, workspace_bytes, out_img, out_txt, (hipStream_t)stream, w));
  m->last = LastFwd{workspace, B, w.n_txt, feat_out, droppath, ids};
  return 0;
}

// ---------------------------------------------------------------- backward
static int dw_flush_every() {
  // round-2 sweeps (ms/step, one box each).  128x128 dW tiles: 1: 6.2, 2: 5.32, 3: 5.28, 4: 5.19, 6: 5.25.  128x384 tiles (whole CUs): 3: 5.32,
  // 4: 5.00, 5: 5.11, 6: 4.93, 8: 5.18, 12 (no overlap with the backward at all): 4.97 -- overlapping the weight gradients with
  // the backward buys 1 %: the backward is throughput-bound, what runs beside it slows it by about what it saves
  static int v = fc_knob("FC_DW_FLUSH", 6);
  return v > 0 ? v : 6;
}
// flush after layer l?  (phase 1 would make the last, un-overlapped chunk the smallest -- layer 0 + embedding -- but
// measured 2 % slower than phase 0 on the ViT-S step)
static bool dw_flush_here(int l) {
  // FC_DW_FLUSH_AT="6,1": explicit list of layers after which the queued weight gradients are launched (experiments)
  static const char* at = fc_knob_str("FC_DW_FLUSH_AT");
  if (at) {
    for (const char* p = at; *p;) {
      if (atoi(p) == l && l > 0) return true;
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
    return false;
  }
  static int ph = fc_knob("FC_DW_PHASE", 0);
  const int e = dw_flush_every();
  return l > 0 && (l % e) == (ph % e);
}
struct DwState {
  size_t flushed = 0;     // problems [0, flushed) have been launched
  int tiles = 0;
  bool changed = false;   // some chunk differed from the cached table
  FcTnProblem* dev = nullptr;
  int max_probs = 0;
};
// launch the problems queued since the last flush as one grouped GEMM on the dW stream, ordered after everything enqueued so
// far on this tower's stream (their dY / X operands are complete by then)
static int flush_dw(const Ctx& c, bool narrow = false, hipStream_t target = nullptr) {   // narrow: 128x128 tiles for every problem; target: default the dW stream
  if (!c.defer || !c.dw) return 0;
  std::vector<FcTnProblem>& all = *c.defer;
  DwState& st = *c.dw;
  const size_t beg = st.flushed, n = all.size() - beg;
  if (n == 0) return 0;
  FC_REQUIRE((int)all.size() <= st.max_probs, "internal: too many deferred weight-gradient problems");
  // problems whose `in` is a multiple of 384 take the 128x384-tile kernel, the rest the 128x128 one: two launches over two
  // contiguous parts of the table, each with its own tile numbering
  const size_t nw = narrow ? 0 : (size_t)(std::stable_partition(all.begin() + beg, all.end(), [](const FcTnProblem& p) { return fc_gemm_dw_wide_supported(p) != 0; }) -
                                          (all.begin() + beg));
  int tiles_w = 0, tiles = 0;
  for (size_t i = beg; i < beg + nw; ++i) {
    all[i].tile_start = tiles_w;
    tiles_w += fc_gemm_dw_wide_tiles(all[i], &all[i].tiles_n);
  }
  for (size_t i = beg + nw; i < all.size(); ++i) {
    all[i].tile_start = tiles;
    all[i].tiles_n = fc_cdiv(all[i].N, 128);
    tiles += fc_cdiv(all[i].M, 128) * all[i].tiles_n;
  }
  const fc_model* m = c.m;
  const void* tab = nullptr;
  FC_TRY(cached_table(all.data() + beg, n * sizeof(FcTnProblem), &tab));
  const FcTnProblem* chunk = (const FcTnProblem*)tab;
  const hipStream_t tgt = target ? target : m->dws;
  if (c.s != tgt) {
    FC_CHECK_HIP(hipEventRecord(m->ev_dw_in, c.s));
    FC_CHECK_HIP(hipStreamWaitEvent(tgt, m->ev_dw_in, 0));
  }
  for (int k = 0; k < c.n_more; ++k) {
    const hipStream_t ms = c.more_s ? c.more_s[k] : m->mbs[k];
    if (ms == tgt) continue;
    FC_CHECK_HIP(hipEventRecord(m->ev_dw_in2[k], ms));
    FC_CHECK_HIP(hipStreamWaitEvent(tgt, m->ev_dw_in2[k], 0));
  }
  if (!FC_ABLATED("dw")) {
    FC_TRY(fc_gemm_dw_wide(chunk, (int)nw, tiles_w, tgt, c.fopt));
    FC_TRY(fc_gemm_tn_grouped(chunk + nw, (int)(n - nw), tiles, tgt, c.fopt));
  }
  st.flushed = all.size();
  return 0;
}

static int weight_grad(const Ctx& c, const void* dY, const void* X, int M, int out, int in, float* dW, float* db) {
  if (c.no_wgrad) return 0;
  if (c.defer) {
    FcTnProblem p{(const bf16_t*)dY, (const bf16_t*)X, dW, db, out, in, in, out, in, M, 0, 0};
    if (fc_gemm_tn_grouped_supported(p)) {
      c.defer->push_back(p);
      if (c.fused_seg)
        for (size_t k = 0; k < c.m->segs.size(); ++k)
          if (c.m->segs[k].offset == dW - c.fopt->g0 || (db && c.m->segs[k].offset == db - c.fopt->g0)) (*c.fused_seg)[k] = 1;
      return 0;
    }
  }
  c.mark_ungraphable();      // launches that accumulate into the gradient buffer: keep such layers out of graphs
  if (c.dt == FC_F32) {      // fp32 mode: split-operand MFMA products, the reduction over the rows cut into slices, db from the same pass
    const int r = fc_dw_x3((const float*)dY, (const float*)X, dW, db, M, out, in, c.s);
    if (r <= 0) return r;
  }
  FC_TRY(c.gemm_dw(dY, X, dW, M, out, in));
  FC_TRY(fc_colsum(c.dt, dY, db, M, out, 1, c.s));
  return 0;
}
static int linear_bwd_params(const Ctx& c, const LinearP& L, const void* dY, const void* X, int M, float* grads) {
  float* g = L.shared ? c.gshared : grads;     // a tower that borrows the linear accumulates beside the owner, summed afterwards
  return weight_grad(c, dY, X, M, L.out, L.in, g + L.w, g + L.b);
}

enum { PH_ALL = -1, PH_HEAD = 0, PH_LAYER = 1, PH_EMBED = 2, PH_WGRAD_LAYER = 3, PH_WGRAD_EMBED = 4 };
// phase PH_ALL runs the whole tower; the micro-batched driver calls it phase by phase (PH_LAYER: layer `lsel` only).
// PH_WGRAD_*: only queue the weight-gradient problems of layer `lsel` / of the patch embedding (full-batch view).
static int tower_backward(const Ctx& c, Ws& w, int i, const float* d_out, float* grads, int phase = PH_ALL, int lsel = -1) {
  const fc_model* m = c.m;
  const fc_model_cfg& cf = m->cfg;
  const TowerP& tp = m->tw[i];
  TowerWs& t = w.t[i];
  const int D = cf.dim, N = t.N, M = t.M, B = M / N, Hd = cf.mlp_hidden;
  const float* P = c.params;
  int normalize = w.feat_out || tp.task == FC_TASK_RTV;
  const float* din = d_out;
  const bool wg_only = (phase == PH_WGRAD_LAYER || phase == PH_WGRAD_EMBED);
  if (phase == PH_ALL || phase == PH_HEAD) {
  if (!normalize) {  // ClassificationHead backward (mome.py:647-649)
    GemmEpi e;
    FC_TRY(fc_gemm_generic(FC_F32, FC_F32, FC_F32, d_out, tp.ncls, 1, P + tp.head_w, D, 1, t.df, D, B, D, tp.ncls, e, c.s));      // df = dlogits . Wh
    GemmEpi ea; ea.accumulate = 1;
    FC_TRY(fc_gemm_generic(FC_F32, FC_F32, FC_F32, d_out, 1, tp.ncls, t.f, D, 1, grads + tp.head_w, D, tp.ncls, D, B, ea, c.s));   // dWh = dlogits^T f
    FC_TRY(fc_colsum(FC_F32, d_out, grads + tp.head_b, B, tp.ncls, 1, c.s));
    din = t.df;
  }
  FC_TRY(fc_head_bwd(c.dt, din, t.out, t.nrm, normalize, t.x[cf.depth], t.hmean, t.hrstd, P + m->normw, t.gx[cf.depth], grads + m->normw,
                     grads + m->normb, B, N, D, c.s));
  }
  for (int l = cf.depth - 1; l >= 0; --l) {
    if (!(phase == PH_ALL || ((phase == PH_LAYER || phase == PH_WGRAD_LAYER) && l == lsel))) continue;
    const BlockP& b = tp.blocks[l];
    LayerWs& L = t.L[l];
    const void* dx = t.gx[l + 1];
    if (wg_only) {   // same operand choice as below, no kernels: queue dW / db of this layer over the full batch
      const void* dm = dp_ptr(m, w, i, l, 1) ? (const void*)L.gdm : dx;
      const void* da = dp_ptr(m, w, i, l, 0) ? (const void*)L.gda : (const void*)L.gxmid;
      FC_TRY(linear_bwd_params(c, b.fc2, dm, L.gact, M, grads));
      FC_TRY(linear_bwd_params(c, b.fc1, L.gdu, L.h2, M, grads));
      FC_TRY(linear_bwd_params(c, b.proj, da, L.o, M, grads));
      FC_TRY(linear_bwd_params(c, b.qkv, L.gdqkv, L.h1, M, grads));
      continue;
    }
    FC_REQUIRE(false, "internal: tower_backward runs the heads, the embeddings and the weight-gradient queueing; the layers run through chain_layer_backward");
  }
  if (phase == PH_WGRAD_EMBED) {
    if (i == 0) FC_TRY(weight_grad(c, t.dtok, t.patches, B * (N - 1), D, cf.in_chans * cf.patch * cf.patch, grads + tp.pw, grads + tp.pb));
    return 0;
  }
  if (!(phase == PH_ALL || phase == PH_EMBED)) return 0;
  const void* dx = t.gx[0];
  if (i == 0) {
    int np = N - 1, kp = cf.in_chans * cf.patch * cf.patch;
    FC_TRY(fc_img_embed_bwd(c.dt, dx, grads + tp.pos, grads + tp.cls, t.dtok, B, N, D, c.s));
    FC_TRY(weight_grad(c, t.dtok, t.patches, B * np, D, kp, grads + tp.pw, grads + tp.pb));
  } else {
    FC_TRY(fc_txt_embed_bwd(c.dt, dx, w.ids, P + tp.word, P + tp.tpos, P + tp.ttype, t.emb_mean, t.emb_rstd, P + tp.lnw, grads + tp.word,
                            grads + tp.tpos, grads + tp.ttype, grads + tp.lnw, grads + tp.lnb, B, N, D, cf.vocab, c.s));
  }
  if (phase == PH_ALL) FC_TRY(flush_dw(c));
  return 0;
}

// Chain schedules, backward of layer l for the listed towers: every LayerNorm backward and every dX product takes all their rows in one
// launch; the weight gradients are queued through linear_bwd_params (a micro-batch chain runs with no_wgrad: the driver queues them
// over the full batch).
static int chain_layer_backward(const Ctx& c, Ws& w, const TowerList& T, int l, float* grads) {
  const fc_model* m = c.m;
  const fc_model_cfg& cf = m->cfg;
  const int D = cf.dim, Hd = cf.mlp_hidden, nt = T.n;
  const float* P = c.params;
  const int* tw = T.idx;
  LnBwdD ln[2];
  GemmD gd[2];
  const void* dm[2];
  const void* da[2];
  // ---- MLP branch: x_{l+1} = xmid + s2 * (gact.W2^T + b2);  dm = dx * s2 (written by the LayerNorm backward above it, or here for the top layer)
  for (int q = 0; q < nt; ++q) {
    const int i = tw[q];
    const BlockP& b = m->tw[i].blocks[l]; TowerWs& t = w.t[i]; LayerWs& L = t.L[l];
    const float* s2 = dp_ptr(m, w, i, l, 1);
    dm[q] = t.gx[l + 1];
    if (s2) {
      if (l == cf.depth - 1) FC_TRY(fc_rowscale(c.dt, t.gx[l + 1], L.gdm, s2, t.N, t.M, D, c.s));
      dm[q] = L.gdm;
    }
    FC_TRY(linear_bwd_params(c, b.fc2, dm[q], L.gact, t.M, grads));
    gd[q] = GemmD{dm[q], c.W(b.fc2.w), L.gdu, t.M, GemmEpi()};
    gd[q].e.gelu_in = L.u; gd[q].e.gelu_saved_grad = (c.dt == FC_BF16);                 // du = (dm.W2) * gelu'(u)
  }
  bool mlp_done = false;
#ifdef FC_PROBES
  if (m->mlp_fused) {      // du = (dm . W2) * gelu'(u) (stored: fc1's weight gradient reads it) and dh2 = du . W1 in one launch per tower
    mlp_done = true;
    for (int q = 0; q < nt && mlp_done; ++q) {
      const int i = tw[q];
      TowerWs& t = w.t[i]; LayerWs& L = t.L[l];
      const int r = fc_mlp_fused(1, dm[q], m->mlp_stream(c.wc, i, l, 1), nullptr, nullptr, L.gdu, L.u, nullptr, nullptr, 1, t.dh, t.M, D, Hd, c.s);
      if (r < 0) return r;
      FC_REQUIRE(r == 0 || q == 0, "internal: the fused MLP covered one tower of a layer and declined the other");
      mlp_done = r == 0;
    }
  }
#endif
  if (!mlp_done) FC_TRY(gemm_multi(c, FC_GEMM_NN, gd, nt, Hd, D));
  for (int q = 0; q < nt; ++q) {
    const int i = tw[q];
    const BlockP& b = m->tw[i].blocks[l]; TowerWs& t = w.t[i]; LayerWs& L = t.L[l];
    FC_TRY(linear_bwd_params(c, b.fc1, L.gdu, L.h2, t.M, grads));
    gd[q] = GemmD{L.gdu, c.W(b.fc1.w), t.dh, t.M, GemmEpi()};                            // dh2
  }
  if (!mlp_done) FC_TRY(gemm_multi(c, FC_GEMM_NN, gd, nt, D, Hd));
  // ---- attention branch: xmid = x_l + s1 * (o.Wp^T + bp); da = gxmid * s1 comes out of the same LayerNorm-backward pass
  for (int q = 0; q < nt; ++q) {
    const int i = tw[q];
    const BlockP& b = m->tw[i].blocks[l]; TowerWs& t = w.t[i]; LayerWs& L = t.L[l];
    const size_t lnp = (size_t)t.ln_stride;
    const float* s1 = dp_ptr(m, w, i, l, 0);
    ln[q] = LnBwdD{t.dh, L.xmid, L.mean2, L.rstd2, P + b.n2w, t.gx[l + 1], L.gxmid, grads + b.n2w, grads + b.n2b, t.M,
                   t.ln_partial + (2 * l + 1) * lnp, s1 ? L.gda : nullptr, s1, t.N};
    da[q] = s1 ? (const void*)L.gda : (const void*)L.gxmid;
  }
  FC_TRY(ln_bwd_multi(c, ln, nt, D));
  for (int q = 0; q < nt; ++q) {
    const int i = tw[q];
    const BlockP& b = m->tw[i].blocks[l]; TowerWs& t = w.t[i]; LayerWs& L = t.L[l];
    FC_TRY(linear_bwd_params(c, b.proj, da[q], L.o, t.M, grads));
    gd[q] = GemmD{da[q], c.W(b.proj.w), t.dO, t.M, GemmEpi()};
  }
  FC_TRY(gemm_multi(c, FC_GEMM_NN, gd, nt, D, D));
  for (int q = 0; q < nt; ++q) {
    TowerWs& t = w.t[tw[q]]; LayerWs& L = t.L[l];
    FC_TRY(c.attn_bwd(L.qkv, L.o, t.dO, L.lse, t.delta, L.gdqkv, t.M / t.N, t.N));
  }
  for (int q = 0; q < nt; ++q) {
    const int i = tw[q];
    const BlockP& b = m->tw[i].blocks[l]; TowerWs& t = w.t[i]; LayerWs& L = t.L[l];
    FC_TRY(linear_bwd_params(c, b.qkv, L.gdqkv, L.h1, t.M, grads));
    gd[q] = GemmD{L.gdqkv, c.W(b.qkv.w), t.dh, t.M, GemmEpi()};                         // dh1
  }
  FC_TRY(gemm_multi(c, FC_GEMM_NN, gd, nt, D, 3 * D));
  for (int q = 0; q < nt; ++q) {
    const int i = tw[q];
    const BlockP& b = m->tw[i].blocks[l]; TowerWs& t = w.t[i]; LayerWs& L = t.L[l];
    const size_t lnp = (size_t)t.ln_stride;
    const float* s2_below = l > 0 ? dp_ptr(m, w, i, l - 1, 1) : nullptr;      // the layer below wants gx[l] * s2(l-1) as its dm
    ln[q] = LnBwdD{t.dh, t.x[l], L.mean1, L.rstd1, P + b.n1w, L.gxmid, t.gx[l], grads + b.n1w, grads + b.n1b, t.M,
                   t.ln_partial + (2 * l) * lnp, s2_below ? t.L[l - 1].gdm : nullptr, s2_below, t.N};
  }
  return ln_bwd_multi(c, ln, nt, D);
}

// CrossModalReparamLinear: route dW_eff (mome.py:58-60).  Runs after the weight gradients exist (i.e. after the grouped launch).
static int tower_reparam_grads(const Ctx& c, int i, float* grads) {
  // CrossModalReparamLinear: route dW_eff (mome.py:58-60) -- every re-param linear of the model in one launch (re-param models are
  // uni-modal, mome.py:768: the call for the one present tower covers them all)
  const FcReparam* tab; int n;
  FC_TRY(reparam_table(c.m, &tab, &n));
  return fc_reparam_grad_grouped(tab, n, c.params, grads, c.s);
}

// The LAST weight-gradient chunk (the lowest layers of the image tower + the patch embedding) starts when the whole backward is done and
// nothing is left to overlap it with -- except the optimizer: fc_client_step steps every parameter that does not depend on that
// chunk first and waits for the chunk only before stepping the rest.
struct LateDw {
  bool pending = false;
  bool fused = false;             // fused optimizer: no segment waits for a chunk; the caller waits for ev_dw_out once, at the end
  bool wait_last = false;         // ... and for ev_dw_last: the last chunk was launched on a chain's stream AFTER the caller's stream joined the chains
  std::vector<char> late_seg;     // per segment: its gradient is written by the last chunk
};
// covered (optional, out): per segment, 1 when this backward OVERWRITES the segment's gradient with plain stores (grouped weight-gradient
// launches, LayerNorm reductions with ln_store): the caller need not zero it beforehand
static int backward_impl(const fc_model* m, const float* params, const void* wc, const float* d_out_img, const float* d_out_txt, float* grads,
                         Ws& w, hipStream_t s, LateDw* late = nullptr, const FcAdamW* fopt = nullptr, std::vector<char>* fused_seg = nullptr,
                         bool ln_store = false, std::vector<char>* covered = nullptr) {
  Ctx c{m, params, m->need_wc ? (const char*)wc : (const char*)params, s, m->dt, fc_esize(m->dt)};
  c.ln_accumulate = ln_store ? 0 : 1;
  std::vector<FcTnProblem> probs;
  std::vector<FcLnReduce> lnq;
  DwState dwst;
  FC_TRY(check_device(grads, "fc_backward"));
  FcTnProblem* probs_dev = nullptr;
  FcLnReduce* lntab_dev = nullptr;
  FC_TRY(ensure_tables(m, w, s, &probs_dev, &lntab_dev));
  if (m->dt == FC_BF16) {   // dW / db products are queued and launched in grouped chunks on the dW stream
    FC_TRY(ensure_side(m, s));
    c.defer = &probs;
    dwst.dev = probs_dev;
    dwst.max_probs = w.max_probs;
    c.dw = &dwst;
    if (fopt) {
      c.fopt = fopt;
      c.fused_seg = fused_seg;
      fused_seg->assign(m->segs.size(), 0);
    }
  }
  c.lnq = &lnq;
  const bool shared_attn = w.shared_g != nullptr;
  if (shared_attn) {      // the fallback weight-gradient paths accumulate (split-K atomics, colsum): start from zero like `grads`
    FC_CHECK_HIP(hipMemsetAsync(w.shared_g, 0, sizeof(float) * (size_t)(m->shared_hi - m->shared_lo), s));
    c.gshared = w.shared_g - m->shared_lo;
    late = nullptr;       // every weight gradient is complete before the sum below
  }
  const bool run0 = m->tw[0].present && d_out_img, run1 = m->tw[1].present && d_out_txt;
  const int nmb = (run0 && c.defer) ? microbatches(m, w.B) : 1;
  if (chain_schedule()) {
    const bool r1 = run1 && !FC_ABLATED("txt");
    const ChainPlan pl = chain_plan(m, w.B, run0, r1);
    const fc_model_cfg& cf = m->cfg;
    if (pl.nchains == 1) {
      TowerList T;
      if (run0) T.idx[T.n++] = 0;
      if (r1) T.idx[T.n++] = 1;
      const float* const douts[2] = {d_out_img, d_out_txt};
      for (int q = 0; q < T.n; ++q) FC_TRY(tower_backward(c, w, T.idx[q], douts[T.idx[q]], grads, PH_HEAD));
      for (int l = cf.depth - 1; l >= 0; --l) {
        FC_TRY(chain_layer_backward(c, w, T, l, grads));
        if (dw_flush_here(l)) FC_TRY(flush_dw(c));   // this chunk's weight gradients (all towers) start now, under the layers below
      }
      for (int q = 0; q < T.n; ++q) FC_TRY(tower_backward(c, w, T.idx[q], douts[T.idx[q]], grads, PH_EMBED));
      FC_TRY(flush_dw(c));
    } else {
      const int b0 = pl.b0, B = w.B;
      const size_t ow = (size_t)((w.feat_out || m->tw[0].task == FC_TASK_RTV) ? cf.dim : m->tw[0].ncls);
      // a classification head's weight gradients ACCUMULATE (generic GEMM, colsum): the head of the full batch on the caller's stream
      // before the chains fork, instead of once per chain on concurrent streams
      const bool cls_head = !(w.feat_out || m->tw[0].task == FC_TASK_RTV);
      if (cls_head) FC_TRY(tower_backward(c, w, 0, d_out_img, grads, PH_HEAD));
      Ws wa = slice_ws(m, w, 0, 0, b0, 0, false), wb = slice_ws(m, w, 0, b0, B - b0, 1, false);
      Ctx ca = c, cb = c, cf_ = c;
      ca.no_wgrad = cb.no_wgrad = true;
      cb.s = m->mbs[0];
      cf_.n_more = 1;
      TowerList TA, TB;
      TA.idx[TA.n++] = 0;
      TB.idx[TB.n++] = 0;
      if (r1) TB.idx[TB.n++] = 1;
      FC_CHECK_HIP(hipEventRecord(m->ev_fork, s));
      FC_CHECK_HIP(hipStreamWaitEvent(cb.s, m->ev_fork, 0));
      const float* dob = d_out_img + (size_t)b0 * ow;
      if (!cls_head) {
        FC_TRY(tower_backward(ca, wa, 0, d_out_img, grads, PH_HEAD));
        FC_TRY(tower_backward(cb, wb, 0, dob, grads, PH_HEAD));
      }
      if (r1) FC_TRY(tower_backward(cb, wb, 1, d_out_txt, grads, PH_HEAD));
      for (int l = cf.depth - 1; l >= 0; --l) {
        FC_TRY(chain_layer_backward(ca, wa, TA, l, grads));
        FC_TRY(chain_layer_backward(cb, wb, TB, l, grads));
        FC_TRY(tower_backward(cf_, w, 0, d_out_img, grads, PH_WGRAD_LAYER, l));       // full-batch weight gradients of this layer
        if (r1) FC_TRY(tower_backward(cf_, w, 1, d_out_txt, grads, PH_WGRAD_LAYER, l));
        if (dw_flush_here(l)) FC_TRY(flush_dw(cf_));
      }
      FC_TRY(tower_backward(ca, wa, 0, d_out_img, grads, PH_EMBED));
      FC_TRY(tower_backward(cb, wb, 0, dob, grads, PH_EMBED));
      if (r1) FC_TRY(tower_backward(cb, wb, 1, d_out_txt, grads, PH_EMBED));
      FC_TRY(tower_backward(cf_, w, 0, d_out_img, grads, PH_WGRAD_EMBED));
      FC_TRY(flush_dw(cf_));
      FC_STREAM_EV(4, cb.s); FC_STREAM_EV(5, s);
      FC_CHECK_HIP(hipEventRecord(m->ev_mb_join[0], cb.s));
      FC_CHECK_HIP(hipStreamWaitEvent(s, m->ev_mb_join[0], 0));
    }
  } else {
    const bool r1 = run1 && !FC_ABLATED("txt");
    const fc_model_cfg& cf = m->cfg;
    const bool deferred = c.defer != nullptr;        // bf16: the driver queues the weight gradients over the full batch, per layer
    ChainDef ch[4];
    const int n = build_chains(m, w, s, false, run0, r1, deferred, ch);
    int nimg = 0;
    for (int k = 0; k < n; ++k) nimg += ch[k].tower == 0;
    const float* const douts[2] = {d_out_img, d_out_txt};
    const size_t ow = (size_t)((w.feat_out || m->tw[0].task == FC_TASK_RTV) ? cf.dim : m->tw[0].ncls);
    // a classification head's weight gradients ACCUMULATE (generic GEMM, colsum): the head of the full batch on the caller's stream
    // before the chains fork, instead of once per chain on concurrent streams
    const bool cls_head_first = run0 && nimg > 1 && !(w.feat_out || m->tw[0].task == FC_TASK_RTV);
    if (cls_head_first) FC_TRY(tower_backward(c, w, 0, d_out_img, grads, PH_HEAD));
    FC_TRY(chains_fork(m, s, ch, n));
    bool joined = false;                             // the caller's stream already waits for every chain (early join below)
    Ctx cx[4], cf_ = c;
    hipStream_t extra[3];
    int nextra = 0;
    for (int k = 0; k < n; ++k) {
      cx[k] = c;
      cx[k].s = ch[k].s;
      cx[k].no_wgrad = deferred;
      if (ch[k].s != s) extra[nextra++] = ch[k].s;
    }
    cf_.n_more = nextra;
    cf_.more_s = extra;
    auto dptr = [&](int k) { return ch[k].tower == 0 ? d_out_img + (size_t)ch[k].b0 * ow : d_out_txt; };
    for (int k = 0; k < n; ++k)
      if (!(ch[k].tower == 0 && cls_head_first)) FC_TRY(tower_backward(cx[k], ch[k].w, ch[k].tower, dptr(k), grads, PH_HEAD));
    auto layer = [&](int k, int l) -> int {
      TowerList T;
      T.idx[T.n++] = ch[k].tower;
      Ws& wk = ch[k].w;
      return run_layer(cx[k], graph_key(cx[k], wk, 1, ch[k].tower, l, grads), [&](const Ctx& q) { return chain_layer_backward(q, wk, T, l, grads); });
    };
    // The text tower (short) first, all of it, with weight-gradient chunks of its own behind ITS stream only: they start early and keep
    // the weight-gradient stream busy under the image tower's backward, and the last, exposed chunk holds image layers only.  (All
    // towers' gradients in common chunks, flushed behind every chain: 4.74 -> 4.85 ms per ViT-S step, round 3.)
    static const bool text_first = fc_knob("FC_TEXT_FIRST", 1) != 0;
    int kt = -1;
    for (int k = 0; k < n; ++k)
      if (ch[k].tower == 1 && ch[k].s != s && deferred && text_first) kt = k;
    // Three backward chains: the text tower shares the weight-gradient stream, so its layers are enqueued chunk by chunk BETWEEN the
    // image chunks -- [text layers of chunk 1][text chunk][image chunk 1][text layers of chunk 2] ... -- instead of all of it first: the
    // image chunk of a group of layers then starts when those layers are done, not after the whole text backward.
    const bool interleave = kt >= 0 && ch[kt].s == m->dws;
    bool image_done = false;
    if (kt >= 0) {
      nextra = 0;                                        // the image chunks wait for the image chains only
      for (int k = 0; k < n; ++k)
        if (k != kt && ch[k].s != s) extra[nextra++] = ch[k].s;
    }
    if (interleave) {
      Ctx ct = c;
      ct.s = ch[kt].s;
      ct.n_more = 0;
      cf_.n_more = nextra;
      for (int l_hi = cf.depth - 1; l_hi >= 0;) {
        int l_lo = l_hi;
        while (l_lo > 0 && !dw_flush_here(l_lo)) --l_lo;
        for (int l = l_hi; l >= l_lo; --l) {
          FC_TRY(layer(kt, l));
          FC_TRY(tower_backward(ct, w, 1, d_out_txt, grads, PH_WGRAD_LAYER, l));
        }
        if (l_lo == 0) FC_TRY(tower_backward(cx[kt], ch[kt].w, 1, d_out_txt, grads, PH_EMBED));
        FC_TRY(flush_dw(ct));
        for (int l = l_hi; l >= l_lo; --l) {
          for (int k = 0; k < n; ++k)
            if (k != kt) FC_TRY(layer(k, l));
          if (run0) FC_TRY(tower_backward(cf_, w, 0, d_out_img, grads, PH_WGRAD_LAYER, l));
        }
        if (l_lo > 0) FC_TRY(flush_dw(cf_));             // the last image chunk follows the embedding backward below
        l_hi = l_lo - 1;
      }
      image_done = true;
    } else if (kt >= 0) {
      Ctx ct = c;
      ct.s = ch[kt].s;
      for (int l = cf.depth - 1; l >= 0; --l) {
        FC_TRY(layer(kt, l));
        FC_TRY(tower_backward(ct, w, 1, d_out_txt, grads, PH_WGRAD_LAYER, l));
        if (dw_flush_here(l)) FC_TRY(flush_dw(ct));
      }
      FC_TRY(tower_backward(cx[kt], ch[kt].w, 1, d_out_txt, grads, PH_EMBED));
      FC_TRY(flush_dw(ct));
      cf_.n_more = nextra;
    }
    for (int l = cf.depth - 1; l >= 0 && !image_done; --l) {
      for (int k = 0; k < n; ++k)
        if (k != kt) FC_TRY(layer(k, l));
      if (deferred) {
        for (int i = 0; i < 2; ++i)
          if ((i == 0 ? run0 : r1) && !(i == 1 && kt >= 0)) FC_TRY(tower_backward(cf_, w, i, douts[i], grads, PH_WGRAD_LAYER, l));
        if (dw_flush_here(l)) FC_TRY(flush_dw(cf_));   // this chunk's weight gradients start now, under the layers below
      }
    }
    for (int k = 0; k < n; ++k)
      if (k != kt) FC_TRY(tower_backward(cx[k], ch[k].w, ch[k].tower, dptr(k), grads, PH_EMBED));
    if (deferred) {
      if (run0) FC_TRY(tower_backward(cf_, w, 0, d_out_img, grads, PH_WGRAD_EMBED));
      if (late && !late->fused && dwst.flushed > 0 && dwst.flushed < probs.size()) {
        // two-phase optimizer: the segments the LAST chunk writes (whole blocks) are stepped after it, everything else while it runs
        FC_CHECK_HIP(hipEventRecord(m->ev_dw_prev, m->dws));      // every chunk but the last
        late->late_seg.assign(m->segs.size(), 0);
        for (size_t q = dwst.flushed; q < probs.size(); ++q) {
          const int64_t offs[2] = {(int64_t)(probs[q].C - grads), probs[q].bias_grad ? (int64_t)(probs[q].bias_grad - grads) : -1};
          for (int64_t off : offs)
            for (size_t k = 0; off >= 0 && k < m->segs.size(); ++k)
              if (m->segs[k].offset == off) late->late_seg[k] = 1;
        }
        for (size_t k = 0; k < m->segs.size(); ++k) {
          if (!late->late_seg[k] || strncmp(m->segs[k].name, "blockses.", 9) != 0) continue;
          const char* dot = strchr(m->segs[k].name + 9, '.');
          dot = dot ? strchr(dot + 1, '.') : nullptr;
          if (!dot) continue;
          const size_t plen = (size_t)(dot - m->segs[k].name) + 1;              // "blockses.<tower>.<layer>."
          for (size_t j = 0; j < m->segs.size(); ++j)
            if (strncmp(m->segs[j].name, m->segs[k].name, plen) == 0) late->late_seg[j] = 1;
        }
        late->pending = true;
      }
      // The last chunk is the only weight-gradient work nothing overlaps.  Making it shorter (FC_DW_FLUSH_AT = 6,2 / 6,1 / 7,3 / 6,3) and / or
      // narrow-tiled (more, smaller tiles) measured SLOWER every time (tools build, same box, ms per step: default 4.68-4.73, "6,2" 4.79-4.81,
      // "6,2" narrow 4.77-4.78, "6,1" narrow 4.81, "7,3" narrow 4.80, narrow alone 4.91; profiles/r03/dw_tail_ab.txt): the extra middle
      // chunk competes with the chains for CUs and costs more than the shorter tail returns.
      static const bool last_narrow = fc_knob("FC_DW_LAST_NARROW", 0) != 0;
      // three backward chains: the weight-gradient stream still carries the text tower's backward and the earlier chunks when the image
      // chains end, and their own streams fall idle -- the last chunk goes to the first extra image chain's stream (FC_DW_LAST_ON=0: dW stream)
      hipStream_t last_on = nullptr;
      static const int last_knob = fc_knob("FC_DW_LAST_ON", 1);
      if (kt >= 0 && ch[kt].s == m->dws && last_knob)
        for (int k = 0; k < n && !last_on; ++k)
          if (ch[k].tower == 0 && ch[k].s != s) last_on = ch[k].s;
      // Round 5 experiment (FC_DW_EARLY_JOIN=1, tools build): the caller's stream joins the chains WITHOUT waiting for the last chunk, so that what
      // follows on it (the LayerNorm reduction, the optimizer's remainder: ~70 us of launches) runs beside the chunk instead of behind it, and
      // fc_client_step waits for ev_dw_last at its very end.  Measured (profiles/r05/tail_overlap_*.txt): the two kernels do run under the chunk
      // (next step starts 31 us after it instead of 81), but the chunk itself starts ~40 us later behind the extra event records and runs 6 %
      // longer beside them: 4.432 vs 4.427 ms per step -- a wash, off.
      static const int early_knob = fc_knob("FC_DW_EARLY_JOIN", 0);
      if (last_on && late && late->fused && early_knob) {      // the chains' join events are recorded BEFORE the chunk is enqueued on one of them ...
        for (int k = 0; k < n; ++k)
          if (ch[k].s != s) FC_CHECK_HIP(hipEventRecord(ch[k].join, ch[k].s));
        joined = true;
      }
      FC_TRY(flush_dw(cf_, last_narrow, last_on));             // (its own dependencies stay direct: one event hop, as before)
      if (joined) {                                            // ... and waited for after it: the caller's stream does not wait for the chunk
        for (int k = 0; k < n; ++k)
          if (ch[k].s != s) FC_CHECK_HIP(hipStreamWaitEvent(s, ch[k].join, 0));
        FC_CHECK_HIP(hipEventRecord(m->ev_dw_last, last_on));
        late->wait_last = true;
      }
    }
    FC_STREAM_EV(3, m->side); FC_STREAM_EV(4, m->mbs[0]); FC_STREAM_EV(5, s);
    if (!joined) FC_TRY(chains_join(s, ch, n));
  }
  if (covered) {
    covered->assign(m->segs.size(), 0);
    auto mark = [&](const float* p) {
      if (!p || p < grads || p >= grads + m->total) return;
      const int64_t off = p - grads;
      size_t lo = 0, hi = m->segs.size();
      while (lo + 1 < hi) { const size_t mid = (lo + hi) / 2; if (m->segs[mid].offset <= off) lo = mid; else hi = mid; }
      if (m->segs[lo].offset == off) (*covered)[lo] = 1;
    };
    for (const FcTnProblem& q : probs) { mark(q.C); mark(q.bias_grad); }
    if (ln_store) for (const FcLnReduce& e : lnq) { mark(e.dg); mark(e.db); }
  }
  if (!lnq.empty()) {
    FC_REQUIRE((int)lnq.size() <= w.max_ln, "internal: too many queued LayerNorm reductions");
    const void* lt = nullptr;
    FC_TRY(cached_table(lnq.data(), lnq.size() * sizeof(FcLnReduce), &lt));
    lntab_dev = (FcLnReduce*)lt;
    FC_TRY(fc_ln_reduce_grouped(lntab_dev, (int)lnq.size(), m->cfg.dim, s));
  }
  if (!probs.empty()) {   // every chunk was launched by flush_dw; the main stream continues after the last one
    FC_CHECK_HIP(hipEventRecord(m->ev_dw_out, m->dws));
    FC_STREAM_EV(6, m->dws);
    if (late && late->fused) late->pending = true;                                       // the caller waits for ev_dw_out itself, later
    else if (late && late->pending) FC_CHECK_HIP(hipStreamWaitEvent(s, m->ev_dw_prev, 0));   // ... likewise, after the first optimizer phase
    else FC_CHECK_HIP(hipStreamWaitEvent(s, m->ev_dw_out, 0));
  }
  if (shared_attn) {      // both towers' contributions exist now (dW stream joined above): owner += borrower
    if (!m->shared_dev) {
      FC_CHECK_HIP(hipMalloc(&m->shared_dev, m->shared_chunks.size() * sizeof(FcProxChunk)));
      FC_CHECK_HIP(hipMemcpy(m->shared_dev, m->shared_chunks.data(), m->shared_chunks.size() * sizeof(FcProxChunk), hipMemcpyHostToDevice));
    }
    FC_TRY(fc_add_chunks(grads, c.gshared, (const FcProxChunk*)m->shared_dev, (int)m->shared_chunks.size(), s));
  }
  if (m->tw[0].present && d_out_img) FC_TRY(tower_reparam_grads(c, 0, grads));
  if (m->tw[1].present && d_out_txt) FC_TRY(tower_reparam_grads(c, 1, grads));
  return 0;
}

extern "C" int fc_backward(const fc_model_t* m, const float* params, const void* wc, const float* d_out_img, const float* d_out_txt,
                           float* grads, void* workspace, size_t workspace_bytes, void* stream) {
  FC_REQUIRE(m->last.ws == workspace && workspace, "fc_backward: no matching fc_forward on this workspace precedes this call");
  Ws w;
  FC_TRY(check_ws(m, m->last.B, m->last.n_txt, workspace, workspace_bytes, w));
  w.feat_out = m->last.feat_out; w.droppath = m->last.droppath; w.ids = m->last.ids;
  return backward_impl(m, params, wc, d_out_img, d_out_txt, grads, w, (hipStream_t)stream);
}

// ---------------------------------------------------------------- losses / optimizer / step
extern "C" int fc_contrastive_loss_fwd_bwd(const float* a, const float* b, int32_t B, int32_t D, float tau, float* scratch, size_t scratch_floats,
                                           float* lossbuf, float* da, float* db, void* stream) {
  FC_REQUIRE(scratch_floats >= fc_contrastive_scratch_floats(B), "contrastive: scratch too small");
  return fc_contrastive_fwd_bwd(a, b, B, D, tau, scratch, lossbuf, da, db, (hipStream_t)stream);
}
extern "C" int fc_ce_loss_fwd_bwd(const float* logits, const int64_t* y, int32_t B, int32_t C, float* lossbuf, float* dlogits, void* stream) {
  return fc_ce_fwd_bwd(logits, y, B, C, lossbuf, dlogits, (hipStream_t)stream);
}

// AdamW of every trainable segment the fused weight-gradient epilogue did NOT step (flag 0 in `fused`), as one chunked launch.  The
// chunk table depends only on the model and on which problems took the grouped path, so it is built once and kept in the handle.
static int adamw_rest(const fc_model* m, const std::vector<char>& fused, const FcAdamW& o, hipStream_t s) {
  if (m->fused_host != fused || !m->rest_valid) {
    std::vector<FcProxChunk> ch;
    int64_t beg = -1, end = -1;
    auto cut = [&]() {
      for (int64_t q = beg; q >= 0 && q < end; q += FC_PROX_CHUNK) ch.push_back(FcProxChunk{q, (int32_t)std::min<int64_t>(FC_PROX_CHUNK, end - q), 0});
      beg = end = -1;
    };
    for (size_t k = 0; k < m->segs.size(); ++k) {
      const fc_segment& sg = m->segs[k];
      const bool take = sg.trainable && sg.numel > 0 && !(k < fused.size() && fused[k]);
      if (!take) continue;
      if (beg >= 0 && sg.offset != end) cut();
      if (beg < 0) beg = sg.offset;
      end = sg.offset + sg.numel;
    }
    cut();
    m->rest_host.assign((const char*)ch.data(), (const char*)ch.data() + ch.size() * sizeof(FcProxChunk));
    m->rest_chunks = (int)ch.size();
    m->fused_host = fused;
    m->rest_valid = true;
  }
  if (m->rest_chunks == 0) return 0;
  // the device table is looked up on EVERY step: the cache owns it and may have started over since the last one (cached_table)
  const void* t = nullptr;
  FC_TRY(cached_table(m->rest_host.data(), m->rest_host.size(), &t));
  return fc_adamw_chunks((const FcProxChunk*)t, m->rest_chunks, o, s);
}
static int adamw_ranges(const fc_model* m, float* p, float* g, float* mm, float* vv, float lr, float b1, float b2, float eps, float wd, int step,
                        hipStream_t s, bf16_t* shadow = nullptr, const std::vector<char>* late_seg = nullptr, int want_late = 0) {
  // contiguous runs of trainable segments (padding included) -> one launch each; frozen segments are skipped like torch.
  // late_seg / want_late: only the segments whose flag equals want_late (fc_client_step's two optimizer phases)
  auto take = [&](size_t k) { return m->segs[k].trainable && (!late_seg || (int)(*late_seg)[k] == want_late); };
  size_t i = 0, n = m->segs.size();
  while (i < n) {
    if (!take(i)) { ++i; continue; }
    size_t j = i;
    while (j + 1 < n && take(j + 1)) ++j;
    int64_t beg = m->segs[i].offset;
    int64_t end = (j + 1 < n) ? m->segs[j + 1].offset : m->total;
    FC_TRY(fc_adamw(p + beg, g + beg, mm + beg, vv + beg, (size_t)(end - beg), lr, b1, b2, eps, wd, step, shadow ? shadow + beg : nullptr, 0, s));
    i = j + 1;
  }
  return 0;
}
extern "C" int fc_adamw_step(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int32_t step, void* stream) {
  FC_REQUIRE(step >= 1, "adamw: step is 1-based");
  return adamw_ranges(m, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}

// torch.optim.SGD.step over the trainable ranges of the flat buffers (frozen segments skipped like torch skips grad-less parameters)
extern "C" int fc_sgd_step(const fc_model_t* m, float* params, const float* grads, float* momentum_buf, float lr, float momentum, int32_t nesterov,
                           float weight_decay, int32_t step, void* stream) {
  FC_REQUIRE(step >= 1, "sgd: step is 1-based");
  FC_REQUIRE(momentum == 0.f || momentum_buf, "sgd: momentum needs a buffer");
  FC_REQUIRE(!nesterov || momentum > 0.f, "Nesterov momentum requires a momentum and zero dampening");      // torch's own check
  size_t i = 0, n = m->segs.size();
  while (i < n) {
    if (!m->segs[i].trainable) { ++i; continue; }
    size_t j = i;
    while (j + 1 < n && m->segs[j + 1].trainable) ++j;
    const int64_t beg = m->segs[i].offset, end = (j + 1 < n) ? m->segs[j + 1].offset : m->total;
    FC_TRY(fc_sgd(params + beg, grads + beg, momentum_buf ? momentum_buf + beg : nullptr, (size_t)(end - beg), lr, momentum, nesterov, weight_decay, step == 1,
                  nullptr, (hipStream_t)stream));
    i = j + 1;
  }
  return 0;
}

// exp(clamp(log(1/0.07), 0, log 100)) evaluated in fp32 like the upstream nn.Parameter (a fresh criterion is built
// every step at fedavgclient.py:95, so the temperature never trains)
static float contrastive_tau() { return expf(fminf(fmaxf(logf(1.0f / 0.07f), 0.0f), logf(100.0f))); }

// ---- FedProx proximal term (src/client/fedproxclient.py:64-67) over the trainable parameter tensors
static void prox_tables(const fc_model* m, std::vector<FcProxChunk>& chunks, std::vector<int32_t>& first) {
  int seg = 0;
  first.clear();
  chunks.clear();
  for (const fc_segment& sg : m->segs) {
    if (!sg.trainable || sg.numel <= 0) continue;   // a frozen tensor never leaves its global value: norm 0, no contribution
    first.push_back((int32_t)chunks.size());
    for (int64_t o = 0; o < sg.numel; o += FC_PROX_CHUNK)
      chunks.push_back(FcProxChunk{sg.offset + o, (int32_t)std::min<int64_t>(FC_PROX_CHUNK, sg.numel - o), seg});
    ++seg;
  }
  first.push_back((int32_t)chunks.size());
}
static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
extern "C" size_t fc_prox_scratch_bytes(const fc_model_t* m) {
  std::vector<FcProxChunk> ch;
  std::vector<int32_t> first;
  prox_tables(m, ch, first);
  return align16(ch.size() * sizeof(FcProxChunk)) + align16(first.size() * sizeof(int32_t)) + align16(ch.size() * sizeof(float)) +
         align16(first.size() * sizeof(float));
}
extern "C" int fc_prox_term(const fc_model_t* m, const float* params, const float* global_params, float mu, int32_t B, float* grads, float* lossbuf,
                            void* scratch, size_t scratch_bytes, void* stream) {
  FC_REQUIRE(params && global_params && grads && lossbuf && scratch, "fc_prox_term: null buffer");
  FC_REQUIRE(scratch_bytes >= fc_prox_scratch_bytes(m), "fc_prox_term: scratch too small (%zu < %zu)", scratch_bytes, fc_prox_scratch_bytes(m));
  hipStream_t s = (hipStream_t)stream;
  std::vector<FcProxChunk> ch;
  std::vector<int32_t> first;
  prox_tables(m, ch, first);
  const size_t b0 = align16(ch.size() * sizeof(FcProxChunk)), b1 = align16(first.size() * sizeof(int32_t)), b2 = align16(ch.size() * sizeof(float));
  char* base = (char*)scratch;
  std::vector<char> img(b0 + b1, 0);
  memcpy(img.data(), ch.data(), ch.size() * sizeof(FcProxChunk));
  memcpy(img.data() + b0, first.data(), first.size() * sizeof(int32_t));
  // The tables are re-sent on every call (~50 KB): the scratch is the caller's, and a freed-and-reallocated buffer can come back at
  // the same address with other contents, so "same pointer" proves nothing about what it holds.
  if (m->prox_host != img) {
    FC_CHECK_HIP(hipStreamSynchronize(s));        // an in-flight copy may still read the old host image
    m->prox_host = img;
  }
  m->prox_dev = scratch;
  FC_CHECK_HIP(hipMemcpyAsync(base, m->prox_host.data(), m->prox_host.size(), hipMemcpyHostToDevice, s));
  return fc_prox_term_impl(params, global_params, (const FcProxChunk*)base, (int)ch.size(), (const int32_t*)(base + b0), (int)first.size() - 1,
                           (float*)(base + b0 + b1), (float*)(base + b0 + b1 + b2), mu, B, grads, lossbuf, s);
}

// ---- torch.nn.utils.clip_grad_norm_ over the trainable parameter tensors (creamflclient.py:232, creamflserver.py:334)
int fc_clip_impl(float* grads, const FcProxChunk* chunks, int nchunks, float* partial, float* coef_norm, float max_norm, hipStream_t s);
extern "C" size_t fc_clip_scratch_bytes(const fc_model_t* m) {
  std::vector<FcProxChunk> ch;
  std::vector<int32_t> first;
  prox_tables(m, ch, first);
  return align16(ch.size() * sizeof(FcProxChunk)) + align16(ch.size() * sizeof(float)) + 16;
}
extern "C" int fc_clip_grad_norm(const fc_model_t* m, float* grads, float max_norm, void* scratch, size_t scratch_bytes, float* total_norm_out,
                                 void* stream) {
  FC_REQUIRE(grads && scratch, "fc_clip_grad_norm: null buffer");
  FC_REQUIRE(scratch_bytes >= fc_clip_scratch_bytes(m), "fc_clip_grad_norm: scratch too small");
  hipStream_t s = (hipStream_t)stream;
  std::vector<FcProxChunk> ch;
  std::vector<int32_t> first;
  prox_tables(m, ch, first);
  const size_t b0 = align16(ch.size() * sizeof(FcProxChunk)), b1 = align16(ch.size() * sizeof(float));
  std::vector<char> img(b0, 0);
  memcpy(img.data(), ch.data(), ch.size() * sizeof(FcProxChunk));
  char* base = (char*)scratch;
  if (m->clip_host != img) {                      // (see fc_prox_term: the table is re-sent on every call)
    FC_CHECK_HIP(hipStreamSynchronize(s));
    m->clip_host = img;
  }
  m->clip_dev = scratch;
  FC_CHECK_HIP(hipMemcpyAsync(base, m->clip_host.data(), m->clip_host.size(), hipMemcpyHostToDevice, s));
  float* coef_norm = (float*)(base + b0 + b1);
  FC_TRY(fc_clip_impl(grads, (const FcProxChunk*)base, (int)ch.size(), (float*)(base + b0), coef_norm, max_norm, s));
  if (total_norm_out) FC_CHECK_HIP(hipMemcpyAsync(total_norm_out, coef_norm + 1, sizeof(float), hipMemcpyDeviceToDevice, s));
  return 0;
}

// ---- torch.optim.AdamW with per-parameter step counts: seg_steps[i] is the 1-based step of segment i in THIS call, 0 = the
// parameter has no gradient and is skipped (torch skips parameters whose .grad is None and does not advance their step)
extern "C" int fc_adamw_step_segs(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, const int32_t* seg_steps, int32_t n_segments, void* wc, void* stream) {
  FC_REQUIRE(params && grads && exp_avg && exp_avg_sq && seg_steps, "fc_adamw_step_segs: null buffer");
  FC_REQUIRE(n_segments == (int32_t)m->segs.size(), "fc_adamw_step_segs: %d step entries for %zu segments", n_segments, m->segs.size());
  bool has_aux = false;
  for (const fc_segment& sg : m->segs)
    if (strstr(sg.name, "aux_weight")) has_aux = true;
  bf16_t* shadow = (wc && m->need_wc && m->dt == FC_BF16 && !has_aux) ? (bf16_t*)wc : nullptr;
  const size_t n = m->segs.size();
  size_t i = 0;
  while (i < n) {
    if (!m->segs[i].trainable || seg_steps[i] <= 0) { ++i; continue; }
    size_t j = i;
    while (j + 1 < n && m->segs[j + 1].trainable && seg_steps[j + 1] == seg_steps[i]) ++j;
    const int64_t beg = m->segs[i].offset, end = (j + 1 < n) ? m->segs[j + 1].offset : m->total;
    FC_TRY(fc_adamw(params + beg, grads + beg, exp_avg + beg, exp_avg_sq + beg, (size_t)(end - beg), lr, beta1, beta2, eps, weight_decay,
                    seg_steps[i], shadow ? shadow + beg : nullptr, 0, (hipStream_t)stream));
    i = j + 1;
  }
  if (wc && m->need_wc && !shadow) FC_TRY(fc_prepare_weights(m, params, wc, stream));
  else if (shadow) FC_TRY(mlp_pack_all(m, wc, (hipStream_t)stream));
  return 0;
}

static int client_step_impl(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* wc, const float* img,
                            const int64_t* ids, const int64_t* labels, int32_t B, int32_t n_txt, const float* droppath, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int32_t step, float* lossbuf, void* workspace, size_t workspace_bytes,
                            void* stream, const float* global_params, float mu, void* prox_scratch, size_t prox_scratch_bytes);
#ifdef FC_PROBES
static hipEvent_t g_phase_ev[64 * 5];
static bool g_phase_on = false;
#endif
// Whole-step graph replay.  A step of the ViT-S client is ~550 launches on four streams and costs the host 2.7-4 ms to enqueue (of a 4.4-ms
// step): after two eager steps with the same buffers and shapes the third is captured (thread-local stream capture: the forks and joins of
// the internal streams become graph edges) and every later one is ONE hipGraphLaunch.  Kernel arguments are baked in, so the key holds
// every address and shape, and the step-dependent AdamW constants are read from device memory (FcAdamW::dyn).  Anything the capture cannot
// hold (FedProx term, separate optimizer, a first step that still records which gradients need zeroing) declines and runs eagerly.
enum { FC_STEP_DECLINED = 77 };
static thread_local bool g_step_capture = false;
extern "C" int fc_client_step(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* wc, const float* img,
                              const int64_t* ids, const int64_t* labels, int32_t B, int32_t n_txt, const float* droppath, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int32_t step, float* lossbuf, void* workspace, size_t workspace_bytes,
                              void* stream) {
  auto eager = [&]() {
    return client_step_impl(m, params, grads, exp_avg, exp_avg_sq, wc, img, ids, labels, B, n_txt, droppath, lr, beta1, beta2, eps, weight_decay, step,
                            lossbuf, workspace, workspace_bytes, stream, nullptr, 0.f, nullptr, 0);
  };
#ifndef FC_PROBES
  return eager();
#else
  if (!m->step_graph || m->dt != FC_BF16) return eager();
  hipStream_t s = (hipStream_t)stream;
  fc_model::StepKey key;
  memset(&key, 0, sizeof(key));
  const void* ptrs[12] = {params, grads, exp_avg, exp_avg_sq, wc, img, ids, labels, droppath, lossbuf, workspace, stream};
  memcpy(key.p, ptrs, sizeof(ptrs));
  key.B = B; key.n_txt = n_txt; key.ws_bytes = workspace_bytes;
  key.beta1 = beta1; key.beta2 = beta2; key.eps = eps; key.weight_decay = weight_decay;
  if (m->step_graphs.size() >= 16 && !m->step_graphs.count(key)) {      // the addresses keep changing (a drop-path table reallocated every step, say)
    for (auto it = m->step_graphs.begin(); it != m->step_graphs.end();)  // forget the keys that never became a graph; stay eager if all of them did
      it = it->second.exec ? std::next(it) : m->step_graphs.erase(it);
    if (m->step_graphs.size() >= 16) return eager();
  }
  fc_model::StepEntry& e = m->step_graphs[key];
  if (e.declined) return eager();
  const FcAdamW consts = fc_adamw_consts(lr, beta1, beta2, eps, weight_decay, step);
  if (e.exec) {
    FC_TRY(fc_adamw_set_dyn(m->adamw_dyn, consts, s));
    FC_CHECK_HIP(hipGraphLaunch(e.exec, s));
    ++m->step_graph_hits;
    m->last = LastFwd{workspace, B, n_txt, (m->tw[0].present && m->tw[1].present) ? 1 : 0, droppath, ids};
    return 0;
  }
  if (e.seen < 2) { ++e.seen; return eager(); }
  if (!m->adamw_dyn) FC_CHECK_HIP(hipMalloc(&m->adamw_dyn, 4 * sizeof(float)));
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); e.declined = true; fc_set_error("step graph declined: hipStreamBeginCapture failed"); return eager(); }
  g_step_capture = true;
  const int r = eager();
  g_step_capture = false;
  const hipError_t ce = hipStreamEndCapture(s, &graph);
  if (r != 0 || ce != hipSuccess || !graph) {
    const std::string why = r == FC_STEP_DECLINED ? std::string("the step's preconditions (fused optimizer, known gradient coverage, every segment trainable)")
                                                  : (r != 0 ? std::string("error during capture: ") + g_err : std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    e.declined = true;
    fc_set_error("step graph declined: %s", why.c_str());
    if (r != 0 && r != FC_STEP_DECLINED) return r;
    return eager();
  }
  const hipError_t ie = hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ie != hipSuccess) { (void)hipGetLastError(); e.exec = nullptr; e.declined = true; fc_set_error("step graph declined: hipGraphInstantiate: %s", hipGetErrorString(ie)); return eager(); }
  FC_TRY(fc_adamw_set_dyn(m->adamw_dyn, consts, s));
  FC_CHECK_HIP(hipGraphLaunch(e.exec, s));
  ++m->step_graph_hits;
  return 0;
#endif
}
extern "C" int fc_client_step_prox(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* wc, const float* img,
                                   const int64_t* ids, const int64_t* labels, int32_t B, int32_t n_txt, const float* droppath, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int32_t step, float* lossbuf, void* workspace,
                                   size_t workspace_bytes, void* stream, const float* global_params, float mu, void* prox_scratch,
                                   size_t prox_scratch_bytes) {
  FC_REQUIRE(global_params && prox_scratch, "fc_client_step_prox: null global parameters / scratch");
  return client_step_impl(m, params, grads, exp_avg, exp_avg_sq, wc, img, ids, labels, B, n_txt, droppath, lr, beta1, beta2, eps, weight_decay, step,
                          lossbuf, workspace, workspace_bytes, stream, global_params, mu, prox_scratch, prox_scratch_bytes);
}
static int client_step_impl(const fc_model_t* m, float* params, float* grads, float* exp_avg, float* exp_avg_sq, void* wc, const float* img,
                            const int64_t* ids, const int64_t* labels, int32_t B, int32_t n_txt, const float* droppath, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int32_t step, float* lossbuf, void* workspace, size_t workspace_bytes,
                            void* stream, const float* global_params, float mu, void* prox_scratch, size_t prox_scratch_bytes) {
  hipStream_t s = (hipStream_t)stream;
#ifdef FC_PROBES
#define FC_PHASE(i) do { if (g_phase_on) (void)hipEventRecord(g_phase_ev[(step & 63) * 5 + (i)], s); } while (0)
  static const bool g_phase_env = getenv("FC_STEP_PHASES") != nullptr;
  if (g_phase_env && !g_phase_on) {
    for (int i = 0; i < 64 * 5; ++i) (void)hipEventCreate(&g_phase_ev[i]);
    for (int i = 0; i < 64 * 8; ++i) (void)hipEventCreate(&g_stream_ev[i]);
    g_phase_on = g_stream_on = true;
  }
  g_stream_step = step;
#else
#define FC_PHASE(i) do {} while (0)
#endif
  FC_PHASE(0);
  FC_REQUIRE(grads && exp_avg && exp_avg_sq && lossbuf, "fc_client_step: null buffer");
  FC_REQUIRE(step >= 1, "fc_client_step: step is 1-based");
  bool both = m->tw[0].present && m->tw[1].present;
  if (!both) FC_REQUIRE(labels != nullptr, "fc_client_step: uni-modal clients need labels");
  Ws w;
  // optimizer.zero_grad() fedavgclient.py:79 -- only what the backward ACCUMULATES into (embeddings, the shared final norm, heads, re-param
  // routing): the linears' and the blocks' LayerNorm gradients (93 % of the buffer) are overwritten by plain stores.  Which segments those
  // are is recorded by the first step of a handle (which zeroes everything) and depends on the model and the batch shape only.
  const bool know_cover = m->cover_B == B && m->cover_ntxt == n_txt && !m->zero_runs.empty();
  if (g_step_capture) {      // a captured step must take the fused-optimizer path (its constants come from device memory); nothing is enqueued yet
    bool plain = know_cover && !global_params && m->need_wc && m->dt == FC_BF16 && !m->cfg.colearn_attn && fc_knob("FC_FUSED_OPT", 1) != 0;
    for (const fc_segment& sg : m->segs)
      if (!sg.trainable || strstr(sg.name, "aux_weight")) plain = false;
    if (!plain) return FC_STEP_DECLINED;
  }
  // The zero-fills (four small launches, ~7 us each back to back) go to the text tower's stream, which has slack in the forward, instead of
  // standing in front of the image chains; that stream first waits for the caller's (the previous step's optimizer read these gradients), and
  // the caller's stream waits for it again behind the forward (FC_ZERO_SIDE=0, tools build: on the caller's stream as before)
  hipStream_t zs = s;
  static const int zero_side = fc_knob("FC_ZERO_SIDE", 1);
  if (both && m->dt == FC_BF16 && zero_side && !chain_schedule()) {
    FC_TRY(ensure_side(m, s));
    if (m->side && m->side != s) {
      FC_CHECK_HIP(hipEventRecord(m->ev_zero, s));
      FC_CHECK_HIP(hipStreamWaitEvent(m->side, m->ev_zero, 0));
      zs = m->side;
    }
  }
  if (know_cover) {
    for (const std::pair<int64_t, int64_t>& r : m->zero_runs) FC_CHECK_HIP(hipMemsetAsync(grads + r.first, 0, sizeof(float) * (size_t)r.second, zs));
  } else {
    FC_CHECK_HIP(hipMemsetAsync(grads, 0, sizeof(float) * (size_t)m->total, zs));
  }
  FC_CHECK_HIP(hipMemsetAsync(lossbuf + 1, 0, sizeof(float), zs));
  FC_TRY(forward_impl(m, params, wc, img, ids, B, n_txt, both ? 1 : 0, droppath, workspace, workspace_bytes, nullptr, nullptr, s, w));
  if (zs != s) {      // (the forward joined the text stream already; this names the dependency of the loss and the backward on the zero-fills)
    FC_CHECK_HIP(hipEventRecord(m->ev_zero, zs));
    FC_CHECK_HIP(hipStreamWaitEvent(s, m->ev_zero, 0));
  }
  m->last = LastFwd{workspace, B, w.n_txt, both ? 1 : 0, droppath, ids};
  FC_PHASE(1);
  const float *d0 = nullptr, *d1 = nullptr;
  if (both) {  // fedavgclient.py:91-95
    FC_TRY(fc_contrastive_fwd_bwd(w.t[0].out, w.t[1].out, B, m->cfg.dim, contrastive_tau(), w.loss_scratch, lossbuf, w.dout[0], w.dout[1], s));
    d0 = w.dout[0]; d1 = w.dout[1];
  } else {     // fedavgclient.py:81-90
    int i = m->tw[0].present ? 0 : 1;
    FC_REQUIRE(m->tw[i].task == FC_TASK_CLS, "fc_client_step: uni-modal tower must be a 'cls' task");
    FC_TRY(fc_ce_fwd_bwd(w.t[i].logits, labels, B, m->tw[i].ncls, lossbuf, w.dout[i], s));
    (i == 0 ? d0 : d1) = w.dout[i];
  }
  bool aux_any = false;
  for (const fc_segment& sg : m->segs)
    if (strstr(sg.name, "aux_weight")) aux_any = true;
  static const bool late_opt = fc_knob("FC_LATE_OPT", 1) != 0;
  static const bool fused_opt = fc_knob("FC_FUSED_OPT", 1) != 0;
  // compute weights for the next step: without re-param linears and with every segment trainable the bf16 shadow is written
  // by the optimizer itself (one pass over the parameters instead of two)
  bool all_trainable = true;
  for (const fc_segment& sg : m->segs)
    if (!sg.trainable) all_trainable = false;
  const bool fuse_shadow = m->need_wc && m->dt == FC_BF16 && !aux_any && all_trainable;
  // ... and in that case the AdamW step of every linear's weight and bias is taken by the weight-gradient launches themselves, chunk by
  // chunk under the backward; what they do not cover (LayerNorm, embeddings, heads: 7 % of the parameters) is one small launch.
  // (FedProx adds its term to the finished gradients, so it keeps the separate optimizer.)
  const bool fused = fused_opt && fuse_shadow && !global_params && !m->cfg.colearn_attn;
  LateDw late;
  FcAdamW fo = fc_adamw_consts(lr, beta1, beta2, eps, weight_decay, step);
  if (g_step_capture) fo.dyn = m->adamw_dyn;
  fo.g0 = grads; fo.p = params; fo.m = exp_avg; fo.v = exp_avg_sq; fo.shadow = (bf16_t*)wc;
  std::vector<char> fseg;
  late.fused = fused;
  FC_PHASE(2);
  std::vector<char> covered;
  FC_TRY(backward_impl(m, params, wc, d0, d1, grads, w, s, (fused || (late_opt && !aux_any && !global_params)) ? &late : nullptr,
                       fused ? &fo : nullptr, &fseg, /*ln_store=*/true, know_cover ? nullptr : &covered));
  if (!know_cover) {
    m->zero_runs.clear();
    int64_t beg = -1, end = -1;
    for (size_t k = 0; k < m->segs.size(); ++k) {
      if (covered[k]) continue;
      const int64_t o = m->segs[k].offset, e = (k + 1 < m->segs.size()) ? m->segs[k + 1].offset : m->total;
      if (beg >= 0 && o == end) { end = e; continue; }
      if (beg >= 0) m->zero_runs.push_back({beg, end - beg});
      beg = o; end = e;
    }
    if (beg >= 0) m->zero_runs.push_back({beg, end - beg});
    if (m->zero_runs.empty()) m->zero_runs.push_back({0, 0});
    m->cover_B = B; m->cover_ntxt = n_txt;
  }
  FC_PHASE(3);
  if (global_params)   // FedproxClient.update: loss += mu * 0.5 * sum ||p - p_global||, before the optimizer step (fedproxclient.py:64-72)
    FC_TRY(fc_prox_term(m, params, global_params, mu, B, grads, lossbuf, prox_scratch, prox_scratch_bytes, stream));
  if (fused) {
    FC_TRY(adamw_rest(m, fseg, fo, s));
    if (late.pending) FC_CHECK_HIP(hipStreamWaitEvent(s, m->ev_dw_out, 0));
    if (late.wait_last) FC_CHECK_HIP(hipStreamWaitEvent(s, m->ev_dw_last, 0));
  } else if (late.pending) {   // everything that does not wait for the last weight-gradient chunk, then the chunk, then the rest
    FC_TRY(adamw_ranges(m, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, s, fuse_shadow ? (bf16_t*)wc : nullptr,
                        &late.late_seg, 0));
    FC_CHECK_HIP(hipStreamWaitEvent(s, m->ev_dw_out, 0));
    FC_TRY(adamw_ranges(m, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, s, fuse_shadow ? (bf16_t*)wc : nullptr,
                        &late.late_seg, 1));
  } else {
    FC_TRY(adamw_ranges(m, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, s, fuse_shadow ? (bf16_t*)wc : nullptr));
  }
  if (m->need_wc && !fuse_shadow) FC_TRY(fc_prepare_weights(m, params, wc, stream));
  else if (fuse_shadow) FC_TRY(mlp_pack_all(m, wc, s));      // the optimizer wrote the bf16 shadow: refresh the fused MLP's streams from it
  FC_PHASE(4);
  return 0;
}
#ifdef FC_PROBES
// tools build only: ms from the start of `step` to the end of each internal stream's part: [0..2] forward text / chain 1 / chain 0,
// [3..5] backward text / chain 1 / chain 0, [6] last weight-gradient chunk
extern "C" int fc_dbg_stream_events(int step, float* ms7) {
  if (!g_stream_on) return -1;
  hipEvent_t e0 = g_phase_ev[(step & 63) * 5];
  for (int i = 0; i < 7; ++i) {
    (void)hipEventSynchronize(g_stream_ev[(step & 63) * 8 + i]);
    (void)hipEventElapsedTime(ms7 + i, e0, g_stream_ev[(step & 63) * 8 + i]);
  }
  return 0;
}
// tools build only (FC_STEP_PHASES=1): GPU time of the last step's phases on the caller's stream: forward, loss, backward (up to the
// late weight-gradient chunk), optimizer tail
extern "C" int fc_dbg_step_phases(int step, float* ms5) {     // [0..3] phases of `step`, [4] gap from the previous step's end to this step's start
  if (!g_phase_on) return -1;
  hipEvent_t* e = g_phase_ev + (step & 63) * 5;
  (void)hipEventSynchronize(e[4]);
  for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(ms5 + i, e[i], e[i + 1]);
  ms5[4] = 0.f;
  if (step > 1) (void)hipEventElapsedTime(ms5 + 4, g_phase_ev[((step - 1) & 63) * 5 + 4], e[0]);
  return 0;
}
#endif

// ---------------------------------------------------------------- aggregation
extern "C" int fc_aggregate_blend(float* out, const float* global, const float* const* client_bases, int32_t n_clients,
                                  const int64_t* seg_offset, const int64_t* seg_numel, const int64_t* src_offset, const float* seg_weights,
                                  int32_t n_segments, void* stream) {
  return fc_blend_segments(out, global, client_bases, n_clients, seg_offset, seg_numel, src_offset, seg_weights, n_segments, (hipStream_t)stream);
}
// copy of the last forward's head outputs (logits / features) out of the workspace: metric tracking in FedavgClient.update
extern "C" int fc_copy_outputs(const fc_model_t* m, void* workspace, size_t workspace_bytes, float* out_img, float* out_txt, void* stream) {
  FC_REQUIRE(m->last.ws == workspace && workspace, "fc_copy_outputs: no forward on this workspace");
  Ws w;
  FC_TRY(check_ws(m, m->last.B, m->last.n_txt, workspace, workspace_bytes, w));
  float* outs[2] = {out_img, out_txt};
  for (int i = 0; i < 2; ++i) {
    if (!m->tw[i].present || !outs[i]) continue;
    bool normalize = m->last.feat_out || m->tw[i].task == FC_TASK_RTV;
    const float* src = normalize ? w.t[i].out : w.t[i].logits;
    size_t width = normalize ? (size_t)m->cfg.dim : (size_t)m->tw[i].ncls;
    FC_CHECK_HIP(hipMemcpyAsync(outs[i], src, sizeof(float) * m->last.B * width, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  }
  return 0;
}
extern "C" int fc_scale_segments(float* buf, const int64_t* seg_offset, const int64_t* seg_numel, const float* seg_weight, int32_t n_segments,
                                 void* stream) {
  return fc_scale_segments_impl(buf, seg_offset, seg_numel, seg_weight, n_segments, (hipStream_t)stream);
}

// ---------------------------------------------------------------- workspace inspection (tests: op-by-op parity of the bf16 path)
extern "C" int fc_workspace_tensor(const fc_model_t* m, int32_t B, int32_t n_txt, int32_t tower, int32_t layer, const char* name, size_t* offset,
                                   size_t* bytes) {
  FC_REQUIRE(m && name && offset && bytes, "fc_workspace_tensor: null argument");
  FC_REQUIRE(tower >= 0 && tower < 2 && m->tw[tower].present, "fc_workspace_tensor: tower %d absent", tower);
  Ws w;
  carve(m, B, m->tw[1].present ? n_txt : 0, (void*)256, w);      // any non-null base: offsets are differences
  char* base = (char*)256;
  const TowerWs& t = w.t[tower];
  const fc_model_cfg& c = m->cfg;
  const size_t es = fc_esize(m->dt), MD = (size_t)t.M * c.dim * es, MH = (size_t)t.M * c.mlp_hidden * es, M4 = sizeof(float) * (size_t)t.M;
  const void* p = nullptr;
  size_t n = 0;
  const std::string k = name;
  if (k == "x" || k == "gx") {
    FC_REQUIRE(layer >= 0 && layer <= c.depth, "fc_workspace_tensor: layer %d out of range", layer);
    p = k == "x" ? t.x[layer] : t.gx[layer]; n = MD;
  } else if (k == "patches") { p = t.patches; n = (size_t)B * (t.N - 1) * c.in_chans * c.patch * c.patch * es; }
  else if (k == "dtok") { p = t.dtok; n = (size_t)B * (t.N - 1) * c.dim * es; }
  else if (k == "out") { p = t.out; n = sizeof(float) * (size_t)B * c.dim; }
  else if (k == "f") { p = t.f; n = sizeof(float) * (size_t)B * c.dim; }
  else if (k == "dout") { p = w.dout[tower]; n = sizeof(float) * (size_t)B * c.dim; }
  else {
    FC_REQUIRE(layer >= 0 && layer < c.depth, "fc_workspace_tensor: layer %d out of range", layer);
    const LayerWs& L = t.L[layer];
    if (k == "h1") { p = L.h1; n = MD; } else if (k == "qkv") { p = L.qkv; n = 3 * MD; } else if (k == "o") { p = L.o; n = MD; }
    else if (k == "xmid") { p = L.xmid; n = MD; } else if (k == "h2") { p = L.h2; n = MD; } else if (k == "u") { p = L.u; n = MH; }
    else if (k == "gact") { p = L.gact; n = MH; } else if (k == "gxmid") { p = L.gxmid; n = MD; } else if (k == "gdm") { p = L.gdm; n = MD; }
    else if (k == "gda") { p = L.gda; n = MD; } else if (k == "gdu") { p = L.gdu; n = MH; } else if (k == "gdqkv") { p = L.gdqkv; n = 3 * MD; }
    else if (k == "mean1") { p = L.mean1; n = M4; } else if (k == "rstd1") { p = L.rstd1; n = M4; } else if (k == "mean2") { p = L.mean2; n = M4; }
    else if (k == "rstd2") { p = L.rstd2; n = M4; } else if (k == "lse") { p = L.lse; n = sizeof(float) * (size_t)B * c.heads * t.N; }
  }
  FC_REQUIRE(p != nullptr, "fc_workspace_tensor: unknown tensor '%s'", name);
  *offset = (size_t)((const char*)p - base);
  *bytes = n;
  return 0;
}

// ---------------------------------------------------------------- kernel-level test entry points
extern "C" int fc_k_layernorm_fwd(int32_t dt, const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int32_t M,
                                  int32_t D, float eps, void* stream) {
  return fc_layernorm_fwd(dt, x, g, b, y, mean, rstd, M, D, eps, (hipStream_t)stream);
}
extern "C" int fc_k_layernorm_bwd(int32_t dt, const void* dy, const void* x, const float* mean, const float* rstd, const float* g, const void* res,
                                  void* dx, float* dg, float* db, int32_t M, int32_t D, void* stream) {
  return fc_layernorm_bwd(dt, dy, x, mean, rstd, g, res, dx, dg, db, M, D, (hipStream_t)stream);
}
// the in-model form of the LayerNorm backward: per-block dgamma / dbeta partial rows (room for fc_k_layernorm_partial_floats(M, D)
// floats) + the grouped reduction, instead of atomics
extern "C" size_t fc_k_layernorm_partial_floats(int32_t M, int32_t D) {      // (the partial rows are fp64: two floats per element)
  return 2 * ((size_t)fc_layernorm_bwd_partial_blocks(M) * 2 * D) + 64;
}
extern "C" int fc_k_layernorm_bwd_partial(int32_t dt, const void* dy, const void* x, const float* mean, const float* rstd, const float* g,
                                          const void* res, void* dx, float* dg, float* db, int32_t M, int32_t D, float* partial, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  fc_ln_part_t* part = (fc_ln_part_t*)partial;
  int r = fc_layernorm_bwd(dt, dy, x, mean, rstd, g, res, dx, dg, db, M, D, s, part);
  if (r != 1) return r;
  // one-entry reduction table at the tail of `partial` (the grouped reduction adds into dg / db)
  FcLnReduce e{part, nullptr, dg, db, fc_layernorm_bwd_partial_blocks(M), 0, D, 1, nullptr, 0, dt == FC_F32 ? 1 : 0};
  FcLnReduce* tab = (FcLnReduce*)(part + (size_t)fc_layernorm_bwd_partial_blocks(M) * 2 * D);
  FC_CHECK_HIP(hipMemcpyAsync(tab, &e, sizeof(e), hipMemcpyHostToDevice, s));
  FC_CHECK_HIP(hipStreamSynchronize(s));      // `e` is a stack temporary (test entry point)
  return fc_ln_reduce_grouped(tab, 1, D, s);
}
// the bf16 MFMA GEMM with the epilogues the model uses (tools / tests): bias [+ residual] | bias + GELU with gelu'(u) saved | x mul_in
extern "C" int fc_k_gemm_epi(int32_t kind, const void* A, const void* Bm, void* C, int32_t M, int32_t N, int32_t K, const float* bias, const void* res,
                             void* gelu_grad_out, const void* mul_in, void* stream) {
  GemmEpi e;
  e.bias = bias; e.res = res;
  if (gelu_grad_out) { e.preact = gelu_grad_out; e.gelu_saved_grad = 1; }
  if (mul_in) { e.gelu_in = mul_in; e.gelu_saved_grad = 1; }
  FC_REQUIRE(kind == FC_GEMM_NT || kind == FC_GEMM_NN, "fc_k_gemm_epi: kind must be NT (0) or NN (1)");
  const long ldb = kind == FC_GEMM_NT ? K : N;
  const int r = fc_gemm_mfma(kind, FC_BF16, (const bf16_t*)A, K, (const bf16_t*)Bm, ldb, C, N, M, N, K, e, (hipStream_t)stream);
  FC_REQUIRE(r <= 0, "fc_k_gemm_epi: shape / alignment not covered by the MFMA kernel");
  return r;
}
extern "C" int fc_k_gemm(int32_t impl, int32_t kind, int32_t dt_in, int32_t dt_out, const void* A, const void* Bm, void* C, int32_t M, int32_t N,
                         int32_t K, const float* bias, int32_t gelu, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GemmEpi e;
  e.bias = bias;
  FC_REQUIRE(!gelu, "fc_k_gemm: gelu epilogue is exercised through fc_forward");
  if (impl == 1) {
    long lda = kind == FC_GEMM_TN ? M : K, ldb = kind == FC_GEMM_NT ? K : N;
    if (dt_in == FC_F32) {          // fp32 in and out: three bf16 MFMA products of split operands (the fp32 mode's GEMM)
      FC_REQUIRE(dt_out == FC_F32, "fc_k_gemm: the split-operand MFMA path writes fp32");
      return fc_gemm_x3(kind, (const float*)A, lda, (const float*)Bm, ldb, (float*)C, N, M, N, K, e, s);
    }
    FC_REQUIRE(dt_in == FC_BF16, "fc_k_gemm: MFMA path takes bf16 or fp32 inputs");
    return fc_gemm_mfma(kind, dt_out, (const bf16_t*)A, lda, (const bf16_t*)Bm, ldb, C, N, M, N, K, e, s);
  }
  long sam, sak, sbk, sbn;
  if (kind == FC_GEMM_NT) { sam = K; sak = 1; sbk = 1; sbn = K; }
  else if (kind == FC_GEMM_NN) { sam = K; sak = 1; sbk = N; sbn = 1; }
  else { sam = 1; sak = M; sbk = N; sbn = 1; }
  return fc_gemm_generic(dt_in, dt_in, dt_out, A, sam, sak, Bm, sbk, sbn, C, N, M, N, K, e, s);
}
extern "C" int fc_k_attention_fwd(int32_t impl, int32_t dt, const void* qkv, void* o, float* lse, int32_t B, int32_t N, int32_t H, int32_t d,
                                  float scale, void* stream) {
  if (impl == 1) {
    if (dt == FC_F32) return fc_attn_f32_fwd((const float*)qkv, (float*)o, lse, B, N, H, d, scale, (hipStream_t)stream);   // fp32 MFMA (the fp32 mode's)
    FC_REQUIRE(dt == FC_BF16, "MFMA attention takes bf16 or fp32");
    return fc_attn_fwd_mfma((const bf16_t*)qkv, (bf16_t*)o, lse, B, N, H, d, scale, (hipStream_t)stream);
  }
  return fc_attn_fwd_generic(dt, qkv, o, lse, B, N, H, d, scale, (hipStream_t)stream);
}
extern "C" int fc_k_attention_bwd(int32_t impl, int32_t dt, const void* qkv, const void* o, const void* dout, const float* lse, float* delta,
                                  void* dqkv, int32_t B, int32_t N, int32_t H, int32_t d, float scale, void* stream) {
  if (impl == 1) {
    if (dt == FC_F32)
      return fc_attn_f32_bwd((const float*)qkv, (const float*)o, (const float*)dout, lse, delta, (float*)dqkv, B, N, H, d, scale, (hipStream_t)stream);
    FC_REQUIRE(dt == FC_BF16, "MFMA attention takes bf16 or fp32");
    return fc_attn_bwd_mfma((const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, B, N, H, d, scale,
                            (hipStream_t)stream);
  }
  return fc_attn_bwd_generic(dt, qkv, o, dout, lse, delta, dqkv, B, N, H, d, scale, (hipStream_t)stream);
}
// one weight-gradient problem through the grouped kernels (wide = 1: 128x384 tiles, needs in % 384 == 0): tests / tools
extern "C" int fc_k_dw(int32_t wide, const void* dY, const void* X, float* dW, float* db, int32_t rows, int32_t out, int32_t in, void* stream) {
  if (wide == 3) {      // the fp32 mode's form: fp32 operands, split-operand MFMA products, sliced reduction; db ACCUMULATES (as in the model)
    const int r = fc_dw_x3((const float*)dY, (const float*)X, dW, db, rows, out, in, (hipStream_t)stream);
    FC_REQUIRE(r <= 0, "fc_k_dw: fp32 operands must be 16-B aligned with out, in multiples of 8");
    FC_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return r;
  }
  FcTnProblem p{(const bf16_t*)dY, (const bf16_t*)X, dW, db, out, in, in, out, in, rows, 0, 0};   // lda, ldb, ldc, M, N, K
  FC_REQUIRE(fc_gemm_tn_grouped_supported(p), "fc_k_dw: operands must be 16-B aligned with out, in multiples of 8");
  int tiles;
  if (wide) {
    FC_REQUIRE(fc_gemm_dw_wide_supported(p), "fc_k_dw: the wide kernel needs in %% 384 == 0");
    tiles = fc_gemm_dw_wide_tiles(p, &p.tiles_n);
  } else {
    p.tiles_n = fc_cdiv(in, 128);
    tiles = fc_cdiv(out, 128) * p.tiles_n;
  }
  FcTnProblem* dev = nullptr;
  FC_CHECK_HIP(hipMalloc(&dev, sizeof(p)));
  FC_CHECK_HIP(hipMemcpy(dev, &p, sizeof(p), hipMemcpyHostToDevice));
  int r = wide ? fc_gemm_dw_wide(dev, 1, tiles, (hipStream_t)stream, nullptr, wide) : fc_gemm_tn_grouped(dev, 1, tiles, (hipStream_t)stream);
  FC_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  FC_CHECK_HIP(hipFree(dev));
  return r;
}
#ifdef FC_PROBES
extern "C" int fc_k_mlp_pack(const void* W1, const void* W2, void* stream_fwd, void* stream_bwd, int32_t D, int32_t Hd, void* stream) {
  FcMlpPackJob job{(const bf16_t*)W1, (const bf16_t*)W2, (bf16_t*)stream_fwd, (bf16_t*)stream_bwd};
  FcMlpPackJob* dev = nullptr;
  FC_CHECK_HIP(hipMalloc(&dev, sizeof(job)));
  FC_CHECK_HIP(hipMemcpy(dev, &job, sizeof(job), hipMemcpyHostToDevice));
  const int r = fc_mlp_pack(dev, 1, D, Hd, (hipStream_t)stream);
  FC_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  FC_CHECK_HIP(hipFree(dev));
  return r;
}
extern "C" int fc_k_mlp_fused(int32_t bwd, const void* X, const void* Wp, const float* b1, const float* b2, void* act, void* gsave, const void* res,
                              const float* rowscale, int32_t rows_per_sample, void* out, int32_t M, int32_t D, int32_t Hd, void* stream) {
  return fc_mlp_fused(bwd, X, Wp, b1, b2, act, gsave, res, rowscale, rows_per_sample, out, M, D, Hd, (hipStream_t)stream);
}
#endif
extern "C" int fc_k_adamw(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float wd, int32_t step,
                          void* stream) {
  return fc_adamw(p, g, m, v, (size_t)n, lr, beta1, beta2, eps, wd, step, nullptr, 0, (hipStream_t)stream);
}
extern "C" int fc_k_cast(int32_t dt_out, const float* src, void* dst, int64_t n, void* stream) {
  return fc_cast(dt_out, src, dst, (size_t)n, (hipStream_t)stream);
}
