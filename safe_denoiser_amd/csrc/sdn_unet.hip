// UNet2DConditionModel forward (row U1/U2) as a static launch plan over libsdn's operators.
//
// Host-only logic: from the config it derives (a) the parameter manifest -- every diffusers state_dict key the
// network needs, with the layout it takes inside ONE packed weight buffer -- and (b) per batch size, a linear
// list of kernel launches with all activation addresses resolved at plan time inside ONE workspace (liveness-
// based reuse so the hot working set stays L2 / Infinity-Cache resident).  sdn_unet_forward() then only walks
// the list and launches; it never allocates or synchronises (hipGraph-capturable).
//
// Wiring follows the reference's vendored spec: models/unet.py:683-932 (forward), models/unet_2d_blocks.py
// :769-924 (mid), :1174-1426 (down), :2416-2704 (up), models/transformer_2d.py:239-359,505-540,810-858, and the
// diffusers-0.29.0 leaf definitions restated in SURVEY.md appendix A.
//
// Fusions relative to the reference graph (results identical up to rounding):
//   * to_q/to_k/to_v of self-attention = one GEMM over the stacked [3C, C] weight; to_k/to_v of cross-attention
//     = one GEMM over [2C, 768];
//   * all 22 ResnetBlock2D time_emb_proj linears = ONE GEMM per forward ([B,1280] x [sum Cout, 1280]); its f32
//     rows are added inside conv1's epilogue together with the conv bias;
//   * SiLU(temb) folded into time_embedding.linear_2's epilogue (temb is only ever consumed through SiLU);
//   * residual adds, shortcut adds, GEGLU, bias: GEMM epilogues; torch.cat([h, skip]) is never materialised for
//     the 1x1 shortcut (two-source A operand) and is written once, already normalised, by GroupNorm for conv1;
//   * nearest-2x upsample and the stride-2 downsample are index arithmetic inside the conv's im2col loader;
//   * conv_out writes the fp32 NCHW latent layout directly.
#include <stdio.h>
#include <string.h>

#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "sdn_common.h"
#include "sdn_ops.h"

namespace {

enum Space { SP_NONE = 0, SP_W = 1, SP_WS = 2, SP_LATENTS = 3, SP_TEXT = 4, SP_OUT = 5, SP_POOLED = 6,
             SP_KV = 7 };   // SP_KV: the workspace's persistent tail (text K / V slots: written by one op, read by one op, never recycled)
struct Ref { int space = SP_NONE; int64_t off = 0; };

enum OpKind { OP_TEMB, OP_CONV_IN, OP_GEMM, OP_GN, OP_LN, OP_ATTN, OP_PATCHIFY, OP_UNPATCHIFY, OP_LATENT_MIX, OP_SOFTMAX,
              OP_TRANSPOSE, OP_GAUSS, OP_REPEAT, OP_CLIP_EMBED, OP_MATTN, OP_ROWSTATS, OP_FFN, OP_SPLIT3 };

struct Op {
  int kind;
  sdn_gemm_desc gd;
  Ref a, a2, w, bias, rowbias, rowgate, residual, out, aux;
  Ref q2, k2, v2, out2;      // joint attention: second token stream
  Ref col, cols1, cols2;     // GEMM: column partials to emit; GroupNorm: partials of its input(s) to reduce instead of reading
  Ref ln_c, ln_d, ln_stats;  // GEMM with LayerNorm folded in (sdn_gemm_ln_*); ln_stats unset = statistics inside the kernel
  int ln = 0;
  int text_kv = 0;           // cross-attention K / V projection of the TEXT operand: skipped while the caller's text version stands
  int x3t = 0;               // bf16x3 plan: this GEMM runs on sdn_gemm_bf16 over triple operands (gd holds the EXPANDED K / Cin)
  int pair_in = 0;           // bf16x3 plan: attention whose q / k / v are column blocks of ONE hi | lo pair-row buffer (sdn_attention_x3_pairs)
  int tri_out = 0;           // bf16x3 plan: GroupNorm / LayerNorm / attention write the bf16 hi|lo|hi triple a GEMM will read
  int n1 = 0, mod = 0, ld_mod = 0, patch = 0;
  // GN / LN / conv_in / attention scalars
  int batch = 0, hw = 0, c1 = 0, c2 = 0, groups = 0, silu = 0;
  float eps = 0.f;
  int64_t rows = 0;
  int heads = 0, nq = 0, nk = 0, hd = 0, ldq = 0, ldk = 0, ldv = 0, ldo = 0;
  float scale = 0.f;
  Ref k, v;
  char label[24] = {0};      // kernel symbol this op launches (profiling rows are aggregated by it)
  double flops = 0.0;        // algorithmic FLOPs of this launch
  double bytes = 0.0;        // algorithmic HBM bytes of this launch (operands read once + result written once)
};

struct Arena {                      // plan-time first-fit allocator with coalescing; offsets are 256-B aligned
  std::map<int64_t, int64_t> free_;  // off -> size
  int64_t top = 0, peak = 0;
  static int64_t up(int64_t v) { return (v + 255) & ~(int64_t)255; }
  int64_t alloc(int64_t bytes) {
    bytes = up(bytes);
    for (auto it = free_.begin(); it != free_.end(); ++it) {
      if (it->second >= bytes) {
        const int64_t off = it->first, rest = it->second - bytes;
        free_.erase(it);
        if (rest > 0) free_[off + bytes] = rest;
        return off;
      }
    }
    // extend the top (merge with a free block that touches the top)
    if (!free_.empty()) {
      auto last = std::prev(free_.end());
      if (last->first + last->second == top) {
        const int64_t off = last->first;
        free_.erase(last);
        top = off + bytes;
        if (top > peak) peak = top;
        return off;
      }
    }
    const int64_t off = top;
    top += bytes;
    if (top > peak) peak = top;
    return off;
  }
  void release(int64_t off, int64_t bytes) {
    bytes = up(bytes);
    auto it = free_.emplace(off, bytes).first;
    auto nx = std::next(it);
    if (nx != free_.end() && it->first + it->second == nx->first) { it->second += nx->second; free_.erase(nx); }
    if (it != free_.begin()) {
      auto pv = std::prev(it);
      if (pv->first + pv->second == it->first) { pv->second += it->second; free_.erase(it); }
    }
  }
};

struct Act {                         // a bf16 [rows, C] activation living in the workspace
  int64_t off = -1, bytes = 0; int C = 0, hw = 0, side = 0;
  int64_t content = 0;               // bytes of the tensor in the plan's storage type (== bytes except in the bf16x3 plan, whose
                                     // slots are sized for the 6-byte-per-element triple form as well)
  int64_t st_off = -1, st_bytes = 0;   // column partials its producing GEMM leaves for the GroupNorm that reads it
};

struct Plan {
  int batch = 0;
  int64_t tscalar_off = -1;          // 256-byte workspace slot holding the step's timestep (graph mode)
  std::vector<Op> ops;
  int64_t ws_bytes = 0;
  int64_t kv_base = 0;               // byte offset of the persistent tail inside the workspace (= the recycled arena's peak)
  double flops = 0.0, attn_flops = 0.0;
};

}  // namespace

struct sdn_unet {
  sdn_unet_config cfg;
  sdn_mmdit_config mcfg;
  sdn_vae_config vcfg;
  bool is_mmdit = false;
  bool is_vae = false;
  bool is_vae_encoder = false;
  bool is_clip = false;
  sdn_clip_config ccfg;
  const void* clip_mask = nullptr;      // key-padding mask of the forward in flight (nullable)
  std::vector<sdn_param_info> params;
  std::map<std::string, int> param_index;
  int64_t weight_bytes = 0;
  std::map<int, Plan> plans;
  int tproj_total = 0;
  int64_t subbatch_bytes = 0;            // >0: run transformer blocks on batch slices of at most this many bytes per
                                         // activation.  Measured at B = 64 (tools/profile_ops.py, SUBBATCH=...): 48 MB
                                         // -> +5 %, 24 MB -> +10 % forward time, i.e. no cache-residency win -> OFF.
  bool profile_next = false;
  // graph mode (sdn_unet_set_graph_mode): one captured hipGraph per (batch, operand addresses); replays cost one launch
  bool use_graph = false;
  bool gn_fuse = true;                  // GroupNorm statistics from the producing GEMMs' column partials (hw % 128 == 0)
  bool ln_fold = true;                  // BasicTransformerBlock LayerNorms folded into their consumer GEMMs where it pays
  int ln_prepass_all = 0;               // debug A/B: 1 = every folded LayerNorm takes its row statistics from the pre-pass
  bool ff_fuse = true;                  // FeedForward's output linear and the block's proj_out (no nonlinearity between them)
  bool ffn_own_stats = true;            // k_ffn320 takes norm3's row statistics from its own operand fragments (no sdn_row_stats pass)
  bool ffn_fuse = true;                 // ... and the GEGLU projection in front of them: one launch, hidden activation in LDS (C = 320)
                                        // contracted into ONE GEMM over [ff | h3] with the product weight (sdn_linear_pair_fold)
  struct FoldJob { int64_t w, gamma, beta, bias, wf, c, d; int rows, cols; int kind = 0; int group = 0; };   // kind 0: LayerNorm fold; 1: linear pair; 2: bf16x3 weight expansion (sdn_expand3_weights)
  std::vector<FoldJob> fold_jobs;       // what sdn_unet_prepare has to compute into the SDN_P_DERIVED regions
  // Cross-attention K / V of the text (16 projections per forward, M = batch x 77) depend on the text operand alone, which the
  // denoising loop changes a handful of times in 50 steps: their outputs live in never-recycled workspace slots and the launches
  // are skipped while the caller-declared text version (sdn_unet_set_text_version; 0 = undeclared) equals the one they were
  // computed for, on the same batch / weights / text / workspace addresses.  Same bits: the skipped launches would rewrite them.
  uint64_t text_version = 0, kv_version = 0;
  int kv_batch = 0;
  const void *kv_w = nullptr, *kv_text = nullptr, *kv_ws = nullptr;
  bool x3_pairs = true;                 // bf16x3 plan: self-attention on pre-split operands (qkv projection writes hi | lo pair rows)
  bool res_pre = true;                  // attention output projections: residual into the accumulators before the k loop (sdn_gemm_desc.res_pre)
  bool x3_expand = true;                // dtype 3: GEMM operands as bf16 triples on the LDS-DMA tiles (false: the f32-staging k_gemm_x3 everywhere)
  bool split_k = false;                 // sdn_unet_set_split_k: small-M GEMMs of the plan take the split-K form (off by
                                        // default: it changes fp32 summation order with the batch size, and batch rows are
                                        // otherwise bit-identical whatever the batch)
  struct GraphKey {
    int batch; const void *w, *lat, *text, *pooled, *out, *ws;
    bool operator<(const GraphKey& o) const {
      return std::tie(batch, w, lat, text, pooled, out, ws) < std::tie(o.batch, o.w, o.lat, o.text, o.pooled, o.out, o.ws);
    }
  };
  std::map<GraphKey, hipGraphExec_t> graphs;
  hipStream_t cap_stream = nullptr;     // capture happens here (the caller's stream may be the legacy null stream)
  ~sdn_unet() {
    for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
    for (auto e : ev) (void)hipEventDestroy(e);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
  }
  std::vector<hipEvent_t> ev;          // 2 per op of the profiled forward
  int profiled_batch = 0;
};

namespace {

struct Builder {
  sdn_unet* u;
  Plan* plan;
  Arena arena;
  int B;
  int es = 2;                // bytes per activation / matrix-weight element: 2 (bf16 | f16 storage) or 4 (the fp32 precision mode)
  bool x3t = false;          // bf16x3 by operand expansion (SD-v1.4 UNet plan, dtype 3): GEMM operands are bf16 hi|lo|hi triples
  bool x3t_hold = false;     // ... except inside this scope (the per-sample time-embedding GEMMs: M = batch, nothing to gain)
  std::set<int64_t> tri;     // workspace offsets that currently hold a triple (set by its producer, cleared by drop())
  Ref tproj;                 // f32 [B, tproj_total]
  int tproj_cursor = 0;      // column offset of the next resnet's slice
  Ref gn_stats;

  // ---- parameters -------------------------------------------------------------------------------
  Ref param(const std::string& name, int kind, int rows, int cols, int rows_padded = 0) {
    auto it = u->param_index.find(name);
    if (it != u->param_index.end()) return Ref{SP_W, u->params[it->second].offset};
    sdn_param_info pi;
    memset(&pi, 0, sizeof(pi));
    snprintf(pi.name, sizeof(pi.name), "%s", name.c_str());
    pi.kind = kind; pi.rows = rows; pi.cols = cols; pi.rows_padded = rows_padded > rows ? rows_padded : rows;
    const int64_t esz = (kind == SDN_P_VEC_F32 || kind == SDN_P_GEGLU_VEC) ? 4 : es;
    const int64_t bytes = (int64_t)pi.rows_padded * (cols > 0 ? cols : 1) * esz;
    pi.offset = u->weight_bytes;
    u->weight_bytes += (bytes + 255) & ~(int64_t)255;
    u->param_index[name] = (int)u->params.size();
    u->params.push_back(pi);
    return Ref{SP_W, pi.offset};
  }
  // members of a stacked matrix must be byte-contiguous: their sizes are multiples of 256 B for every SD width
  Ref stacked(const std::vector<std::string>& names, int rows_each, int cols) {
    Ref first;
    int64_t expect = -1;
    for (size_t i = 0; i < names.size(); ++i) {
      Ref r = param(names[i], SDN_P_MAT, rows_each, cols);
      if (i == 0) first = r;
      else if (r.off != expect) { fprintf(stderr, "libsdn: stacked weight %s is not contiguous\n", names[i].c_str()); abort(); }
      expect = r.off + (int64_t)rows_each * cols * es;
    }
    return first;
  }

  // ---- activations ------------------------------------------------------------------------------
  Act act(int64_t rows, int C, int hw = 0, int side = 0, int esz = 0) {
    const bool dflt = esz == 0;
    if (esz == 0) esz = es;
    Act t; t.content = rows * C * esz;
    t.bytes = (x3t && dflt) ? rows * C * 6 : t.content;      // a default-typed slot may hold the f32 tensor or its triple
    t.off = arena.alloc(t.bytes); t.C = C; t.hw = hw; t.side = side; return t;
  }
  // an activation a GroupNorm will read: its producer (a GEMM) also emits per-128-row-block column sums
  Act act_gn(int64_t rows, int C, int hw, int side) {
    Act t = act(rows, C, hw, side);
    if (u->gn_fuse && !u->split_k && hw > 0 && hw % 128 == 0) {
      t.st_bytes = ((rows + 127) / 128) * (int64_t)C * 8;
      t.st_off = arena.alloc(t.st_bytes);
    }
    return t;
  }
  Ref pending_cols;                    // set by want_stats() for the NEXT emitted GEMM
  void want_stats(const Act& out) { pending_cols = out.st_off >= 0 ? Ref{SP_WS, out.st_off} : Ref(); }
  void drop(Act& t) {
    if (t.off >= 0) { arena.release(t.off, t.bytes); tri.erase(t.off); pairs.erase(t.off); }
    if (t.st_off >= 0) arena.release(t.st_off, t.st_bytes);
    t.off = -1; t.st_off = -1;
  }
  static Ref R(const Act& t) { return Ref{SP_WS, t.off}; }

  // ---- bf16x3 by operand expansion (include/sdn.h) -------------------------------------------------
  // expanded copy of an f32 weight region [rows, cols] (stacked matrices are contiguous, so w.off names the whole operand)
  Ref x3_weight(Ref w, int rows, int cols, int group) {
    const std::string name = "x3@" + std::to_string((long long)w.off);
    const bool fresh = u->param_index.find(name) == u->param_index.end();
    Ref d = derived(name, (int64_t)rows * 3 * cols * 2);
    if (fresh) { sdn_unet::FoldJob j{w.off, -1, -1, -1, d.off, -1, -1, rows, cols}; j.kind = 2; j.group = group; u->fold_jobs.push_back(j); }
    return d;
  }
  // the triple of an f32 tensor that no producer could write in that form (a raw residual-stream tensor, a skip concatenation)
  Act split3(Ref a, Ref a2, int64_t rows, int c1, int c2, int hw = 0, int side = 0) {
    Act t = act(rows, c1 + c2, hw, side);
    Op o; o.kind = OP_SPLIT3; o.a = a; o.a2 = a2; o.rows = rows; o.c1 = c1; o.c2 = c2; o.out = R(t);
    o.bytes = 10.0 * (double)rows * (c1 + c2);
    snprintf(o.label, sizeof(o.label), "k_split3");
    plan->ops.push_back(o);
    tri.insert(t.off);
    return t;
  }
  int64_t kv_top = 0;             // bytes of persistent text K / V slots handed out so far (space SP_KV)
  bool res_pre_next = false;      // the next gemm() adds its residual into the accumulators before the k loop (sdn_gemm_desc.res_pre)
  bool triple_out_next = false;   // the next gemm() writes the triple of its result (its only reader is another x3 GEMM)
  bool pair_out_next = false;     // the next gemm() writes hi | lo pair rows (a projection whose only reader is sdn_attention_x3_pairs)
  bool plan_bad = false;          // an emitter met an inconsistency: the finished plan gets ws_bytes = -1
  bool force_x3t_next = false;    // the next gemm() takes the operand-expansion form although its A operand is not a workspace tensor (text states)
  std::set<int64_t> pairs;        // workspace offsets that hold pair rows
  bool x3t_on(const Ref& a) const { return x3t && !x3t_hold && a.space == SP_WS; }

  // ---- op emitters ------------------------------------------------------------------------------
  void gemm(int64_t M, int N, int K, Ref a, Ref w, Ref bias, Ref out, int act_ = SDN_ACT_NONE, Ref residual = Ref(),
            int out_kind = SDN_OUT_BF16, int n_valid = 0, Ref a2 = Ref(), int K1 = 0, Ref rowbias = Ref(),
            int rows_per_batch = 0, int ld_rowbias = 0) {
    const bool forced = force_x3t_next && x3t && !x3t_hold;
    force_x3t_next = false;
    if ((x3t_on(a) || forced) && (act_ == SDN_ACT_NONE || act_ == SDN_ACT_GEGLU) && out_kind != SDN_OUT_F32_NCHW && n_valid == 0) {
      // A' = [hi | lo | hi] (written by the producing GroupNorm / LayerNorm / attention / GEGLU epilogue, or by a split pass),
      // W' = [hi | hi | lo]: one bf16 GEMM with three times the k loop; F32 residual, F32 (or, for GEGLU, triple) output
      Act tmp; Ref au = a;
      if (a.space != SP_WS || !tri.count(a.off)) { tmp = split3(a, a2, M, a2.space != SP_NONE ? K1 : K, a2.space != SP_NONE ? K - K1 : 0); au = R(tmp); }
      Op o; o.kind = OP_GEMM; memset(&o.gd, 0, sizeof(o.gd));
      o.x3t = 1;
      o.gd.M = (int)M; o.gd.N = N; o.gd.K = 3 * K; o.gd.a_mode = SDN_A_PLAIN; o.gd.act = act_; o.gd.out_kind = SDN_OUT_F32;
      const bool tri_o = triple_out_next && act_ == SDN_ACT_NONE;
      const bool pair_o = pair_out_next && act_ == SDN_ACT_NONE && !tri_o && residual.space == SP_NONE;
      triple_out_next = false; pair_out_next = false;
      o.gd.x3_out = act_ == SDN_ACT_GEGLU ? 2 : (tri_o ? 3 : (pair_o ? 4 : 1)); o.gd.rows_per_batch = rows_per_batch; o.gd.ld_rowbias = ld_rowbias;
      if (pair_o && out.space == SP_WS) pairs.insert(out.off);
      o.a = au; o.w = x3_weight(w, N, K, K); o.bias = bias; o.rowbias = rowbias; o.residual = residual; o.out = out;
      o.flops = 2.0 * (double)M * (double)N * (double)K;
      o.bytes = 6.0 * ((double)M * K + (double)N * K) + 4.0 * (double)M * (act_ == SDN_ACT_GEGLU ? 0.75 * N : N) +
                (residual.space != SP_NONE ? 4.0 * (double)M * N : 0.0);
      snprintf(o.label, sizeof(o.label), "k_gemm<%d>x3", sdn_gemm_pick_tile((int)M, N, 3 * K, act_));
      push_gemm(o);
      if ((act_ == SDN_ACT_GEGLU || tri_o) && out.space == SP_WS) tri.insert(out.off);
      if (tmp.off >= 0) drop(tmp);                            // stream order: the next op may reuse it
      return;
    }
    triple_out_next = false; pair_out_next = false;
    const bool rp = res_pre_next && residual.space != SP_NONE && act_ == SDN_ACT_NONE && out_kind == SDN_OUT_BF16 && n_valid == 0 && es == 2;
    res_pre_next = false;
    Op o; o.kind = OP_GEMM; memset(&o.gd, 0, sizeof(o.gd));
    o.gd.res_pre = rp ? 1 : 0;
    o.gd.M = (int)M; o.gd.N = N; o.gd.K = K; o.gd.a_mode = SDN_A_PLAIN; o.gd.K1 = K1; o.gd.act = act_;
    o.gd.out_kind = out_kind; o.gd.n_valid = n_valid; o.gd.rows_per_batch = rows_per_batch; o.gd.ld_rowbias = ld_rowbias;
    o.a = a; o.a2 = a2; o.w = w; o.bias = bias; o.rowbias = rowbias; o.residual = residual; o.out = out;
    o.flops = 2.0 * (double)M * (double)(n_valid > 0 ? n_valid : N) * (double)K;
    // algorithmic bytes: A + W + the output, + the residual operand when the epilogue adds one (it is read once, 16 bit)
    o.bytes = 2.0 * ((double)M * K + (double)N * K + (double)M * (act_ == SDN_ACT_GEGLU ? N / 2 : N)) +
              (residual.space != SP_NONE ? 2.0 * (double)M * N : 0.0);
    snprintf(o.label, sizeof(o.label), "k_gemm<%d>%s", sdn_gemm_pick_tile((int)M, N, K, act_, residual.space != SP_NONE && !rp), rp ? "/rp" : "");
    push_gemm(o);
  }
  // Small-M / long-K GEMMs (one-prompt batches) run in split-K form: the partial buffer lives only for this op.
  void push_gemm(Op& o) {
    o.col = pending_cols; pending_cols = Ref();
    const int nv = o.gd.n_valid > 0 ? o.gd.n_valid : o.gd.N;
    const int split = (!u->split_k || nv != o.gd.N) ? 1 : sdn_gemm_pick_split(o.gd.M, o.gd.N, o.gd.K, o.gd.act, o.gd.out_kind);
    if (split > 1) {
      const int64_t bytes = (int64_t)split * o.gd.M * o.gd.N * 4;
      const int64_t off = arena.alloc(bytes);
      o.gd.split_k = split; o.aux = Ref{SP_WS, off}; o.rows = bytes;
      arena.release(off, bytes);                              // stream order: the next op may reuse it
      const size_t L = strlen(o.label);
      if (L + 3 < sizeof(o.label)) snprintf(o.label + L, sizeof(o.label) - L, "/s%d", split);
    }
    plan->ops.push_back(o);
    plan->flops += o.flops;
  }
  void conv3x3(const Act& in, int cout, int n_pad, Ref w, Ref bias, Ref out, int stride, int upsample, Ref residual,
               Ref rowbias, int ld_rowbias, int out_kind = SDN_OUT_BF16, int n_valid = 0, int asym_pad = 0) {
    const int Hi = upsample ? in.side * 2 : in.side;
    const int Ho = (Hi + (asym_pad ? 1 : 2) - 3) / stride + 1;
    if (x3t_on(R(in))) {
      // the same convolution over an input with 3 Cin channels per pixel ([hi | lo | hi]) and per-tap weights [hi | hi | lo]
      Act tmp; Ref au = R(in);
      if (!tri.count(in.off)) { tmp = split3(R(in), Ref(), (int64_t)B * in.side * in.side, in.C, 0, in.hw, in.side); au = R(tmp); }
      Op o; o.kind = OP_GEMM; memset(&o.gd, 0, sizeof(o.gd));
      o.x3t = 1;
      o.gd.M = B * Ho * Ho; o.gd.N = n_pad; o.gd.K = 27 * in.C; o.gd.a_mode = SDN_A_CONV3X3;
      o.gd.Hs = in.side; o.gd.Ws = in.side; o.gd.Cin = 3 * in.C; o.gd.Ho = Ho; o.gd.Wo = Ho; o.gd.stride = stride;
      o.gd.upsample = upsample; o.gd.asym_pad = asym_pad; o.gd.n_valid = n_valid; o.gd.rows_per_batch = Ho * Ho; o.gd.ld_rowbias = ld_rowbias;
      if (out_kind == SDN_OUT_F32_NCHW) { o.gd.out_kind = out_kind; o.gd.x3_out = 0; }     // conv_out: the general epilogue's NCHW f32 form
      else { o.gd.out_kind = SDN_OUT_F32; o.gd.x3_out = 1; }
      o.a = au; o.w = x3_weight(w, n_pad, 9 * in.C, in.C); o.bias = bias; o.rowbias = rowbias; o.residual = residual; o.out = out;
      o.flops = 2.0 * (double)o.gd.M * (double)cout * 9.0 * in.C;
      o.bytes = 6.0 * ((double)B * in.side * in.side * in.C + (double)n_pad * 9 * in.C) + 4.0 * (double)o.gd.M * cout +
                (residual.space != SP_NONE ? 4.0 * (double)o.gd.M * cout : 0.0);
      if (Ho == in.side && sdn_conv_slab_shape_ok(o.gd.M, n_pad, 3 * in.C, in.side, stride, upsample, asym_pad, SDN_OUT_BF16, n_valid) &&
          sdn_gemm_pick_tile(o.gd.M, n_pad, o.gd.K, SDN_ACT_NONE) == 10)
        snprintf(o.label, sizeof(o.label), "k_conv_slab<%d>/x3", in.side);
      else
        snprintf(o.label, sizeof(o.label), "k_gemm<%d>x3", sdn_gemm_pick_tile(o.gd.M, n_pad, o.gd.K, SDN_ACT_NONE));
      push_gemm(o);
      if (tmp.off >= 0) drop(tmp);
      return;
    }
    Op o; o.kind = OP_GEMM; memset(&o.gd, 0, sizeof(o.gd));
    o.gd.res_pre = (res_pre_next && residual.space != SP_NONE && out_kind == SDN_OUT_BF16 && n_valid == 0 && es == 2) ? 1 : 0;
    res_pre_next = false;
    o.gd.M = B * Ho * Ho; o.gd.N = n_pad; o.gd.K = 9 * in.C; o.gd.a_mode = SDN_A_CONV3X3;
    o.gd.Hs = in.side; o.gd.Ws = in.side; o.gd.Cin = in.C; o.gd.Ho = Ho; o.gd.Wo = Ho; o.gd.stride = stride;
    o.gd.upsample = upsample; o.gd.asym_pad = asym_pad; o.gd.out_kind = out_kind; o.gd.n_valid = n_valid; o.gd.rows_per_batch = Ho * Ho;
    o.gd.ld_rowbias = ld_rowbias;
    o.a = R(in); o.w = w; o.bias = bias; o.rowbias = rowbias; o.residual = residual; o.out = out;
    o.flops = 2.0 * (double)o.gd.M * (double)cout * (double)o.gd.K;
    o.bytes = 2.0 * ((double)B * in.side * in.side * in.C + (double)n_pad * o.gd.K + (double)o.gd.M * cout) +
              (residual.space != SP_NONE ? 2.0 * (double)o.gd.M * cout : 0.0);      // + the residual map the epilogue adds
    if (Ho == in.side && sdn_conv_slab_shape_ok(o.gd.M, n_pad, in.C, in.side, stride, upsample, asym_pad, out_kind, n_valid) &&
        sdn_gemm_pick_tile(o.gd.M, n_pad, o.gd.K, SDN_ACT_NONE) == 10)
      snprintf(o.label, sizeof(o.label), "k_conv_slab<%d>", in.side);
    else
      snprintf(o.label, sizeof(o.label), "k_gemm<%d>", sdn_gemm_pick_tile(o.gd.M, n_pad, o.gd.K, SDN_ACT_NONE));
    push_gemm(o);
  }
  void groupnorm(const Act& x, const Act* x2, float eps, int silu, Ref gamma, Ref beta, const Act& out) {
    Op o; o.kind = OP_GN; o.a = R(x); if (x2) o.a2 = R(*x2);
    o.batch = B; o.hw = x.hw; o.c1 = x.C; o.c2 = x2 ? x2->C : 0; o.groups = u->cfg.norm_groups; o.eps = eps;
    o.silu = silu; o.w = gamma; o.bias = beta; o.out = R(out); o.aux = gn_stats;
    if (x3t) { o.tri_out = 1; tri.insert(out.off); }            // every GroupNorm of the UNet feeds a conv / linear
    if (x.st_off >= 0 && (!x2 || x2->st_off >= 0)) {             // statistics come with the inputs: apply pass only
      o.cols1 = Ref{SP_WS, x.st_off};
      if (x2) o.cols2 = Ref{SP_WS, x2->st_off};
    }
    o.bytes = 2.0 * (o.cols1.space != SP_NONE ? 2.0 : 3.0) * (double)B * x.hw * (o.c1 + o.c2);   // reads (stats?, apply) + one write
    snprintf(o.label, sizeof(o.label), o.cols1.space != SP_NONE ? "k_gn_apply" : "k_gn_stats+apply");
    plan->ops.push_back(o);
  }
  void layernorm(const Act& x, Ref gamma, Ref beta, const Act& out) {
    Op o; o.kind = OP_LN; o.a = R(x); o.rows = (int64_t)B * x.hw; o.c1 = x.C; o.eps = 1e-5f; o.w = gamma; o.bias = beta;
    o.out = R(out);
    if (x3t) { o.tri_out = 1; tri.insert(out.off); }            // ... and every LayerNorm a projection
    o.bytes = 2.0 * 2.0 * (double)o.rows * x.C;
    snprintf(o.label, sizeof(o.label), "k_layernorm");
    plan->ops.push_back(o);
  }
  // LayerNorm(x; gamma, beta) -> GEMM(W, bias) with the norm folded into the GEMM (sdn_gemm_ln_*).  Registers the derived
  // weight regions once; `prepass` = row statistics from a read-only pass instead of inside the kernel (wide N).
  Ref derived(const std::string& name, int64_t bytes) {
    auto it = u->param_index.find(name);
    if (it != u->param_index.end()) return Ref{SP_W, u->params[it->second].offset};
    sdn_param_info pi; memset(&pi, 0, sizeof(pi));
    snprintf(pi.name, sizeof(pi.name), "%s", name.c_str());
    pi.kind = SDN_P_DERIVED; pi.rows = (int)bytes; pi.cols = 0; pi.rows_padded = (int)bytes;
    pi.offset = u->weight_bytes;
    u->weight_bytes += (bytes + 255) & ~(int64_t)255;
    u->param_index[name] = (int)u->params.size();
    u->params.push_back(pi);
    return Ref{SP_W, pi.offset};
  }
  void gemm_ln(const Act& x, int64_t rows, int N, int K, const std::string& wname, Ref w, Ref gamma, Ref beta, Ref bias,
               Ref out, int act_, bool prepass) {
    const bool fresh = u->param_index.find(wname + "#ln") == u->param_index.end();
    Ref wf = derived(wname + "#ln", (int64_t)N * K * 2), c = derived(wname + "#ln_c", (int64_t)N * 4), d = derived(wname + "#ln_d", (int64_t)N * 4);
    if (fresh) u->fold_jobs.push_back({w.off, gamma.off, beta.off, bias.space == SP_NONE ? -1 : bias.off, wf.off, c.off, d.off, N, K});
    Act st;
    if (prepass) {
      st = act(rows, 2, 0, 0, 4);
      Op o; o.kind = OP_ROWSTATS; o.a = R(x); o.rows = rows; o.c1 = K; o.eps = 1e-5f; o.out = R(st);
      o.bytes = 2.0 * rows * K; snprintf(o.label, sizeof(o.label), "k_row_stats"); plan->ops.push_back(o);
    }
    Op o; o.kind = OP_GEMM; memset(&o.gd, 0, sizeof(o.gd));
    o.gd.M = (int)rows; o.gd.N = N; o.gd.K = K; o.gd.a_mode = SDN_A_PLAIN; o.gd.act = act_; o.gd.out_kind = SDN_OUT_BF16;
    o.a = R(x); o.w = wf; o.out = out; o.ln = 1; o.ln_c = c; o.ln_d = d; o.eps = 1e-5f;
    if (prepass) o.ln_stats = R(st);
    o.flops = 2.0 * (double)rows * N * K;
    o.bytes = 2.0 * ((double)rows * K + (double)N * K + (double)rows * (act_ == SDN_ACT_GEGLU ? N / 2 : N));
    { int t = sdn_gemm_pick_tile((int)rows, N, K, act_);
      if (t == 8 && N % 320 == 0) t = 10;                        // (sdn_gemm_impl: the folded forms have no 256-wide instantiation)
      snprintf(o.label, sizeof(o.label), "k_gemm<%d>/ln%d", t, prepass ? 2 : 1); }   // /lnL: L = the symbol's LNF
    plan->ops.push_back(o);
    plan->flops += o.flops;
    if (prepass) drop(st);
  }
  void repeat(const Act& in, const Act& out, int rep) {          // out = cat([in] * rep) along the batch
    Op o; o.kind = OP_REPEAT; o.a = R(in); o.out = R(out); o.rows = in.content; o.c1 = rep;     // (stream tensors: never triples)
    o.bytes = (double)in.content * (1 + rep);
    snprintf(o.label, sizeof(o.label), "k_repeat");
    plan->ops.push_back(o);
  }
  void attention(Ref q, Ref k, Ref v, Ref out, int nq, int nk, int C, int ldq, int ldk, int ldv, bool kv_pairs = false) {
    Op o; o.kind = OP_ATTN; o.a = q; o.k = k; o.v = v; o.out = out; o.batch = B; o.heads = u->cfg.n_heads;
    o.nq = nq; o.nk = nk; o.hd = C / u->cfg.n_heads; o.ldq = ldq; o.ldk = ldk; o.ldv = ldv; o.ldo = C;
    o.scale = 1.0f / sqrtf((float)o.hd);
    if (x3t && out.space == SP_WS) { o.tri_out = 1; tri.insert(out.off); }   // its only reader is the to_out projection
    if (q.space == SP_WS && pairs.count(q.off)) {               // 1: q / k / v = column blocks of ONE pair-row buffer (self-attention);
      o.pair_in = kv_pairs ? 2 : 1; pairs.erase(q.off);         // 2: q = [hi(C) | lo(C)], k / v = column blocks of the text projection's pair rows
    } else if (kv_pairs) {
      plan_bad = true;                                          // K / V were written as pairs but Q was not: refuse the plan (forward rejects it)
    }
    const double f = 4.0 * (double)B * o.heads * (double)nq * (double)nk * (double)o.hd;
    o.flops = f;
    o.bytes = 2.0 * (double)B * C * (2.0 * nq + 2.0 * nk);
    snprintf(o.label, sizeof(o.label), "k_attn<%d>", o.hd);
    plan->ops.push_back(o);
    plan->flops += f; plan->attn_flops += f;
  }

  // ---- blocks --------------------------------------------------------------------------------------
  // ResnetBlock2D: GN-SiLU-conv3x3(+temb) - GN-SiLU-conv3x3 + shortcut.  Input = x (++ skip).
  Act resnet(const std::string& pfx, Act& x, Act* skip, int cout) {
    const int cin = x.C + (skip ? skip->C : 0);
    Ref n1g = param(pfx + ".norm1.weight", SDN_P_VEC_F32, cin, 0), n1b = param(pfx + ".norm1.bias", SDN_P_VEC_F32, cin, 0);
    Ref c1w = param(pfx + ".conv1.weight", SDN_P_CONV3X3, cout, 9 * cin), c1b = param(pfx + ".conv1.bias", SDN_P_VEC_F32, cout, 0);
    const int tcol = tproj_cursor; tproj_cursor += cout;       // time_emb_proj registered up-front (stacked)
    Ref n2g = param(pfx + ".norm2.weight", SDN_P_VEC_F32, cout, 0), n2b = param(pfx + ".norm2.bias", SDN_P_VEC_F32, cout, 0);
    Ref c2w = param(pfx + ".conv2.weight", SDN_P_CONV3X3, cout, 9 * cout), c2b = param(pfx + ".conv2.bias", SDN_P_VEC_F32, cout, 0);
    const int64_t rows = (int64_t)B * x.hw;
    Act g1 = act(rows, cin, x.hw, x.side);
    groupnorm(x, skip, 1e-5f, 1, n1g, n1b, g1);
    Act h = act_gn(rows, cout, x.hw, x.side);
    want_stats(h);
    conv3x3(g1, cout, cout, c1w, c1b, R(h), 1, 0, Ref(), Ref{SP_WS, tproj.off + (int64_t)tcol * 4}, u->tproj_total);
    drop(g1);
    Act g2 = act(rows, cout, x.hw, x.side);
    groupnorm(h, nullptr, 1e-5f, 1, n2g, n2b, g2);
    drop(h);
    Act out = act_gn(rows, cout, x.hw, x.side);
    if (cin != cout) {
      Ref scw = param(pfx + ".conv_shortcut.weight", SDN_P_MAT, cout, cin), scb = param(pfx + ".conv_shortcut.bias", SDN_P_VEC_F32, cout, 0);
      Act sc = act(rows, cout, x.hw, x.side);
      gemm(rows, cout, cin, R(x), scw, scb, R(sc), SDN_ACT_NONE, Ref(), SDN_OUT_BF16, 0, skip ? R(*skip) : Ref(),
           skip ? x.C : 0);
      want_stats(out);
      res_pre_next = u->res_pre;
      conv3x3(g2, cout, cout, c2w, c2b, R(out), 1, 0, R(sc), Ref(), 0);
      drop(sc);
    } else {
      want_stats(out);
      res_pre_next = u->res_pre;
      conv3x3(g2, cout, cout, c2w, c2b, R(out), 1, 0, R(x), Ref(), 0);
    }
    drop(g2);
    return out;
  }

  // Transformer2DModel (continuous) with one BasicTransformerBlock.
  // Optional sub-batching of the transformer blocks (every op of a block is per-sample independent; results are
  // bit-identical).  It was built to test whether slicing the 168 MB activations of the 64x64 level keeps the chain of
  // short-K projections Infinity-Cache resident; it does not pay (see subbatch_bytes), so it is off by default.
  Act transformer(const std::string& pfx, Act& x) {
    const int64_t bytes_full = (int64_t)B * x.hw * x.C * es;
    int nsub = 1;
    if (u->subbatch_bytes > 0)
      while (nsub < B && bytes_full / nsub > u->subbatch_bytes && B % (nsub * 2) == 0) nsub *= 2;
    Act out = nsub == 1 ? act_gn((int64_t)B * x.hw, x.C, x.hw, x.side) : act((int64_t)B * x.hw, x.C, x.hw, x.side);
    const int Bfull = B, Bs = B / nsub;
    for (int sb = 0; sb < nsub; ++sb) {
      B = Bs;
      Act xs = x, os = out;                                   // views: never dropped
      xs.off += (int64_t)sb * Bs * x.hw * x.C * es; os.off += (int64_t)sb * Bs * x.hw * x.C * es;
      transformer_body(pfx, xs, os, (int64_t)sb * Bs * u->cfg.text_len * u->cfg.cross_dim * es);
    }
    B = Bfull;
    return out;
  }

  // rep > 1 (sdn_unet_config.latent_repeat, first block only): `x` holds the B / rep samples the guidance branches share
  // and `x_full` their repetition; everything up to the cross-attention's query is computed once and repeated.
  void transformer_body(const std::string& pfx, Act& x, const Act& out, int64_t text_off, int rep = 1,
                        const Act* x_full = nullptr) {
    const int C = x.C, hw = x.hw, T = u->cfg.text_len, X = u->cfg.cross_dim;
    const int Bfull = B;
    B = Bfull / rep;
    int64_t rows = (int64_t)B * hw;
    const std::string tb = pfx + ".transformer_blocks.0";
    Ref ng = param(pfx + ".norm.weight", SDN_P_VEC_F32, C, 0), nb = param(pfx + ".norm.bias", SDN_P_VEC_F32, C, 0);
    Ref piw = param(pfx + ".proj_in.weight", SDN_P_MAT, C, C), pib = param(pfx + ".proj_in.bias", SDN_P_VEC_F32, C, 0);
    Ref l1g = param(tb + ".norm1.weight", SDN_P_VEC_F32, C, 0), l1b = param(tb + ".norm1.bias", SDN_P_VEC_F32, C, 0);
    Ref qkv = stacked({tb + ".attn1.to_q.weight", tb + ".attn1.to_k.weight", tb + ".attn1.to_v.weight"}, C, C);
    Ref o1w = param(tb + ".attn1.to_out.0.weight", SDN_P_MAT, C, C), o1b = param(tb + ".attn1.to_out.0.bias", SDN_P_VEC_F32, C, 0);
    Ref l2g = param(tb + ".norm2.weight", SDN_P_VEC_F32, C, 0), l2b = param(tb + ".norm2.bias", SDN_P_VEC_F32, C, 0);
    Ref q2w = param(tb + ".attn2.to_q.weight", SDN_P_MAT, C, C);
    Ref kv2 = stacked({tb + ".attn2.to_k.weight", tb + ".attn2.to_v.weight"}, C, X);
    Ref o2w = param(tb + ".attn2.to_out.0.weight", SDN_P_MAT, C, C), o2b = param(tb + ".attn2.to_out.0.bias", SDN_P_VEC_F32, C, 0);
    Ref l3g = param(tb + ".norm3.weight", SDN_P_VEC_F32, C, 0), l3b = param(tb + ".norm3.bias", SDN_P_VEC_F32, C, 0);
    Ref f1w = param(tb + ".ff.net.0.proj.weight", SDN_P_GEGLU_MAT, 8 * C, C), f1b = param(tb + ".ff.net.0.proj.bias", SDN_P_GEGLU_VEC, 8 * C, 0);
    Ref f2w = param(tb + ".ff.net.2.weight", SDN_P_MAT, C, 4 * C), f2b = param(tb + ".ff.net.2.bias", SDN_P_VEC_F32, C, 0);
    Ref pow_ = param(pfx + ".proj_out.weight", SDN_P_MAT, C, C), pob = param(pfx + ".proj_out.bias", SDN_P_VEC_F32, C, 0);

    Act gn = act(rows, C, hw, x.side);
    groupnorm(x, nullptr, 1e-6f, 0, ng, nb, gn);
    Act h = act(rows, C, hw, x.side);
    gemm(rows, C, C, R(gn), piw, pib, R(h));
    drop(gn);
    // LayerNorm folding (gemm_ln): measured per shape (tools/bench_lnfold.py) -- it pays at C = 320 / 640 for the
    // attention projections (statistics inside the kernel while N <= 960, from a read-only pre-pass above) and at
    // C = 320 for the GEGLU projection; the wide, MFMA-bound projections of the lower levels keep the LayerNorm kernel.
    const bool fold12 = u->ln_fold && C <= 640, fold3 = u->ln_fold && C == 320;
    const int hdx = C / u->cfg.n_heads;
    const bool x3p_cross = x3t && u->x3_pairs && u->subbatch_bytes == 0 && (hdx == 40 || hdx == 80 || hdx == 160);   // sdn_attention_x3_pairs on the cross-attention too
    // self-attention
    Act ln;
    if (!fold12 || !fold3) ln = act(rows, C, hw, x.side);
    Act qkvb = act(rows, 3 * C, hw, x.side);
    if (fold12) {
      gemm_ln(h, rows, 3 * C, C, tb + ".attn1.to_q.weight", qkv, l1g, l1b, Ref(), R(qkvb), SDN_ACT_NONE, 3 * C > 960 || u->ln_prepass_all);
    } else {
      layernorm(h, l1g, l1b, ln);
      const int hd1 = C / u->cfg.n_heads;
      if (x3t && u->x3_pairs && (hd1 == 40 || hd1 == 80 || hd1 == 160) && (int64_t)hw * 6 * C * 2 < (1LL << 31)) pair_out_next = true;
      gemm(rows, 3 * C, C, R(ln), qkv, Ref(), R(qkvb));
    }
    Act at = act(rows, C, hw, x.side);
    attention(R(qkvb), Ref{SP_WS, qkvb.off + (int64_t)C * es}, Ref{SP_WS, qkvb.off + (int64_t)2 * C * es}, R(at), hw, hw, C,
              3 * C, 3 * C, 3 * C);
    drop(qkvb);
    Act h2 = act(rows, C, hw, x.side);
    res_pre_next = u->res_pre;
    gemm(rows, C, C, R(at), o1w, o1b, R(h2), SDN_ACT_NONE, R(h));
    drop(h);
    // cross-attention
    Act qb = act(rows, C, hw, x.side);
    if (fold12) {
      gemm_ln(h2, rows, C, C, tb + ".attn2.to_q.weight", q2w, l2g, l2b, Ref(), R(qb), SDN_ACT_NONE, u->ln_prepass_all != 0);
    } else {
      layernorm(h2, l2g, l2b, ln);
      if (x3p_cross) pair_out_next = true;
      gemm(rows, C, C, R(ln), q2w, Ref(), R(qb));
    }
    if (rep > 1) {                                 // from here on the branches differ (their text does)
      if (ln.off >= 0) drop(ln);
      drop(at);
      B = Bfull; rows = (int64_t)B * hw;
      Act h2f = act(rows, C, hw, x.side), qbf = act(rows, C, hw, x.side);
      repeat(h2, h2f, rep); repeat(qb, qbf, rep);
      const bool q_is_pairs = pairs.count(qb.off) > 0;         // (a byte-wise copy: pair rows stay pair rows)
      drop(h2); drop(qb);
      h2 = h2f; qb = qbf;
      if (q_is_pairs) pairs.insert(qb.off);
      if (!fold3) ln = act(rows, C, hw, x.side);
      at = act(rows, C, hw, x.side);
    }
    if (u->subbatch_bytes > 0) {
      Act kvb = act((int64_t)B * T, 2 * C);
      gemm((int64_t)B * T, 2 * C, X, Ref{SP_TEXT, text_off}, kv2, Ref(), R(kvb));
      attention(R(qb), R(kvb), Ref{SP_WS, kvb.off + (int64_t)C * es}, R(at), hw, T, C, C, 2 * C, 2 * C);
      drop(qb); drop(kvb);
    } else {
      // the text's keys / values live in the workspace's persistent tail: no other op ever writes there, so a forward that is
      // handed the same text version can skip this projection and read the previous forward's output (sdn_unet::text_version)
      const Ref kvr{SP_KV, kv_top};
      kv_top += Arena::up((int64_t)B * T * 2 * C * es);
      if (x3p_cross) { force_x3t_next = true; pair_out_next = true; }       // text K / V as pair rows [hi(2C) | lo(2C)] (same bytes as f32)
      gemm((int64_t)B * T, 2 * C, X, Ref{SP_TEXT, text_off}, kv2, Ref(), kvr);
      plan->ops.back().text_kv = 1;
      attention(R(qb), kvr, Ref{SP_KV, kvr.off + (int64_t)C * es}, R(at), hw, T, C, C, 2 * C, 2 * C, x3p_cross);
      drop(qb);
    }
    Act h3 = act(rows, C, hw, x.side);
    res_pre_next = u->res_pre;
    gemm(rows, C, C, R(at), o2w, o2b, R(h3), SDN_ACT_NONE, R(h2));
    drop(h2); drop(at);
    // GEGLU feed-forward
    if (fold3 && u->ff_fuse && u->ffn_fuse && C == 320) {
      // norm3 -> GEGLU projection -> [ff | h3] . [Wpo W2 | Wpo]^T + residual as ONE launch (sdn_ffn.hip): the [rows, 4C] hidden
      // activation stays in LDS.  Same derived weights as the two launches below, same bits.
      if (ln.off >= 0) drop(ln);
      const std::string wname = tb + ".ff.net.0.proj.weight";
      const bool fresh1 = u->param_index.find(wname + "#ln") == u->param_index.end();
      Ref wf = derived(wname + "#ln", (int64_t)8 * C * C * 2), c1 = derived(wname + "#ln_c", (int64_t)8 * C * 4), d1 = derived(wname + "#ln_d", (int64_t)8 * C * 4);
      if (fresh1) u->fold_jobs.push_back({f1w.off, l3g.off, l3b.off, f1b.off, wf.off, c1.off, d1.off, 8 * C, C});
      const bool fresh2 = u->param_index.find(pfx + ".proj_out.weight#ff") == u->param_index.end();
      Ref wcat = derived(pfx + ".proj_out.weight#ff", (int64_t)C * 5 * C * 2), bcat = derived(pfx + ".proj_out.bias#ff", (int64_t)C * 4);
      if (fresh2) { sdn_unet::FoldJob j{f2w.off, pow_.off, f2b.off, pob.off, wcat.off, bcat.off, -1, C, 4 * C}; j.kind = 1; u->fold_jobs.push_back(j); }
      // norm3's row statistics: from the operand fragments inside k_ffn320 (ffn_own_stats), or by a read-only pre-pass over h3
      Act st;
      if (!u->ffn_own_stats) {
        st = act(rows, 2, 0, 0, 4);
        Op o; o.kind = OP_ROWSTATS; o.a = R(h3); o.rows = rows; o.c1 = C; o.eps = 1e-5f; o.out = R(st);
        o.bytes = 2.0 * rows * C; snprintf(o.label, sizeof(o.label), "k_row_stats"); plan->ops.push_back(o);
      }
      want_stats(out);
      Op o; o.kind = OP_FFN; o.a = R(h3); if (st.off >= 0) o.ln_stats = R(st); o.w = wf; o.ln_c = c1; o.ln_d = d1; o.a2 = wcat; o.bias = bcat;
      o.residual = R(rep > 1 ? *x_full : x); o.out = R(out); o.rows = rows; o.c1 = C;
      o.col = pending_cols; pending_cols = Ref();
      o.flops = 2.0 * (double)rows * ((double)8 * C * C + (double)C * 5 * C);
      o.bytes = 2.0 * ((double)rows * C * 3 + (double)8 * C * C + (double)5 * C * C);
      snprintf(o.label, sizeof(o.label), "k_ffn320");
      plan->ops.push_back(o);
      plan->flops += o.flops;
      if (st.off >= 0) drop(st);
      drop(h3);
      return;
    }
    Act ff = act(rows, 4 * C, hw, x.side);
    if (fold3) {
      // (the two-launch form of the C = 320 feed-forward is the fallback / equality partner of k_ffn320: it takes its statistics
      //  the way that kernel does -- from the fragments when ffn_own_stats, else from the pre-pass)
      gemm_ln(h3, rows, 8 * C, C, tb + ".ff.net.0.proj.weight", f1w, l3g, l3b, f1b, R(ff), SDN_ACT_GEGLU, !u->ffn_own_stats);
    } else {
      layernorm(h3, l3g, l3b, ln);
      gemm(rows, 8 * C, C, R(ln), f1w, f1b, R(ff), SDN_ACT_GEGLU);
    }
    if (ln.off >= 0) drop(ln);
    if (u->ff_fuse) {
      // out = x + Wpo (h3 + W2 ff + b2) + bpo = x + [ff | h3] . [Wpo W2 | Wpo]^T + (Wpo b2 + bpo): the FeedForward output linear
      // and proj_out are one GEMM (two-source A operand, K = 5C); the [M, C] tensor between them is never written or re-read
      // and the worst-shaped launch of the block (M x C x C) disappears.  The product weight is derived once per weight set.
      const bool fresh = u->param_index.find(pfx + ".proj_out.weight#ff") == u->param_index.end();
      Ref wcat = derived(pfx + ".proj_out.weight#ff", (int64_t)C * 5 * C * 2), bcat = derived(pfx + ".proj_out.bias#ff", (int64_t)C * 4);
      if (fresh) { sdn_unet::FoldJob j{f2w.off, pow_.off, f2b.off, pob.off, wcat.off, bcat.off, -1, C, 4 * C}; j.kind = 1; u->fold_jobs.push_back(j); }
      want_stats(out);
      res_pre_next = u->res_pre && C != 320;       // (C = 320 must keep the bits of the one-launch k_ffn320, which adds it in its epilogue)
      gemm(rows, C, 5 * C, R(ff), wcat, bcat, R(out), SDN_ACT_NONE, R(rep > 1 ? *x_full : x), SDN_OUT_BF16, 0, R(h3), 4 * C);
      drop(ff); drop(h3);
      return;
    }
    Act h4 = act(rows, C, hw, x.side);
    triple_out_next = x3t;                                  // (bf16x3 plan: proj_out is its only reader)
    gemm(rows, C, 4 * C, R(ff), f2w, f2b, R(h4), SDN_ACT_NONE, R(h3));
    drop(ff); drop(h3);
    want_stats(out);
    gemm(rows, C, C, R(h4), pow_, pob, R(out), SDN_ACT_NONE, R(rep > 1 ? *x_full : x));
    drop(h4);
  }


  // =====================================================================================================
  //  SD-v3 MMDiT (SD3Transformer2DModel, diffusers 0.29.0; row U7) -- reached through self.transformer(...) at
  //  models/sdv3/safe_denoiser_pipeline.py:1120-1127.  Two token streams (image, text) with adaLN-zero
  //  modulation from (timestep, pooled text); joint attention over both streams without concatenating them.
  // =====================================================================================================
  void gemm_ex(int64_t M, int N, int K, Ref a, Ref w, Ref bias, Ref out, int act_, Ref residual, int out_kind,
               Ref rowbias, Ref rowgate, int rows_per_batch, int ld_row, int residual_bcast = 0) {
    Op o; o.kind = OP_GEMM; memset(&o.gd, 0, sizeof(o.gd));
    o.gd.M = (int)M; o.gd.N = N; o.gd.K = K; o.gd.a_mode = SDN_A_PLAIN; o.gd.act = act_; o.gd.out_kind = out_kind;
    o.gd.rows_per_batch = rows_per_batch; o.gd.ld_rowbias = ld_row; o.gd.ld_rowgate = ld_row;
    o.gd.residual_bcast = residual_bcast;
    o.a = a; o.w = w; o.bias = bias; o.rowbias = rowbias; o.rowgate = rowgate; o.residual = residual; o.out = out;
    o.flops = 2.0 * (double)M * N * K;
    o.bytes = 2.0 * ((double)M * K + (double)N * K + (double)M * N);
    snprintf(o.label, sizeof(o.label), "k_gemm<%d>", sdn_gemm_pick_tile((int)M, N, K, act_, residual.space != SP_NONE || rowgate.space != SP_NONE));
    push_gemm(o);
  }
  void ln_mod(const Act& x, int64_t rows, int rows_per_batch, Ref scale, Ref shift, int ld, const Act& out) {
    Op o; o.kind = OP_LN; o.a = R(x); o.rows = rows; o.c1 = x.C; o.eps = 1e-6f; o.w = scale; o.bias = shift; o.out = R(out);
    o.mod = 1; o.ld_mod = ld; o.hw = rows_per_batch;
    o.bytes = 2.0 * 2.0 * (double)rows * x.C;
    snprintf(o.label, sizeof(o.label), "k_layernorm");
    plan->ops.push_back(o);
  }
  Ref fcol(int col) const { return Ref{SP_WS, tproj.off + (int64_t)col * 4}; }   // column of the stacked adaLN output

  void build_mmdit() {
    const sdn_mmdit_config& c = u->mcfg;
    const int C = c.num_heads * c.head_dim, S = c.sample_size, ps = c.patch_size, hp = S / ps, N = hp * hp;
    const int T = c.text_len, L = c.num_layers, KP = c.in_channels * ps * ps;
    char buf[128];
    auto nm = [&](int i, const char* suffix) { snprintf(buf, sizeof(buf), "transformer_blocks.%d.%s", i, suffix); return std::string(buf); };

    // ---- parameters of the conditioning path ----
    Ref pew = param("pos_embed.proj.weight", SDN_P_MAT, C, KP), peb = param("pos_embed.proj.bias", SDN_P_VEC_F32, C, 0);
    Ref pos = param("pos_embed.pos_embed", SDN_P_POS_CROP, N, C);
    Ref t1w = param("time_text_embed.timestep_embedder.linear_1.weight", SDN_P_MAT, C, c.time_dim), t1b = param("time_text_embed.timestep_embedder.linear_1.bias", SDN_P_VEC_F32, C, 0);
    Ref t2w = param("time_text_embed.timestep_embedder.linear_2.weight", SDN_P_MAT, C, C), t2b = param("time_text_embed.timestep_embedder.linear_2.bias", SDN_P_VEC_F32, C, 0);
    Ref p1w = param("time_text_embed.text_embedder.linear_1.weight", SDN_P_MAT, C, c.pooled_dim), p1b = param("time_text_embed.text_embedder.linear_1.bias", SDN_P_VEC_F32, C, 0);
    Ref p2w = param("time_text_embed.text_embedder.linear_2.weight", SDN_P_MAT, C, C), p2b = param("time_text_embed.text_embedder.linear_2.bias", SDN_P_VEC_F32, C, 0);
    Ref cew = param("context_embedder.weight", SDN_P_MAT, C, c.joint_dim), ceb = param("context_embedder.bias", SDN_P_VEC_F32, C, 0);

    // ---- ALL adaLN linears stacked into one [sum, C] matrix: one GEMM per forward ----
    // column layout per block i: img 6C at col_img[i], ctx 6C (2C for the last, context_pre_only block) at col_ctx[i]
    std::vector<int> col_img(L), col_ctx(L);
    int total = 0;
    Ref adw, adb;
    {
      std::vector<std::pair<std::string, int>> mods;
      for (int i = 0; i < L; ++i) {
        col_img[i] = total; mods.push_back({nm(i, "norm1.linear"), 6 * C}); total += 6 * C;
        const int nc = (i == L - 1) ? 2 * C : 6 * C;
        col_ctx[i] = total; mods.push_back({nm(i, "norm1_context.linear"), nc}); total += nc;
      }
      const int col_out = total; mods.push_back({"norm_out.linear", 2 * C}); total += 2 * C;
      (void)col_out;
      int64_t expect = -1;
      for (size_t j = 0; j < mods.size(); ++j) {
        Ref r = param(mods[j].first + ".weight", SDN_P_MAT, mods[j].second, C);
        if (j == 0) adw = r; else if (r.off != expect) { fprintf(stderr, "libsdn: adaLN weights not contiguous\n"); abort(); }
        expect = r.off + (int64_t)mods[j].second * C * 2;
      }
      expect = -1;
      for (size_t j = 0; j < mods.size(); ++j) {
        Ref r = param(mods[j].first + ".bias", SDN_P_VEC_F32, mods[j].second, 0);
        if (j == 0) adb = r; else if (r.off != expect) { fprintf(stderr, "libsdn: adaLN biases not contiguous\n"); abort(); }
        expect = r.off + (int64_t)mods[j].second * 4;
      }
    }
    u->tproj_total = total;

    // ---- conditioning: silu(time_emb + pooled_emb) -> all modulation vectors ----
    plan->tscalar_off = arena.alloc(256);
    Act tsin = act(B, c.time_dim);
    { Op o; o.kind = OP_TEMB; o.batch = B; o.c1 = c.time_dim; o.out = R(tsin); snprintf(o.label, sizeof(o.label), "k_temb"); plan->ops.push_back(o); }
    Act th = act(B, C);
    gemm_ex(B, C, c.time_dim, R(tsin), t1w, t1b, R(th), SDN_ACT_SILU, Ref(), SDN_OUT_BF16, Ref(), Ref(), 0, 0);
    drop(tsin);
    Act temb = act(B, C, 0, 0, 4);                                                 // f32 [B, C]
    gemm_ex(B, C, C, R(th), t2w, t2b, R(temb), SDN_ACT_NONE, Ref(), SDN_OUT_F32, Ref(), Ref(), 0, 0);
    drop(th);
    Act ph = act(B, C);
    gemm_ex(B, C, c.pooled_dim, Ref{SP_POOLED, 0}, p1w, p1b, R(ph), SDN_ACT_SILU, Ref(), SDN_OUT_BF16, Ref(), Ref(), 0, 0);
    Act scond = act(B, C);                                                         // silu(time_emb + pooled_emb)
    gemm_ex(B, C, C, R(ph), p2w, p2b, R(scond), SDN_ACT_SILU, Ref(), SDN_OUT_BF16, R(temb), Ref(), 1, C);
    drop(ph); drop(temb);
    Act mods = act(B, total, 0, 0, 4);
    tproj = R(mods);
    gemm_ex(B, total, C, R(scond), adw, adb, R(mods), SDN_ACT_NONE, Ref(), SDN_OUT_F32, Ref(), Ref(), 0, 0);
    drop(scond);

    // ---- token streams ----
    Act patches = act((int64_t)B * N, KP);
    { Op o; o.kind = OP_PATCHIFY; o.batch = B; o.c1 = c.in_channels; o.hw = S; o.patch = ps; o.a = Ref{SP_LATENTS, 0}; o.out = R(patches);
      o.bytes = (double)B * N * KP * 6.0; snprintf(o.label, sizeof(o.label), "k_patchify"); plan->ops.push_back(o); }
    Act x = act((int64_t)B * N, C, N);
    gemm_ex((int64_t)B * N, C, KP, R(patches), pew, peb, R(x), SDN_ACT_NONE, pos, SDN_OUT_BF16, Ref(), Ref(), N, 0, 1);
    drop(patches);
    Act ctx = act((int64_t)B * T, C, T);
    gemm_ex((int64_t)B * T, C, c.joint_dim, Ref{SP_TEXT, 0}, cew, ceb, R(ctx), SDN_ACT_NONE, Ref(), SDN_OUT_BF16, Ref(), Ref(), 0, 0);

    for (int i = 0; i < L; ++i) {
      const bool last = (i == L - 1);
      const int ci = col_img[i], cc = col_ctx[i];
      // AdaLayerNormZero chunk order: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
      // AdaLayerNormContinuous (last block's context): scale, shift
      Ref qkvw = stacked({nm(i, "attn.to_q.weight"), nm(i, "attn.to_k.weight"), nm(i, "attn.to_v.weight")}, C, C);
      Ref qkvb; { Ref r0 = param(nm(i, "attn.to_q.bias"), SDN_P_VEC_F32, C, 0); param(nm(i, "attn.to_k.bias"), SDN_P_VEC_F32, C, 0); param(nm(i, "attn.to_v.bias"), SDN_P_VEC_F32, C, 0); qkvb = r0; }
      Ref aqkvw = stacked({nm(i, "attn.add_q_proj.weight"), nm(i, "attn.add_k_proj.weight"), nm(i, "attn.add_v_proj.weight")}, C, C);
      Ref aqkvb; { Ref r0 = param(nm(i, "attn.add_q_proj.bias"), SDN_P_VEC_F32, C, 0); param(nm(i, "attn.add_k_proj.bias"), SDN_P_VEC_F32, C, 0); param(nm(i, "attn.add_v_proj.bias"), SDN_P_VEC_F32, C, 0); aqkvb = r0; }
      Ref ow = param(nm(i, "attn.to_out.0.weight"), SDN_P_MAT, C, C), ob = param(nm(i, "attn.to_out.0.bias"), SDN_P_VEC_F32, C, 0);
      Ref f1w = param(nm(i, "ff.net.0.proj.weight"), SDN_P_MAT, 4 * C, C), f1b = param(nm(i, "ff.net.0.proj.bias"), SDN_P_VEC_F32, 4 * C, 0);
      Ref f2w = param(nm(i, "ff.net.2.weight"), SDN_P_MAT, C, 4 * C), f2b = param(nm(i, "ff.net.2.bias"), SDN_P_VEC_F32, C, 0);

      Act xn = act((int64_t)B * N, C, N);
      ln_mod(x, (int64_t)B * N, N, fcol(ci + 1 * C), fcol(ci + 0 * C), total, xn);
      Act cn = act((int64_t)B * T, C, T);
      if (last) ln_mod(ctx, (int64_t)B * T, T, fcol(cc + 0 * C), fcol(cc + 1 * C), total, cn);
      else ln_mod(ctx, (int64_t)B * T, T, fcol(cc + 1 * C), fcol(cc + 0 * C), total, cn);
      Act qx = act((int64_t)B * N, 3 * C, N), qc = act((int64_t)B * T, 3 * C, T);
      gemm_ex((int64_t)B * N, 3 * C, C, R(xn), qkvw, qkvb, R(qx), SDN_ACT_NONE, Ref(), SDN_OUT_BF16, Ref(), Ref(), 0, 0);
      gemm_ex((int64_t)B * T, 3 * C, C, R(cn), aqkvw, aqkvb, R(qc), SDN_ACT_NONE, Ref(), SDN_OUT_BF16, Ref(), Ref(), 0, 0);
      drop(xn); drop(cn);
      Act ax = act((int64_t)B * N, C, N), ac = act((int64_t)B * T, C, T);
      {
        Op o; o.kind = OP_ATTN; o.batch = B; o.heads = c.num_heads; o.nq = N + T; o.nk = N + T; o.hd = c.head_dim;
        o.a = R(qx); o.k = Ref{SP_WS, qx.off + (int64_t)C * 2}; o.v = Ref{SP_WS, qx.off + (int64_t)2 * C * 2}; o.out = R(ax);
        o.q2 = R(qc); o.k2 = Ref{SP_WS, qc.off + (int64_t)C * 2}; o.v2 = Ref{SP_WS, qc.off + (int64_t)2 * C * 2}; o.out2 = R(ac);
        o.n1 = N; o.ldq = o.ldk = o.ldv = 3 * C; o.ldo = C; o.scale = 1.0f / sqrtf((float)c.head_dim);
        o.flops = 4.0 * (double)B * o.heads * (double)(N + T) * (double)(N + T) * o.hd;
        o.bytes = 2.0 * (double)B * (N + T) * C * 4.0;
        snprintf(o.label, sizeof(o.label), "k_attn<%d>", o.hd);
        plan->ops.push_back(o);
        plan->flops += o.flops; plan->attn_flops += o.flops;
      }
      drop(qx); drop(qc);
      // image stream: x += gate_msa * to_out(attn);  x += gate_mlp * ff(LNmod(x))
      Act x2 = act((int64_t)B * N, C, N);
      gemm_ex((int64_t)B * N, C, C, R(ax), ow, ob, R(x2), SDN_ACT_NONE, R(x), SDN_OUT_BF16, Ref(), fcol(ci + 2 * C), N, total);
      drop(ax); drop(x);
      Act xm = act((int64_t)B * N, C, N);
      ln_mod(x2, (int64_t)B * N, N, fcol(ci + 4 * C), fcol(ci + 3 * C), total, xm);
      Act hx = act((int64_t)B * N, 4 * C, N);
      gemm_ex((int64_t)B * N, 4 * C, C, R(xm), f1w, f1b, R(hx), SDN_ACT_GELU_TANH, Ref(), SDN_OUT_BF16, Ref(), Ref(), 0, 0);
      drop(xm);
      Act x3 = act((int64_t)B * N, C, N);
      gemm_ex((int64_t)B * N, C, 4 * C, R(hx), f2w, f2b, R(x3), SDN_ACT_NONE, R(x2), SDN_OUT_BF16, Ref(), fcol(ci + 5 * C), N, total);
      drop(hx); drop(x2);
      x = x3;
      // text stream (skipped in the last block: context_pre_only)
      if (!last) {
        Ref aow = param(nm(i, "attn.to_add_out.weight"), SDN_P_MAT, C, C), aob = param(nm(i, "attn.to_add_out.bias"), SDN_P_VEC_F32, C, 0);
        Ref g1w = param(nm(i, "ff_context.net.0.proj.weight"), SDN_P_MAT, 4 * C, C), g1b = param(nm(i, "ff_context.net.0.proj.bias"), SDN_P_VEC_F32, 4 * C, 0);
        Ref g2w = param(nm(i, "ff_context.net.2.weight"), SDN_P_MAT, C, 4 * C), g2b = param(nm(i, "ff_context.net.2.bias"), SDN_P_VEC_F32, C, 0);
        Act c2 = act((int64_t)B * T, C, T);
        gemm_ex((int64_t)B * T, C, C, R(ac), aow, aob, R(c2), SDN_ACT_NONE, R(ctx), SDN_OUT_BF16, Ref(), fcol(cc + 2 * C), T, total);
        drop(ctx);
        Act cm = act((int64_t)B * T, C, T);
        ln_mod(c2, (int64_t)B * T, T, fcol(cc + 4 * C), fcol(cc + 3 * C), total, cm);
        Act hc = act((int64_t)B * T, 4 * C, T);
        gemm_ex((int64_t)B * T, 4 * C, C, R(cm), g1w, g1b, R(hc), SDN_ACT_GELU_TANH, Ref(), SDN_OUT_BF16, Ref(), Ref(), 0, 0);
        drop(cm);
        Act c3 = act((int64_t)B * T, C, T);
        gemm_ex((int64_t)B * T, C, 4 * C, R(hc), g2w, g2b, R(c3), SDN_ACT_NONE, R(c2), SDN_OUT_BF16, Ref(), fcol(cc + 5 * C), T, total);
        drop(hc); drop(c2);
        ctx = c3;
      } else {
        drop(ctx);
      }
      drop(ac);
    }
    // ---- norm_out (AdaLayerNormContinuous: scale, shift) + proj_out + unpatchify ----
    const int col_out = total - 2 * C;
    Act xo = act((int64_t)B * N, C, N);
    ln_mod(x, (int64_t)B * N, N, fcol(col_out), fcol(col_out + C), total, xo);
    drop(x);
    const int PO = ps * ps * c.out_channels;
    Ref pw = param("proj_out.weight", SDN_P_MAT, PO, C), pb = param("proj_out.bias", SDN_P_VEC_F32, PO, 0);
    Act tok = act((int64_t)B * N, PO, N, 0, 4);
    gemm_ex((int64_t)B * N, PO, C, R(xo), pw, pb, R(tok), SDN_ACT_NONE, Ref(), SDN_OUT_F32, Ref(), Ref(), 0, 0);
    drop(xo);
    { Op o; o.kind = OP_UNPATCHIFY; o.batch = B; o.c1 = c.out_channels; o.hw = S; o.patch = ps; o.a = R(tok); o.out = Ref{SP_OUT, 0};
      o.bytes = (double)B * N * PO * 8.0; snprintf(o.label, sizeof(o.label), "k_unpatchify"); plan->ops.push_back(o); }
    drop(tok);
    drop(mods);
    plan->ws_bytes = arena.peak;
  }

  struct Res { std::string pfx; int cout; };
  // Walk the architecture once to list every resnet (execution order) -> stacked time_emb_proj.
  std::vector<Res> enumerate_resnets() const {
    const sdn_unet_config& c = u->cfg;
    std::vector<Res> v;
    char buf[96];
    for (int i = 0; i < c.n_levels; ++i)
      for (int j = 0; j < c.layers_per_block; ++j) {
        snprintf(buf, sizeof(buf), "down_blocks.%d.resnets.%d", i, j);
        v.push_back({buf, c.block_out_channels[i]});
      }
    const int top = c.block_out_channels[c.n_levels - 1];
    v.push_back({"mid_block.resnets.0", top});
    v.push_back({"mid_block.resnets.1", top});
    for (int i = 0; i < c.n_levels; ++i)
      for (int j = 0; j <= c.layers_per_block; ++j) {
        snprintf(buf, sizeof(buf), "up_blocks.%d.resnets.%d", i, j);
        v.push_back({buf, c.block_out_channels[c.n_levels - 1 - i]});
      }
    return v;
  }

  void build() {
    const sdn_unet_config& c = u->cfg;
    const int S = c.sample_size, ch0 = c.block_out_channels[0], tdim = 4 * ch0;
    char buf[96];

    // ---- time embedding + stacked time_emb_proj ----
    Ref l1w = param("time_embedding.linear_1.weight", SDN_P_MAT, tdim, ch0), l1b = param("time_embedding.linear_1.bias", SDN_P_VEC_F32, tdim, 0);
    Ref l2w = param("time_embedding.linear_2.weight", SDN_P_MAT, tdim, tdim), l2b = param("time_embedding.linear_2.bias", SDN_P_VEC_F32, tdim, 0);
    const std::vector<Res> rs = enumerate_resnets();
    int total = 0;
    Ref tpw, tpb;
    {
      int64_t expect = -1;
      for (size_t i = 0; i < rs.size(); ++i) {
        Ref r = param(rs[i].pfx + ".time_emb_proj.weight", SDN_P_MAT, rs[i].cout, tdim);
        if (i == 0) tpw = r; else if (r.off != expect) { fprintf(stderr, "libsdn: time_emb_proj not contiguous\n"); abort(); }
        expect = r.off + (int64_t)rs[i].cout * tdim * es;
        total += rs[i].cout;
      }
      expect = -1;
      for (size_t i = 0; i < rs.size(); ++i) {
        Ref r = param(rs[i].pfx + ".time_emb_proj.bias", SDN_P_VEC_F32, rs[i].cout, 0);
        if (i == 0) tpb = r; else if (r.off != expect) { fprintf(stderr, "libsdn: time_emb_proj bias not contiguous\n"); abort(); }
        expect = r.off + (int64_t)rs[i].cout * 4;
      }
    }
    u->tproj_total = total;
    gn_stats = Ref{SP_WS, arena.alloc((int64_t)B * 129 * 64 * 2 * 4)};
    plan->tscalar_off = arena.alloc(256);
    Act tsin = act(B, ch0);
    { Op o; o.kind = OP_TEMB; o.batch = B; o.c1 = ch0; o.out = R(tsin); snprintf(o.label, sizeof(o.label), "k_temb"); plan->ops.push_back(o); }
    x3t_hold = true;                                            // M = batch: the per-sample time-embedding linears
    Act t1 = act(B, tdim);
    gemm(B, tdim, ch0, R(tsin), l1w, l1b, R(t1), SDN_ACT_SILU);
    drop(tsin);
    Act semb = act(B, tdim);
    gemm(B, tdim, tdim, R(t1), l2w, l2b, R(semb), SDN_ACT_SILU);
    drop(t1);
    Act tp = act(B, total, 0, 0, 4);
    tproj = R(tp);
    gemm(B, total, tdim, R(semb), tpw, tpb, R(tp), SDN_ACT_NONE, Ref(), SDN_OUT_F32);
    drop(semb);
    x3t_hold = false;

    // ---- conv_in ----
    Ref ciw = param("conv_in.weight", SDN_P_CONV3X3, ch0, 9 * c.in_channels), cib = param("conv_in.bias", SDN_P_VEC_F32, ch0, 0);
    // latent_repeat: the r guidance branches share their latents -> conv_in, the first resnet and the first transformer
    // block up to its cross-attention query run on B / r samples (see transformer_body)
    const int rep = (c.latent_repeat > 1 && c.level_has_attn[0] && u->subbatch_bytes == 0) ? c.latent_repeat : 1;
    if (B % rep != 0) { plan->ws_bytes = -1; return; }            // forward rejects this batch
    const int Bfull = B, Bp = B / rep;
    Act hp = act((int64_t)Bp * S * S, ch0, S * S, S);
    { Op o; o.kind = OP_CONV_IN; o.batch = Bp; o.c1 = c.in_channels; o.c2 = ch0; o.hw = S; o.a = Ref{SP_LATENTS, 0}; o.w = ciw; o.bias = cib; o.out = R(hp);
      o.flops = 2.0 * Bp * S * S * (double)ch0 * 9 * c.in_channels; o.bytes = (double)Bp * S * S * (4.0 * c.in_channels + 2.0 * ch0);
      snprintf(o.label, sizeof(o.label), "k_conv_in"); plan->ops.push_back(o);
      plan->flops += o.flops; }
    Act h = hp;
    if (rep > 1) { h = act((int64_t)B * S * S, ch0, S * S, S); repeat(hp, h, rep); }

    std::vector<Act> skips;
    skips.push_back(h);                       // h stays alive as a skip; keep using it as the running tensor
    Act cur = h;
    bool cur_is_skip = true;

    // ---- down path ----
    for (int i = 0; i < c.n_levels; ++i) {
      const int cout = c.block_out_channels[i];
      for (int j = 0; j < c.layers_per_block; ++j) {
        snprintf(buf, sizeof(buf), "down_blocks.%d.resnets.%d", i, j);
        if (rep > 1 && i == 0 && j == 0) {        // shared prefix: resnet 0 and the head of transformer 0 at B / rep
          B = Bp;
          Act rp = resnet(buf, hp, nullptr, cout);
          B = Bfull;
          drop(hp);
          Act rf = act((int64_t)B * rp.hw, cout, rp.hw, rp.side);
          repeat(rp, rf, rep);
          snprintf(buf, sizeof(buf), "down_blocks.%d.attentions.%d", i, j);
          Act t = act_gn((int64_t)B * rp.hw, cout, rp.hw, rp.side);
          transformer_body(buf, rp, t, 0, rep, &rf);
          drop(rp); drop(rf);
          cur = t; skips.push_back(cur); cur_is_skip = true;
          continue;
        }
        Act r = resnet(buf, cur, nullptr, cout);
        if (!cur_is_skip) drop(cur);
        cur = r; cur_is_skip = false;
        if (c.level_has_attn[i]) {
          snprintf(buf, sizeof(buf), "down_blocks.%d.attentions.%d", i, j);
          Act t = transformer(buf, cur);
          drop(cur);
          cur = t;
        }
        skips.push_back(cur); cur_is_skip = true;
      }
      if (i + 1 < c.n_levels) {
        snprintf(buf, sizeof(buf), "down_blocks.%d.downsamplers.0.conv", i);
        Ref w = param(std::string(buf) + ".weight", SDN_P_CONV3X3, cout, 9 * cout), bb = param(std::string(buf) + ".bias", SDN_P_VEC_F32, cout, 0);
        const int s2 = cur.side / 2;
        Act d = act_gn((int64_t)B * s2 * s2, cout, s2 * s2, s2);
        want_stats(d);
        conv3x3(cur, cout, cout, w, bb, R(d), 2, 0, Ref(), Ref(), 0);
        cur = d; skips.push_back(cur); cur_is_skip = true;
      }
    }
    // ---- mid ----
    {
      Act r = resnet("mid_block.resnets.0", cur, nullptr, cur.C);
      cur = r; cur_is_skip = false;
      Act t = transformer("mid_block.attentions.0", cur);
      drop(cur); cur = t;
      Act r2 = resnet("mid_block.resnets.1", cur, nullptr, cur.C);
      drop(cur); cur = r2;
    }
    // ---- up path ----
    for (int i = 0; i < c.n_levels; ++i) {
      const int lvl = c.n_levels - 1 - i, cout = c.block_out_channels[lvl];
      for (int j = 0; j <= c.layers_per_block; ++j) {
        Act skip = skips.back(); skips.pop_back();
        snprintf(buf, sizeof(buf), "up_blocks.%d.resnets.%d", i, j);
        Act r = resnet(buf, cur, &skip, cout);
        drop(cur); drop(skip);
        cur = r;
        if (c.level_has_attn[lvl]) {
          snprintf(buf, sizeof(buf), "up_blocks.%d.attentions.%d", i, j);
          Act t = transformer(buf, cur);
          drop(cur); cur = t;
        }
      }
      if (i + 1 < c.n_levels) {
        snprintf(buf, sizeof(buf), "up_blocks.%d.upsamplers.0.conv", i);
        Ref w = param(std::string(buf) + ".weight", SDN_P_CONV3X3, cout, 9 * cout), bb = param(std::string(buf) + ".bias", SDN_P_VEC_F32, cout, 0);
        const int s2 = cur.side * 2;
        Act up = act_gn((int64_t)B * s2 * s2, cout, s2 * s2, s2);
        want_stats(up);
        conv3x3(cur, cout, cout, w, bb, R(up), 1, 1, Ref(), Ref(), 0);
        drop(cur); cur = up;
      }
    }
    // ---- tail: GN + SiLU + conv_out -> fp32 NCHW ----
    Ref og = param("conv_norm_out.weight", SDN_P_VEC_F32, ch0, 0), ob = param("conv_norm_out.bias", SDN_P_VEC_F32, ch0, 0);
    const int npad = 32;
    Ref cow = param("conv_out.weight", SDN_P_CONV3X3, c.out_channels, 9 * ch0, npad);
    Ref cob = param("conv_out.bias", SDN_P_VEC_F32, c.out_channels, 0, npad);
    Act g = act((int64_t)B * S * S, ch0, S * S, S);
    groupnorm(cur, nullptr, 1e-5f, 1, og, ob, g);
    drop(cur);
    conv3x3(g, c.out_channels, npad, cow, cob, Ref{SP_OUT, 0}, 1, 0, Ref(), Ref(), 0, SDN_OUT_F32_NCHW, c.out_channels);
    drop(g);
    plan->kv_base = Arena::up(arena.peak);
    plan->ws_bytes = plan_bad ? -1 : plan->kv_base + kv_top;
  }

  // =================================================================================================
  // AutoencoderKL decoder: vae.decode(latents / scaling_factor) of
  // StableDiffusionPipeline.decode_latents (...threshold_time.py:589).  diffusers-0.29.0 definitions (third party,
  // restated): Decoder = conv_in -> UNetMidBlock2D(resnet, 1-head attention, resnet) -> UpDecoderBlock2D x n
  // (layers_per_block + 1 resnets, nearest-2x + conv except the last) -> GroupNorm -> SiLU -> conv_out; resnets have no
  // time embedding, every norm uses eps 1e-6.
  // =================================================================================================
  Act vae_resnet(const std::string& pfx, Act& x, int cout) {
    const int cin = x.C;
    Ref n1g = param(pfx + ".norm1.weight", SDN_P_VEC_F32, cin, 0), n1b = param(pfx + ".norm1.bias", SDN_P_VEC_F32, cin, 0);
    Ref c1w = param(pfx + ".conv1.weight", SDN_P_CONV3X3, cout, 9 * cin), c1b = param(pfx + ".conv1.bias", SDN_P_VEC_F32, cout, 0);
    Ref n2g = param(pfx + ".norm2.weight", SDN_P_VEC_F32, cout, 0), n2b = param(pfx + ".norm2.bias", SDN_P_VEC_F32, cout, 0);
    Ref c2w = param(pfx + ".conv2.weight", SDN_P_CONV3X3, cout, 9 * cout), c2b = param(pfx + ".conv2.bias", SDN_P_VEC_F32, cout, 0);
    const int64_t rows = (int64_t)B * x.hw;
    Act g1 = act(rows, cin, x.hw, x.side);
    groupnorm(x, nullptr, 1e-6f, 1, n1g, n1b, g1);
    Act h = act_gn(rows, cout, x.hw, x.side);
    want_stats(h);
    conv3x3(g1, cout, cout, c1w, c1b, R(h), 1, 0, Ref(), Ref(), 0);
    drop(g1);
    Act g2 = act(rows, cout, x.hw, x.side);
    groupnorm(h, nullptr, 1e-6f, 1, n2g, n2b, g2);
    drop(h);
    Act out = act_gn(rows, cout, x.hw, x.side);
    if (cin != cout) {
      Ref scw = param(pfx + ".conv_shortcut.weight", SDN_P_MAT, cout, cin), scb = param(pfx + ".conv_shortcut.bias", SDN_P_VEC_F32, cout, 0);
      Act sc = act(rows, cout, x.hw, x.side);
      gemm(rows, cout, cin, R(x), scw, scb, R(sc));
      want_stats(out);
      conv3x3(g2, cout, cout, c2w, c2b, R(out), 1, 0, R(sc), Ref(), 0);
      drop(sc);
    } else {
      want_stats(out);
      conv3x3(g2, cout, cout, c2w, c2b, R(out), 1, 0, R(x), Ref(), 0);
    }
    drop(g2);
    return out;
  }

  // Attention(C, heads = 1, dim_head = C) with GroupNorm, biased q/k/v/out linears and a residual connection.
  // d = C = 512 does not fit the flash kernel: per image  S = Q K^T (fp32) -> row softmax -> P V.
  Act vae_attention(const std::string& pfx, Act& x) {
    const int C = x.C, hw = x.hw;
    const int64_t rows = (int64_t)B * hw;
    Ref gg = param(pfx + ".group_norm.weight", SDN_P_VEC_F32, C, 0), gb = param(pfx + ".group_norm.bias", SDN_P_VEC_F32, C, 0);
    Ref qw = param(pfx + ".to_q.weight", SDN_P_MAT, C, C), qb = param(pfx + ".to_q.bias", SDN_P_VEC_F32, C, 0);
    Ref kw = param(pfx + ".to_k.weight", SDN_P_MAT, C, C), kb = param(pfx + ".to_k.bias", SDN_P_VEC_F32, C, 0);
    Ref vw = param(pfx + ".to_v.weight", SDN_P_MAT, C, C), vb = param(pfx + ".to_v.bias", SDN_P_VEC_F32, C, 0);
    Ref ow = param(pfx + ".to_out.0.weight", SDN_P_MAT, C, C), ob = param(pfx + ".to_out.0.bias", SDN_P_VEC_F32, C, 0);
    Act gn = act(rows, C, hw, x.side);
    groupnorm(x, nullptr, 1e-6f, 0, gg, gb, gn);
    // three dense projections (not one stacked GEMM): the per-image Q K^T / P V GEMMs below take dense operands
    Act q = act(rows, C, hw, x.side), k = act(rows, C, hw, x.side), v = act(rows, C, hw, x.side);
    gemm(rows, C, C, R(gn), qw, qb, R(q));
    gemm(rows, C, C, R(gn), kw, kb, R(k));
    gemm(rows, C, C, R(gn), vw, vb, R(v));
    drop(gn);
    Act at = act(rows, C, hw, x.side);
    // fp32 scores of ONE image -- of one block of at most 4096 queries when the image has more tokens (1024 x 1024:
    // 16384) -- reused block after block (stream order)
    const int qrows = hw > 4096 ? 4096 : hw;
    Act sc = act(qrows, hw, 0, 0, 4);
    Act pr = act(qrows, hw);
    Act vt = act(C, hw);
    const float scale = 1.0f / sqrtf((float)C);
    for (int b = 0; b < B; ++b) {
      const int64_t img = (int64_t)b * hw * C * 2;
      { Op o; o.kind = OP_TRANSPOSE; o.a = Ref{SP_WS, v.off + img}; o.out = R(vt); o.rows = hw; o.c1 = C; o.ldq = C;
        o.ldo = hw; o.bytes = 4.0 * hw * C; snprintf(o.label, sizeof(o.label), "k_transpose16"); plan->ops.push_back(o); }
      for (int r0 = 0; r0 < hw; r0 += qrows) {
        const int64_t rowoff = (int64_t)r0 * C * 2;
        { // S = Q K^T : A = this block's Q rows, "weight" operand = the image's K rows
          Op o; o.kind = OP_GEMM; memset(&o.gd, 0, sizeof(o.gd));
          o.gd.M = qrows; o.gd.N = hw; o.gd.K = C; o.gd.a_mode = SDN_A_PLAIN; o.gd.out_kind = SDN_OUT_F32;
          o.a = Ref{SP_WS, q.off + img + rowoff}; o.w = Ref{SP_WS, k.off + img}; o.out = R(sc);
          o.flops = 2.0 * qrows * (double)hw * C; o.bytes = 2.0 * (qrows + (double)hw) * C + 4.0 * qrows * (double)hw;
          snprintf(o.label, sizeof(o.label), "k_gemm<%d>", sdn_gemm_pick_tile(qrows, hw, C, SDN_ACT_NONE));
          plan->ops.push_back(o); plan->flops += o.flops; plan->attn_flops += o.flops;
        }
        { Op o; o.kind = OP_SOFTMAX; o.a = R(sc); o.out = R(pr); o.rows = qrows; o.c1 = hw; o.scale = scale;
          o.bytes = 6.0 * qrows * (double)hw; snprintf(o.label, sizeof(o.label), "k_softmax_rows"); plan->ops.push_back(o); }
        { // O = P V : A = P [qrows, hw], "weight" = V^T [C, hw]
          Op o; o.kind = OP_GEMM; memset(&o.gd, 0, sizeof(o.gd));
          o.gd.M = qrows; o.gd.N = C; o.gd.K = hw; o.gd.a_mode = SDN_A_PLAIN; o.gd.out_kind = SDN_OUT_BF16;
          o.a = R(pr); o.w = R(vt); o.out = Ref{SP_WS, at.off + img + rowoff};
          o.flops = 2.0 * qrows * (double)hw * C; o.bytes = 2.0 * (qrows * (double)hw + (qrows + (double)hw) * C);
          snprintf(o.label, sizeof(o.label), "k_gemm<%d>", sdn_gemm_pick_tile(qrows, C, hw, SDN_ACT_NONE));
          plan->ops.push_back(o); plan->flops += o.flops; plan->attn_flops += o.flops;
        }
      }
    }
    drop(sc); drop(pr); drop(vt); drop(q); drop(k); drop(v);
    Act out = act_gn(rows, C, hw, x.side);
    want_stats(out);
    gemm(rows, C, C, R(at), ow, ob, R(out), SDN_ACT_NONE, R(x));
    drop(at);
    return out;
  }

  void build_vae() {
    const sdn_vae_config& c = u->vcfg;
    const int S = c.sample_size, L = c.latent_channels, n = c.n_levels;
    const int ctop = c.block_out_channels[n - 1];
    char buf[96];
    gn_stats = Ref{SP_WS, arena.alloc((int64_t)B * 129 * 64 * 2 * 4)};
    Ref pqw = param("post_quant_conv.weight", SDN_P_VEC_F32, L * L, 0), pqb = param("post_quant_conv.bias", SDN_P_VEC_F32, L, 0);
    Act z = act((int64_t)B * L, S * S, 0, 0, 4);                    // fp32 NCHW
    { Op o; o.kind = OP_LATENT_MIX; o.batch = B; o.c1 = L; o.hw = S * S; o.a = Ref{SP_LATENTS, 0}; o.w = pqw; o.bias = pqb; o.out = R(z);
      o.flops = 2.0 * B * S * S * (double)L * L; o.bytes = 8.0 * B * S * S * L; snprintf(o.label, sizeof(o.label), "k_latent_mix");
      plan->ops.push_back(o); plan->flops += o.flops; }
    Ref ciw = param("decoder.conv_in.weight", SDN_P_CONV3X3, ctop, 9 * L), cib = param("decoder.conv_in.bias", SDN_P_VEC_F32, ctop, 0);
    Act cur = act((int64_t)B * S * S, ctop, S * S, S);
    { Op o; o.kind = OP_CONV_IN; o.batch = B; o.c1 = L; o.c2 = ctop; o.hw = S; o.a = R(z); o.w = ciw; o.bias = cib; o.out = R(cur);
      o.flops = 2.0 * B * S * S * (double)ctop * 9 * L; o.bytes = (double)B * S * S * (4.0 * L + 2.0 * ctop);
      snprintf(o.label, sizeof(o.label), "k_conv_in"); plan->ops.push_back(o); plan->flops += o.flops; }
    drop(z);
    { Act r = vae_resnet("decoder.mid_block.resnets.0", cur, ctop); drop(cur); cur = r; }
    { Act r = vae_attention("decoder.mid_block.attentions.0", cur); drop(cur); cur = r; }
    { Act r = vae_resnet("decoder.mid_block.resnets.1", cur, ctop); drop(cur); cur = r; }
    for (int i = 0; i < n; ++i) {
      const int cout = c.block_out_channels[n - 1 - i];
      for (int j = 0; j <= c.layers_per_block; ++j) {
        snprintf(buf, sizeof(buf), "decoder.up_blocks.%d.resnets.%d", i, j);
        Act r = vae_resnet(buf, cur, cout);
        drop(cur); cur = r;
      }
      if (i + 1 < n) {
        snprintf(buf, sizeof(buf), "decoder.up_blocks.%d.upsamplers.0.conv", i);
        Ref w = param(std::string(buf) + ".weight", SDN_P_CONV3X3, cout, 9 * cout), bb = param(std::string(buf) + ".bias", SDN_P_VEC_F32, cout, 0);
        const int s2 = cur.side * 2;
        Act up = act_gn((int64_t)B * s2 * s2, cout, s2 * s2, s2);
        want_stats(up);
        conv3x3(cur, cout, cout, w, bb, R(up), 1, 1, Ref(), Ref(), 0);
        drop(cur); cur = up;
      }
    }
    const int c0 = c.block_out_channels[0];
    Ref og = param("decoder.conv_norm_out.weight", SDN_P_VEC_F32, c0, 0), ob = param("decoder.conv_norm_out.bias", SDN_P_VEC_F32, c0, 0);
    const int npad = 32;
    Ref cow = param("decoder.conv_out.weight", SDN_P_CONV3X3, c.out_channels, 9 * c0, npad);
    Ref cob = param("decoder.conv_out.bias", SDN_P_VEC_F32, c.out_channels, 0, npad);
    Act g = act((int64_t)B * cur.hw, c0, cur.hw, cur.side);
    groupnorm(cur, nullptr, 1e-6f, 1, og, ob, g);
    drop(cur);
    conv3x3(g, c.out_channels, npad, cow, cob, Ref{SP_OUT, 0}, 1, 0, Ref(), Ref(), 0, SDN_OUT_F32_NCHW, c.out_channels);
    drop(g);
    plan->ws_bytes = arena.peak;
  }
  // AutoencoderKL encoder (the proj_ref builder's embed_fn, run_nudity.py:308): conv_in -> DownEncoderBlock2D x n
  // (layers_per_block resnets; Downsample2D(padding=0) = F.pad (0,1,0,1) + conv3x3 stride 2 on all but the last) ->
  // UNetMidBlock2D -> GroupNorm -> SiLU -> conv_out (2L moments) -> quant_conv 1x1.  Output: fp32 NCHW moments
  // [B, 2L, S, S] (mean | logvar); sampling is sdn_gaussian_sample.
  void build_vae_encoder() {
    const sdn_vae_config& c = u->vcfg;
    const int n = c.n_levels, L = c.latent_channels;
    const int S0 = c.sample_size << (n - 1);                         // image side
    char buf[96];
    gn_stats = Ref{SP_WS, arena.alloc((int64_t)B * 129 * 64 * 2 * 4)};
    const int c0 = c.block_out_channels[0];
    Ref ciw = param("encoder.conv_in.weight", SDN_P_CONV3X3, c0, 9 * c.out_channels), cib = param("encoder.conv_in.bias", SDN_P_VEC_F32, c0, 0);
    Act cur = act((int64_t)B * S0 * S0, c0, S0 * S0, S0);
    { Op o; o.kind = OP_CONV_IN; o.batch = B; o.c1 = c.out_channels; o.c2 = c0; o.hw = S0; o.a = Ref{SP_LATENTS, 0}; o.w = ciw; o.bias = cib; o.out = R(cur);
      o.flops = 2.0 * B * S0 * S0 * (double)c0 * 9 * c.out_channels; o.bytes = (double)B * S0 * S0 * (4.0 * c.out_channels + 2.0 * c0);
      snprintf(o.label, sizeof(o.label), "k_conv_in"); plan->ops.push_back(o); plan->flops += o.flops; }
    for (int i = 0; i < n; ++i) {
      const int cout = c.block_out_channels[i];
      for (int j = 0; j < c.layers_per_block; ++j) {
        snprintf(buf, sizeof(buf), "encoder.down_blocks.%d.resnets.%d", i, j);
        Act r = vae_resnet(buf, cur, cout);
        drop(cur); cur = r;
      }
      if (i + 1 < n) {
        snprintf(buf, sizeof(buf), "encoder.down_blocks.%d.downsamplers.0.conv", i);
        Ref w = param(std::string(buf) + ".weight", SDN_P_CONV3X3, cout, 9 * cout), bb = param(std::string(buf) + ".bias", SDN_P_VEC_F32, cout, 0);
        const int s2 = cur.side / 2;
        Act d = act_gn((int64_t)B * s2 * s2, cout, s2 * s2, s2);
        want_stats(d);
        conv3x3(cur, cout, cout, w, bb, R(d), 2, 0, Ref(), Ref(), 0, SDN_OUT_BF16, 0, 1);
        drop(cur); cur = d;
      }
    }
    const int ctop = c.block_out_channels[n - 1];
    { Act r = vae_resnet("encoder.mid_block.resnets.0", cur, ctop); drop(cur); cur = r; }
    { Act r = vae_attention("encoder.mid_block.attentions.0", cur); drop(cur); cur = r; }
    { Act r = vae_resnet("encoder.mid_block.resnets.1", cur, ctop); drop(cur); cur = r; }
    Ref og = param("encoder.conv_norm_out.weight", SDN_P_VEC_F32, ctop, 0), ob = param("encoder.conv_norm_out.bias", SDN_P_VEC_F32, ctop, 0);
    const int npad = 32, M2 = 2 * L;
    Ref cow = param("encoder.conv_out.weight", SDN_P_CONV3X3, M2, 9 * ctop, npad);
    Ref cob = param("encoder.conv_out.bias", SDN_P_VEC_F32, M2, 0, npad);
    Ref qw = param("quant_conv.weight", SDN_P_VEC_F32, M2 * M2, 0), qb = param("quant_conv.bias", SDN_P_VEC_F32, M2, 0);
    Act g = act((int64_t)B * cur.hw, ctop, cur.hw, cur.side);
    groupnorm(cur, nullptr, 1e-6f, 1, og, ob, g);
    const int hw = cur.hw;
    drop(cur);
    Act mom = act((int64_t)B * M2, hw, 0, 0, 4);                     // fp32 NCHW moments before quant_conv
    conv3x3(g, M2, npad, cow, cob, R(mom), 1, 0, Ref(), Ref(), 0, SDN_OUT_F32_NCHW, M2);
    drop(g);
    { Op o; o.kind = OP_LATENT_MIX; o.batch = B; o.c1 = M2; o.hw = hw; o.a = R(mom); o.w = qw; o.bias = qb; o.out = Ref{SP_OUT, 0};
      o.mod = 1; o.scale = 1.0f;
      o.flops = 2.0 * B * hw * (double)M2 * M2; o.bytes = 8.0 * B * hw * M2; snprintf(o.label, sizeof(o.label), "k_latent_mix");
      plan->ops.push_back(o); plan->flops += o.flops; }
    drop(mom);
    plan->ws_bytes = arena.peak;
  }
  // =================================================================================================
  // CLIP text encoder: `self.text_encoder(input_ids, attention_mask)[0]`
  // (...threshold_time.py:197,225,287,333).  transformers' CLIPTextModel (third party): token + position embeddings,
  // pre-LN transformer layers with CAUSAL self-attention and a quick-GELU MLP, final LayerNorm.
  // =================================================================================================
  Ref stacked_vec(const std::vector<std::string>& names, int n_each) {
    Ref first; int64_t expect = -1;
    for (size_t i = 0; i < names.size(); ++i) {
      Ref r = param(names[i], SDN_P_VEC_F32, n_each, 0);
      if (i == 0) first = r;
      else if (r.off != expect) { fprintf(stderr, "libsdn: stacked bias %s is not contiguous\n", names[i].c_str()); abort(); }
      expect = r.off + (int64_t)n_each * 4;
    }
    return first;
  }
  void build_clip() {
    const sdn_clip_config& c = u->ccfg;
    const int C = c.hidden_size, I = c.intermediate_size, n = c.max_position_embeddings, H = c.num_heads;
    const int64_t rows = (int64_t)B * n;
    Ref tok = param("embeddings.token_embedding.weight", SDN_P_MAT, c.vocab_size, C);
    Ref pos = param("embeddings.position_embedding.weight", SDN_P_MAT, n, C);
    Act x = act(rows, C, n, 0);
    { Op o; o.kind = OP_CLIP_EMBED; o.a = Ref{SP_LATENTS, 0}; o.w = tok; o.bias = pos; o.out = R(x); o.rows = rows; o.hw = n; o.c1 = C;
      o.c2 = c.vocab_size; o.bytes = 6.0 * rows * C; snprintf(o.label, sizeof(o.label), "k_clip_embed"); plan->ops.push_back(o); }
    char buf[96];
    for (int l = 0; l < c.num_layers; ++l) {
      snprintf(buf, sizeof(buf), "encoder.layers.%d", l);
      const std::string p = buf;
      Ref l1g = param(p + ".layer_norm1.weight", SDN_P_VEC_F32, C, 0), l1b = param(p + ".layer_norm1.bias", SDN_P_VEC_F32, C, 0);
      Ref qkvw = stacked({p + ".self_attn.q_proj.weight", p + ".self_attn.k_proj.weight", p + ".self_attn.v_proj.weight"}, C, C);
      Ref qkvb = stacked_vec({p + ".self_attn.q_proj.bias", p + ".self_attn.k_proj.bias", p + ".self_attn.v_proj.bias"}, C);
      Ref ow = param(p + ".self_attn.out_proj.weight", SDN_P_MAT, C, C), ob = param(p + ".self_attn.out_proj.bias", SDN_P_VEC_F32, C, 0);
      Ref l2g = param(p + ".layer_norm2.weight", SDN_P_VEC_F32, C, 0), l2b = param(p + ".layer_norm2.bias", SDN_P_VEC_F32, C, 0);
      Ref f1w = param(p + ".mlp.fc1.weight", SDN_P_MAT, I, C), f1b = param(p + ".mlp.fc1.bias", SDN_P_VEC_F32, I, 0);
      Ref f2w = param(p + ".mlp.fc2.weight", SDN_P_MAT, C, I), f2b = param(p + ".mlp.fc2.bias", SDN_P_VEC_F32, C, 0);
      Act ln = act(rows, C, n, 0);
      layernorm(x, l1g, l1b, ln);
      Act qkv = act(rows, 3 * C, n, 0);
      gemm(rows, 3 * C, C, R(ln), qkvw, qkvb, R(qkv));
      Act at = act(rows, C, n, 0);
      { Op o; o.kind = OP_MATTN; o.a = R(qkv); o.k = Ref{SP_WS, qkv.off + (int64_t)C * es}; o.v = Ref{SP_WS, qkv.off + (int64_t)2 * C * es};
        o.out = R(at); o.batch = B; o.heads = H; o.nq = n; o.nk = n; o.hd = C / H; o.ldq = o.ldk = o.ldv = 3 * C; o.ldo = C;
        o.scale = 1.0f / sqrtf((float)o.hd);
        o.flops = 4.0 * B * H * (double)n * n * o.hd; o.bytes = 2.0 * 4.0 * rows * C;
        snprintf(o.label, sizeof(o.label), "k_attn<%d>", o.hd); plan->ops.push_back(o); plan->flops += o.flops; plan->attn_flops += o.flops; }
      drop(qkv);
      Act x2 = act(rows, C, n, 0);
      gemm(rows, C, C, R(at), ow, ob, R(x2), SDN_ACT_NONE, R(x));
      drop(at); drop(x);
      layernorm(x2, l2g, l2b, ln);
      Act h = act(rows, I, n, 0);
      gemm(rows, I, C, R(ln), f1w, f1b, R(h), SDN_ACT_QUICK_GELU);
      drop(ln);
      x = act(rows, C, n, 0);
      gemm(rows, C, I, R(h), f2w, f2b, R(x), SDN_ACT_NONE, R(x2));
      drop(h); drop(x2);
    }
    Ref fg = param("final_layer_norm.weight", SDN_P_VEC_F32, C, 0), fb = param("final_layer_norm.bias", SDN_P_VEC_F32, C, 0);
    { Op o; o.kind = OP_LN; o.a = R(x); o.rows = rows; o.c1 = C; o.eps = 1e-5f; o.w = fg; o.bias = fb; o.out = Ref{SP_OUT, 0};
      o.bytes = 4.0 * rows * C; snprintf(o.label, sizeof(o.label), "k_layernorm"); plan->ops.push_back(o); }
    drop(x);
    plan->ws_bytes = arena.peak;
  }
};

Plan* get_plan(sdn_unet* u, int batch) {
  auto it = u->plans.find(batch);
  if (it != u->plans.end()) return &it->second;
  Plan& p = u->plans[batch];
  p.batch = batch;
  Builder b{u, &p};
  b.B = batch;
  b.es = (!u->is_vae && !u->is_mmdit && u->cfg.dtype >= 2) ? 4 : 2;       // fp32 storage: SD-v1.4 UNet and CLIP text encoder plans
  b.x3t = !u->is_vae && !u->is_mmdit && !u->is_clip && u->cfg.dtype == 3 && u->x3_expand;
  if (u->is_clip) b.build_clip(); else if (u->is_vae_encoder) b.build_vae_encoder(); else if (u->is_vae) b.build_vae(); else if (u->is_mmdit) b.build_mmdit(); else b.build();
  return &p;
}

inline const char* resolve(const Ref& r, const char* w, const char* ws, const char* lat, const char* text, const char* out,
                           const char* pooled, const char* kv = nullptr) {
  switch (r.space) {
    case SP_W: return w + r.off;
    case SP_WS: return ws + r.off;
    case SP_KV: return kv + r.off;
    case SP_LATENTS: return lat + r.off;
    case SP_TEXT: return text + r.off;
    case SP_OUT: return out + r.off;
    case SP_POOLED: return pooled + r.off;
    default: return nullptr;
  }
}

}  // namespace

extern "C" {

int sdn_unet_create(const sdn_unet_config* cfg, sdn_unet** out) {
  if (!cfg || !out) return SDN_E_INVALID;
  if (cfg->n_levels < 1 || cfg->n_levels > 4 || cfg->layers_per_block < 1 || cfg->n_heads <= 0 ||
      cfg->in_channels <= 0 || cfg->in_channels > 16 || cfg->out_channels <= 0 || cfg->out_channels > 32 ||
      cfg->sample_size <= 0 || (cfg->sample_size % (1 << (cfg->n_levels - 1))) != 0 || cfg->cross_dim % 64 != 0 ||
      cfg->text_len <= 0 || cfg->norm_groups <= 0 || cfg->norm_groups > 64 || cfg->dtype < 0 || cfg->dtype > 3 ||
      cfg->latent_repeat < 0 || cfg->latent_repeat > 8)
    return SDN_E_INVALID;
  for (int i = 0; i < cfg->n_levels; ++i) {
    const int c = cfg->block_out_channels[i];
    if (c <= 0 || c % 64 != 0 || c % cfg->norm_groups != 0 || c % cfg->n_heads != 0) return SDN_E_INVALID;
    const int hd = c / cfg->n_heads;
    if (cfg->level_has_attn[i] && hd != 40 && hd != 64 && hd != 80 && hd != 160) return SDN_E_INVALID;
    if (sdn_gemm_pick_nrep(c, SDN_ACT_NONE) == 0 || sdn_gemm_pick_nrep(8 * c, SDN_ACT_GEGLU) == 0) return SDN_E_INVALID;
  }
  {
    const int hd_mid = cfg->block_out_channels[cfg->n_levels - 1] / cfg->n_heads;       // mid block always attends
    if (hd_mid != 40 && hd_mid != 64 && hd_mid != 80 && hd_mid != 160) return SDN_E_INVALID;
  }
  sdn_unet* u = new sdn_unet();
  u->cfg = *cfg;
  if (cfg->dtype >= 2) { u->gn_fuse = false; u->ln_fold = false; u->ff_fuse = false; }   // fp32-storage modes: the plain operator chain (sdn_f32.hip)
  get_plan(u, cfg->latent_repeat > 1 ? cfg->latent_repeat : 1);   // registers the parameter manifest (batch-independent)
  *out = u;
  return SDN_OK;
}

int sdn_mmdit_create(const sdn_mmdit_config* cfg, sdn_unet** out) {
  if (!cfg || !out) return SDN_E_INVALID;
  const int C = cfg->num_heads * cfg->head_dim;
  if (cfg->in_channels <= 0 || cfg->out_channels <= 0 || cfg->sample_size <= 0 || cfg->patch_size <= 0 ||
      cfg->sample_size % cfg->patch_size != 0 || cfg->num_layers <= 0 || cfg->num_heads <= 0 || cfg->head_dim != 64 ||
      C % 128 != 0 || cfg->joint_dim % 64 != 0 || cfg->pooled_dim % 64 != 0 || cfg->time_dim % 64 != 0 ||
      (cfg->in_channels * cfg->patch_size * cfg->patch_size) % 64 != 0 ||
      (cfg->out_channels * cfg->patch_size * cfg->patch_size) % 32 != 0 || cfg->text_len <= 0 || cfg->dtype < 0 ||
      cfg->dtype > 1 || C > 2048)
    return SDN_E_INVALID;
  sdn_unet* u = new sdn_unet();
  memset(&u->cfg, 0, sizeof(u->cfg));
  u->mcfg = *cfg;
  u->is_mmdit = true;
  get_plan(u, 1);
  *out = u;
  return SDN_OK;
}

int sdn_vae_decoder_create(const sdn_vae_config* cfg, sdn_unet** out) {
  if (!cfg || !out) return SDN_E_INVALID;
  if (cfg->n_levels < 1 || cfg->n_levels > 4 || cfg->layers_per_block < 1 || cfg->latent_channels <= 0 ||
      cfg->latent_channels > 16 || cfg->out_channels <= 0 || cfg->out_channels > 32 || cfg->sample_size <= 0 ||
      cfg->norm_groups <= 0 || cfg->norm_groups > 64 || cfg->dtype < 0 || cfg->dtype > 1)
    return SDN_E_INVALID;
  for (int i = 0; i < cfg->n_levels; ++i) {
    const int c = cfg->block_out_channels[i];
    if (c <= 0 || c % 64 != 0 || c % cfg->norm_groups != 0 || sdn_gemm_pick_nrep(c, SDN_ACT_NONE) == 0) return SDN_E_INVALID;
  }
  const int hw = cfg->sample_size * cfg->sample_size;         // tokens of the mid-block attention
  if (hw % 64 != 0 || hw > 16384 || (hw > 4096 && hw % 4096 != 0) || sdn_gemm_pick_nrep(hw, SDN_ACT_NONE) == 0) return SDN_E_INVALID;
  sdn_unet* u = new sdn_unet();
  memset(&u->cfg, 0, sizeof(u->cfg));
  u->cfg.norm_groups = cfg->norm_groups;
  u->cfg.dtype = cfg->dtype;
  u->vcfg = *cfg;
  u->is_vae = true;
  get_plan(u, 1);
  *out = u;
  return SDN_OK;
}

int sdn_vae_encoder_create(const sdn_vae_config* cfg, sdn_unet** out) {
  if (!cfg || !out) return SDN_E_INVALID;
  sdn_unet* u = nullptr;
  const int rc = sdn_vae_decoder_create(cfg, &u);              // same config checks; the plan is rebuilt as an encoder
  if (rc != SDN_OK) return rc;
  if (2 * cfg->latent_channels > 16) { delete u; return SDN_E_INVALID; }
  u->is_vae_encoder = true;
  u->plans.clear(); u->params.clear(); u->param_index.clear(); u->weight_bytes = 0;
  get_plan(u, 1);
  *out = u;
  return SDN_OK;
}

int sdn_clip_create(const sdn_clip_config* cfg, sdn_unet** out) {
  if (!cfg || !out) return SDN_E_INVALID;
  if (cfg->vocab_size <= 0 || cfg->hidden_size <= 0 || cfg->hidden_size % 128 != 0 || cfg->hidden_size > 1024 ||
      cfg->intermediate_size <= 0 || cfg->intermediate_size % 128 != 0 || cfg->num_layers <= 0 || cfg->num_heads <= 0 ||
      cfg->hidden_size != 64 * cfg->num_heads || cfg->max_position_embeddings <= 0 || cfg->max_position_embeddings > 4096 ||
      cfg->dtype < 0 || cfg->dtype > 3)
    return SDN_E_INVALID;
  sdn_unet* u = new sdn_unet();
  memset(&u->cfg, 0, sizeof(u->cfg));
  u->cfg.dtype = cfg->dtype;
  if (cfg->dtype >= 2) { u->gn_fuse = false; u->ln_fold = false; u->ff_fuse = false; }   // fp32-storage modes: the plain operator chain
  u->ccfg = *cfg;
  u->is_clip = true;
  get_plan(u, 1);
  *out = u;
  return SDN_OK;
}

int sdn_clip_forward(sdn_unet* m, const void* weights, const int32_t* input_ids, const int32_t* attention_mask,
                     void* last_hidden_state, int32_t batch, void* workspace, size_t workspace_bytes, void* stream);

int sdn_unet_prepare(sdn_unet* u, void* weights, void* stream) {
  if (!u || !weights) return SDN_E_INVALID;
  char* W = (char*)weights;
  const int dt = (u->is_mmdit ? u->mcfg.dtype : u->cfg.dtype) == 1 ? 1 : 0;
  for (const auto& j : u->fold_jobs) {
    if (j.kind == 2) {
      const int rc2 = sdn_expand3_weights((const float*)(W + j.w), j.rows, j.cols, j.group, W + j.wf, stream);
      if (rc2 != SDN_OK) return rc2;
      continue;
    }
    if (j.kind == 1) {
      const int rc1 = sdn_linear_pair_fold(dt, W + j.w, W + j.gamma, (const float*)(W + j.beta), (const float*)(W + j.bias), j.rows, j.cols,
                                           W + j.wf, (float*)(W + j.c), stream);
      if (rc1 != SDN_OK) return rc1;
      continue;
    }
    const int rc = sdn_ln_fold(dt, W + j.w, (const float*)(W + j.gamma), (const float*)(W + j.beta),
                               j.bias >= 0 ? (const float*)(W + j.bias) : nullptr, j.rows, j.cols, W + j.wf, (float*)(W + j.c),
                               (float*)(W + j.d), stream);
    if (rc != SDN_OK) return rc;
  }
  return SDN_OK;
}

void sdn_unet_destroy(sdn_unet* u) { delete u; }

int sdn_unet_param_count(const sdn_unet* u) { return u ? (int)u->params.size() : 0; }

int sdn_unet_param_info(const sdn_unet* u, int32_t index, sdn_param_info* info) {
  if (!u || !info || index < 0 || index >= (int)u->params.size()) return SDN_E_INVALID;
  *info = u->params[index];
  return SDN_OK;
}

size_t sdn_unet_weight_bytes(const sdn_unet* u) { return u ? (size_t)u->weight_bytes : 0; }

static int vae_chunk(const sdn_unet* v);
size_t sdn_unet_workspace_bytes(sdn_unet* u, int32_t batch) {
  if (!u || batch <= 0) return 0;
  if (u->is_vae) {                                                 // VAE entry points run large batches in chunks of vae_chunk()
    const int cap = vae_chunk(u);
    if (cap == 0) return 0;
    if (batch > cap) batch = cap;
  }
  const int64_t b = get_plan(u, batch)->ws_bytes;
  return b < 0 ? 0 : (size_t)b;
}

double sdn_unet_flops(sdn_unet* u, int32_t batch, double* attn) {
  if (!u || batch <= 0) return 0.0;
  Plan* p = get_plan(u, batch);
  if (attn) *attn = p->attn_flops;
  return p->flops;
}

static int run_plan(sdn_unet* u, const void* weights, const float* latents, float timestep, const void* text,
                    const void* pooled, float* out, int32_t batch, void* workspace, size_t workspace_bytes, void* stream);

// Images one plan invocation of a VAE takes: byte offsets inside one activation are 32-bit in the GEMM's DMA descriptors, so the
// largest tensor (batch x side^2 x widest channel count x 2 B) must stay below 4 GiB; larger batches are cut into chunks INSIDE
// the entry points (each chunk replays the same plan on the same workspace, stream-ordered).
static int vae_chunk(const sdn_unet* v) {
  const sdn_vae_config& c = v->vcfg;
  const int64_t side = (int64_t)c.sample_size << (c.n_levels - 1);
  int64_t cmax = 0;
  for (int i = 0; i < c.n_levels; ++i) if (c.block_out_channels[i] > cmax) cmax = c.block_out_channels[i];
  int64_t cap = (((int64_t)1 << 32) - 1) / (side * side * cmax * 2);
  if (cap > 8) cap = 8;                                            // (beyond 8 images the kernels are saturated; bounds the workspace)
  return cap < 1 ? 0 : (int)cap;
}

int sdn_vae_decode(sdn_unet* v, const void* weights, const float* latents, float latent_scale, float* image, int32_t batch,
                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!v || !v->is_vae || v->is_vae_encoder || batch < 0) return SDN_E_INVALID;
  const int cap = vae_chunk(v);
  if (cap == 0) return SDN_E_INVALID;                              // a single image already exceeds the 32-bit offsets
  const sdn_vae_config& c = v->vcfg;
  const int64_t side = (int64_t)c.sample_size << (c.n_levels - 1);
  const int64_t lat_n = (int64_t)c.latent_channels * c.sample_size * c.sample_size, img_n = (int64_t)c.out_channels * side * side;
  for (int lo = 0; lo < batch; lo += cap) {
    const int nb = batch - lo < cap ? batch - lo : cap;
    const int rc = run_plan(v, weights, latents + lo * lat_n, latent_scale, weights /* no text operand */, nullptr, image + lo * img_n,
                            nb, workspace, workspace_bytes, stream);
    if (rc != SDN_OK) return rc;
  }
  return SDN_OK;
}

int sdn_vae_encode(sdn_unet* v, const void* weights, const float* image, float* moments, int32_t batch, void* workspace,
                   size_t workspace_bytes, void* stream) {
  if (!v || !v->is_vae || !v->is_vae_encoder || batch < 0) return SDN_E_INVALID;
  const int cap = vae_chunk(v);
  if (cap == 0) return SDN_E_INVALID;
  const sdn_vae_config& c = v->vcfg;
  const int64_t side = (int64_t)c.sample_size << (c.n_levels - 1);
  const int64_t img_n = (int64_t)c.out_channels * side * side, mom_n = (int64_t)2 * c.latent_channels * c.sample_size * c.sample_size;
  for (int lo = 0; lo < batch; lo += cap) {
    const int nb = batch - lo < cap ? batch - lo : cap;
    const int rc = run_plan(v, weights, image + lo * img_n, 1.0f, weights /* no text operand */, nullptr, moments + lo * mom_n, nb,
                            workspace, workspace_bytes, stream);
    if (rc != SDN_OK) return rc;
  }
  return SDN_OK;
}

int sdn_clip_forward(sdn_unet* m, const void* weights, const int32_t* input_ids, const int32_t* attention_mask,
                     void* last_hidden_state, int32_t batch, void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !m->is_clip) return SDN_E_INVALID;
  m->clip_mask = attention_mask;
  return run_plan(m, weights, (const float*)input_ids, 0.f, weights /* no text operand */, nullptr, (float*)last_hidden_state,
                  batch, workspace, workspace_bytes, stream);
}

int sdn_unet_forward(sdn_unet* u, const void* weights, const float* latents, float timestep, const void* text,
                     float* out, int32_t batch, void* workspace, size_t workspace_bytes, void* stream) {
  if (!u || u->is_mmdit || u->is_vae || u->is_clip) return SDN_E_INVALID;
  return run_plan(u, weights, latents, timestep, text, nullptr, out, batch, workspace, workspace_bytes, stream);
}

int sdn_mmdit_forward(sdn_unet* u, const void* weights, const float* latents, float timestep, const void* text,
                      const void* pooled, float* out, int32_t batch, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (!u || !u->is_mmdit || u->is_vae || u->is_clip || !pooled) return SDN_E_INVALID;
  return run_plan(u, weights, latents, timestep, text, pooled, out, batch, workspace, workspace_bytes, stream);
}

// A GEMM of the bf16x3 plan on sdn_gemm_bf16 (triple operands, expanded weights).  The LDS-DMA tiles address each operand with
// 31-bit byte offsets, and a triple is 1.5 x its f32 tensor: the launch is cut into row chunks (whole samples for a conv) that
// stay below 2 GiB per operand.  Rows are independent, so the chunks are the same arithmetic.
static long g_x3_chunk_limit = (1L << 31) - 4096;      // bytes of A operand per launch (tests lower it: sdn_debug_set_x3_chunk_bytes)
extern "C" void sdn_debug_set_x3_chunk_bytes(long long bytes) { g_x3_chunk_limit = bytes > 0 ? (long)bytes : (1L << 31) - 4096; }

static int launch_x3t_gemm(const Op& o, const char* a, const char* w, const float* bias, const float* rowbias, const char* residual,
                           void* out, void* stream) {
  const sdn_gemm_desc& d = o.gd;
  const bool conv = d.a_mode == SDN_A_CONV3X3;
  const long rows_out_unit = conv ? (long)d.Ho * d.Wo : 256;                 // chunk granularity in output rows
  const long a_bytes_unit = conv ? (long)d.Hs * d.Ws * d.Cin * 2 : 256L * d.K * 2;
  const long units = conv ? d.M / rows_out_unit : (d.M + 255) / 256;
  long per = g_x3_chunk_limit / a_bytes_unit;                                 // units per launch
  if (per < 1) return SDN_E_INVALID;
  if (per > units) per = units;
  if (per < units) {
    // More than one launch: EQUAL chunks that are whole waves of tiles where the operand allows it.  "As many units as fit, then the
    // rest" gave 192 samples of a 960-channel 64^2 conv as 91 + 91 + 10 (6 + 6 + 1 waves of 256-row tiles on 256 CUs against 12 for
    // the rows themselves) and the 3072 row blocks of FF2 at 64^2 as 1092 + 1092 + 888 (5 + 5 + 4); 64 + 64 + 64 and 3 x 1024 are 12.
    const long n = (units + per - 1) / per;
    long even = (units + n - 1) / n;
    const int nrep = sdn_gemm_pick_tile((int)(even * rows_out_unit < d.M ? even * rows_out_unit : d.M), d.N, d.K, d.act, 0);
    const long tile_rows = nrep >= 8 ? 256 : 128, tile_cols = 32L * (nrep > 0 ? nrep : 2);
    const long tiles_per_unit = ((rows_out_unit + tile_rows - 1) / tile_rows) * ((d.N + tile_cols - 1) / tile_cols);
    long a_ = 256, b_ = tiles_per_unit % 256;                                 // q = units per whole wave of 256 tiles = 256 / gcd(256, tiles_per_unit)
    while (b_) { const long t_ = a_ % b_; a_ = b_; b_ = t_; }
    const long q = 256 / a_;
    const long aligned = (even + q - 1) / q * q;
    if (aligned <= per) even = aligned;
    per = even;
  }
  const int n_cols = d.x3_out == 2 ? d.N / 2 : d.N;                           // logical output width
  const long out_row_bytes = d.x3_out == 0 ? 0 : ((d.x3_out == 1 || d.x3_out == 4) ? 4L * n_cols : 6L * n_cols);
  if (d.x3_out == 0 && per < units) return SDN_E_INVALID;                     // (the NCHW output of conv_out is not chunked: 4 channels)
  for (long u0 = 0; u0 < units; u0 += per) {
    const long nu = units - u0 < per ? units - u0 : per;
    sdn_gemm_desc c = d;
    const long r0 = u0 * rows_out_unit;
    long rn = nu * rows_out_unit;
    if (r0 + rn > d.M) rn = d.M - r0;
    c.M = (int)rn;
    const float* rb = rowbias;
    if (rowbias && d.rows_per_batch > 0) rb = rowbias + (r0 / d.rows_per_batch) * d.ld_rowbias;   // chunks start on sample boundaries when it matters (conv)
    const int rc = sdn_gemm_bf16(&c, a + u0 * a_bytes_unit, nullptr, w, bias, rb, nullptr,
                                 residual ? residual + r0 * 4L * n_cols : nullptr, (char*)out + r0 * out_row_bytes, stream);
    if (rc != SDN_OK) return rc;
  }
  return SDN_OK;
}

// Launches every op of the plan on `stream`.  t_dev != nullptr: the timestep is read from device memory (graph mode).
static int launch_ops(sdn_unet* u, Plan* p, const char* W, const char* WS, const char* L, const char* T, const char* O,
                      const char* PL, float timestep, const float* t_dev, bool prof, void* stream, bool skip_text_kv = false) {
  const char* KV = WS + p->kv_base;
  auto P = [&](const Ref& r) { return resolve(r, W, WS, L, T, O, PL, KV); };
  size_t opi = 0;
  const bool f16 = (u->is_mmdit ? u->mcfg.dtype : u->cfg.dtype) == 1;       // (VAE / CLIP creators mirror dtype into cfg)
  const bool f32 = !u->is_mmdit && u->cfg.dtype >= 2;                        // fp32-storage modes (SD-v1.4 UNet plans only)
  const bool x3 = f32 && u->cfg.dtype == 3;                                  // ... with bf16x3 contractions (sdn_gemm_x3 / sdn_attention_x3)
  for (const Op& o : p->ops) {
    int rc = SDN_OK;
    if (skip_text_kv && o.text_kv) { ++opi; continue; }       // its output of the previous forward stands (same text version)
    if (prof) (void)hipEventRecord(u->ev[2 * opi], (hipStream_t)stream);
    if (f32) {                                                               // same plan, fp32 operators (sdn_f32.hip)
      switch (o.kind) {
        case OP_TEMB:
          rc = sdn_temb_f32(timestep, t_dev, o.batch, o.c1, (void*)P(o.out), stream);
          break;
        case OP_CONV_IN:
          rc = sdn_conv_in_f32((const float*)P(o.a), P(o.w), (const float*)P(o.bias), o.batch, o.c1, o.hw, o.hw, o.c2, (void*)P(o.out), stream);
          break;
        case OP_GEMM:
          if (o.x3t) { rc = launch_x3t_gemm(o, P(o.a), P(o.w), (const float*)P(o.bias), (const float*)P(o.rowbias), P(o.residual), (void*)P(o.out), stream); break; }
          rc = (o.ln || o.gd.split_k > 1) ? SDN_E_INVALID
               : (x3 ? sdn_gemm_x3 : sdn_gemm_f32)(&o.gd, P(o.a), P(o.a2), P(o.w), (const float*)P(o.bias), (const float*)P(o.rowbias),
                                                   (const float*)P(o.rowgate), P(o.residual), (void*)P(o.out), stream);
          break;
        case OP_SPLIT3:
          rc = sdn_split3((const float*)P(o.a), (const float*)P(o.a2), o.rows, o.c1, o.c2, (void*)P(o.out), stream);
          break;
        case OP_GN:
          rc = (o.tri_out ? sdn_groupnorm_f32_triple : sdn_groupnorm_f32)(P(o.a), P(o.a2), o.batch, o.hw, o.c1, o.c2, o.groups, o.eps, o.silu,
                                                                         (const float*)P(o.w), (const float*)P(o.bias), (void*)P(o.out),
                                                                         (float*)P(o.aux), stream);
          break;
        case OP_LN:
          rc = o.mod ? SDN_E_INVALID
                     : (o.tri_out ? sdn_layernorm_f32_triple : sdn_layernorm_f32)(P(o.a), o.rows, o.c1, o.eps, (const float*)P(o.w),
                                                                                  (const float*)P(o.bias), (void*)P(o.out), stream);
          break;
        case OP_ATTN:
          if (o.pair_in == 1) {                        // ld = 2 x (qkv width) bf16 elements, lo plane = one width on
            rc = sdn_attention_x3_pairs(P(o.a), P(o.a) + (size_t)o.ldo * 2, P(o.a) + (size_t)o.ldo * 4, o.ldq, o.ldq, (void*)P(o.out), o.batch, o.heads,
                                        o.nq, o.nk, o.hd, 2 * o.ldq, 2 * o.ldk, 2 * o.ldv, o.ldo, o.scale, o.tri_out, stream);
            break;
          }
          if (o.pair_in == 2) {                        // q rows [hi(C) | lo(C)]; k / v = column blocks 0 / C of the rows [hi(2C) | lo(2C)]
            rc = sdn_attention_x3_pairs(P(o.a), P(o.k), P(o.k) + (size_t)o.ldo * 2, o.ldq, o.ldk, (void*)P(o.out), o.batch, o.heads,
                                        o.nq, o.nk, o.hd, 2 * o.ldq, 2 * o.ldk, 2 * o.ldv, o.ldo, o.scale, o.tri_out, stream);
            break;
          }
          rc = o.n1 > 0 ? SDN_E_INVALID
                        : (o.tri_out ? sdn_attention_x3_triple : (x3 ? sdn_attention_x3 : sdn_attention_f32))(
                              P(o.a), P(o.k), P(o.v), (void*)P(o.out), o.batch, o.heads, o.nq, o.nk, o.hd, o.ldq, o.ldk, o.ldv, o.ldo, o.scale, stream);
          break;
        case OP_REPEAT:
          rc = sdn_repeat(P(o.a), (size_t)o.rows, o.c1, (void*)P(o.out), stream);
          break;
        case OP_CLIP_EMBED:
          rc = sdn_clip_embed_f32((const int32_t*)P(o.a), P(o.w), P(o.bias), o.rows, o.hw, o.c1, o.c2, (void*)P(o.out), stream);
          break;
        case OP_MATTN:                               // 1.7 % of the encoder's FLOPs: exact f32 products in both fp32-storage modes
          rc = sdn_masked_attention_f32(P(o.a), P(o.k), P(o.v), (void*)P(o.out), (const int32_t*)u->clip_mask, 1, o.batch, o.heads,
                                        o.nq, o.hd, o.ldq, o.ldk, o.ldv, o.ldo, o.scale, stream);
          break;
        default:
          rc = SDN_E_INVALID;
      }
      if (prof) (void)hipEventRecord(u->ev[2 * opi + 1], (hipStream_t)stream);
      ++opi;
      if (rc != SDN_OK) return rc;
      continue;
    }
    switch (o.kind) {
      case OP_TEMB:
        if (t_dev) rc = sdn_temb_from_device(f16 ? 1 : 0, t_dev, o.batch, o.c1, (void*)P(o.out), stream);
        else rc = (f16 ? sdn_timestep_embed_f16 : sdn_timestep_embed_bf16)(timestep, o.batch, o.c1, (void*)P(o.out), stream);
        break;
      case OP_CONV_IN:
        rc = (f16 ? sdn_conv_in_f16 : sdn_conv_in_bf16)((const float*)P(o.a), P(o.w), (const float*)P(o.bias), o.batch, o.c1, o.hw, o.hw, o.c2,
                              (void*)P(o.out), stream);
        break;
      case OP_ROWSTATS:
        rc = (f16 ? sdn_row_stats_f16 : sdn_row_stats_bf16)(P(o.a), o.rows, o.c1, o.eps, (float*)P(o.out), stream);
        break;
      case OP_FFN:
        rc = sdn_ffn_geglu_fused(f16 ? 1 : 0, o.rows, o.c1, P(o.a), (const float*)P(o.ln_stats), P(o.w), (const float*)P(o.ln_c),
                                 (const float*)P(o.ln_d), P(o.a2), (const float*)P(o.bias), P(o.residual), (void*)P(o.out),
                                 o.col.space != SP_NONE ? (float*)P(o.col) : nullptr, stream);
        break;
      case OP_GEMM:
        if (o.ln) {
          rc = (f16 ? sdn_gemm_ln_f16 : sdn_gemm_ln_bf16)(&o.gd, P(o.a), P(o.w), (const float*)P(o.ln_c), (const float*)P(o.ln_d), o.eps,
                                                          (const float*)P(o.ln_stats), (void*)P(o.out), stream);
          break;
        }
        if (o.col.space != SP_NONE && o.gd.split_k <= 1) {
          rc = (f16 ? sdn_gemm_stats_f16 : sdn_gemm_stats_bf16)(&o.gd, P(o.a), P(o.a2), P(o.w), (const float*)P(o.bias),
                                                                (const float*)P(o.rowbias), P(o.residual), (void*)P(o.out),
                                                                (float*)P(o.col), stream);
          break;
        }
        if (o.gd.split_k > 1) {
          rc = (f16 ? sdn_gemm_splitk_f16 : sdn_gemm_splitk_bf16)(&o.gd, P(o.a), P(o.a2), P(o.w), (const float*)P(o.bias),
                                                                  (const float*)P(o.rowbias), (const float*)P(o.rowgate), P(o.residual),
                                                                  (void*)P(o.out), (void*)P(o.aux), (size_t)o.rows, stream);
          break;
        }
        rc = (f16 ? sdn_gemm_f16 : sdn_gemm_bf16)(&o.gd, P(o.a), P(o.a2), P(o.w), (const float*)P(o.bias), (const float*)P(o.rowbias),
                           (const float*)P(o.rowgate), P(o.residual), (void*)P(o.out), stream);
        break;
      case OP_GN:
        if (o.cols1.space != SP_NONE) {
          rc = (f16 ? sdn_groupnorm_cols_f16 : sdn_groupnorm_cols_bf16)(P(o.a), P(o.a2), o.batch, o.hw, o.c1, o.c2, o.groups, o.eps,
                                                                        o.silu, (const float*)P(o.w), (const float*)P(o.bias),
                                                                        (void*)P(o.out), (float*)P(o.aux), (const float*)P(o.cols1),
                                                                        (const float*)P(o.cols2), stream);
          break;
        }
        rc = (f16 ? sdn_groupnorm_f16 : sdn_groupnorm_bf16)(P(o.a), P(o.a2), o.batch, o.hw, o.c1, o.c2, o.groups, o.eps, o.silu,
                                (const float*)P(o.w), (const float*)P(o.bias), (void*)P(o.out), (float*)P(o.aux), stream);
        break;
      case OP_CLIP_EMBED:
        rc = sdn_clip_embed(f16 ? 1 : 0, (const int32_t*)P(o.a), P(o.w), P(o.bias), o.rows, o.hw, o.c1, o.c2, (void*)P(o.out), stream);
        break;
      case OP_MATTN:
        rc = sdn_masked_attention(f16 ? 1 : 0, P(o.a), P(o.k), P(o.v), (void*)P(o.out), (const int32_t*)u->clip_mask, 1, o.batch,
                                  o.heads, o.nq, o.hd, o.ldq, o.ldk, o.ldv, o.ldo, o.scale, stream);
        break;
      case OP_REPEAT:
        rc = sdn_repeat(P(o.a), (size_t)o.rows, o.c1, (void*)P(o.out), stream);
        break;
      case OP_LATENT_MIX:
        rc = sdn_latent_mix((const float*)P(o.a), (const float*)P(o.w), (const float*)P(o.bias), o.batch, o.c1, o.hw,
                            o.mod ? o.scale : timestep /* decoder: the caller's latent_scale */,
                            (float*)P(o.out), stream);
        break;
      case OP_SOFTMAX:
        rc = sdn_softmax_rows(f16 ? 1 : 0, (const float*)P(o.a), o.c1, o.rows, o.c1, o.scale, (void*)P(o.out), o.c1, stream);
        break;
      case OP_TRANSPOSE:
        rc = sdn_transpose16(P(o.a), (int)o.rows, o.c1, o.ldq, (void*)P(o.out), o.ldo, stream);
        break;
      case OP_PATCHIFY:
        rc = (f16 ? sdn_patchify_f16 : sdn_patchify_bf16)((const float*)P(o.a), o.batch, o.c1, o.hw, o.hw, o.patch,
                                                          (void*)P(o.out), stream);
        break;
      case OP_UNPATCHIFY:
        rc = sdn_unpatchify_f32((const float*)P(o.a), o.batch, o.c1, o.hw, o.hw, o.patch, (float*)P(o.out), stream);
        break;
      case OP_LN:
        if (o.mod) {
          rc = (f16 ? sdn_layernorm_mod_f16 : sdn_layernorm_mod_bf16)(P(o.a), o.rows, o.c1, o.eps, (const float*)P(o.w),
                                                                      (const float*)P(o.bias), o.ld_mod, o.hw,
                                                                      (void*)P(o.out), stream);
          break;
        }
        rc = (f16 ? sdn_layernorm_f16 : sdn_layernorm_bf16)(P(o.a), o.rows, o.c1, o.eps, (const float*)P(o.w), (const float*)P(o.bias),
                                (void*)P(o.out), stream);
        break;
      case OP_ATTN:
        if (o.n1 > 0) {
          sdn_attn_segment2 s2{P(o.q2), P(o.k2), P(o.v2), (void*)P(o.out2), o.n1, o.ldq, o.ldk, o.ldv, o.ldo};
          rc = sdn_joint_attention(f16 ? 1 : 0, P(o.a), P(o.k), P(o.v), (void*)P(o.out), &s2, o.batch, o.heads, o.nq, o.hd,
                                   o.ldq, o.ldk, o.ldv, o.ldo, o.scale, stream);
          break;
        }
        rc = (f16 ? sdn_attention_f16 : sdn_attention_bf16)(P(o.a), P(o.k), P(o.v), (void*)P(o.out), o.batch, o.heads, o.nq, o.nk, o.hd, o.ldq,
                                o.ldk, o.ldv, o.ldo, o.scale, stream);
        break;
    }
    if (prof) (void)hipEventRecord(u->ev[2 * opi + 1], (hipStream_t)stream);
    ++opi;
    if (rc != SDN_OK) return rc;
  }
  return SDN_OK;
}


static int run_plan(sdn_unet* u, const void* weights, const float* latents, float timestep, const void* text,
                    const void* pooled, float* out, int32_t batch, void* workspace, size_t workspace_bytes, void* stream) {
  if (!u || !weights || !latents || !text || !out || !workspace || batch <= 0) return SDN_E_INVALID;
  Plan* p = get_plan(u, batch);
  if (p->ws_bytes < 0) return SDN_E_INVALID;                      // e.g. batch not a multiple of latent_repeat
  if (workspace_bytes < (size_t)p->ws_bytes) return SDN_E_WORKSPACE;
  const char* W = (const char*)weights; const char* WS = (const char*)workspace;
  const char* L = (const char*)latents; const char* T = (const char*)text; const char* O = (const char*)out;
  const char* PL = (const char*)pooled;
  const bool prof = u->profile_next;
  if (prof) {                                    // opt-in diagnostics: HIP events around every launch of this forward
    u->profile_next = false;
    while (u->ev.size() < 2 * p->ops.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return SDN_E_LAUNCH; u->ev.push_back(e); }
    u->profiled_batch = batch;
  }
  hipStream_t hs = (hipStream_t)stream;
  if (u->use_graph && !prof && !u->is_vae && !u->is_clip && p->tscalar_off >= 0) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(hs, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
      // Graph mode: small batches are launch-bound (~850 launches per forward); the forward is captured once per
      // (batch, operand addresses) and replayed.  The timestep is the only per-step scalar: it is stored to the
      // workspace by one ordinary launch and k_temb reads it from there.
      float* t_dev = (float*)(WS + p->tscalar_off);
      int rc = sdn_set_scalar(t_dev, timestep, stream);
      if (rc != SDN_OK) return rc;
      const sdn_unet::GraphKey key{batch, weights, latents, text, pooled, out, workspace};
      auto it = u->graphs.find(key);
      if (it == u->graphs.end()) {
        if (u->graphs.size() >= 16) {                          // operands keep moving: graphs do not pay, stop hoarding
          (void)hipStreamSynchronize(hs);                      // a replay may still be executing on the launch stream
          for (auto& kv : u->graphs) (void)hipGraphExecDestroy(kv.second);
          u->graphs.clear();
        }
        hipGraph_t graph = nullptr;
        if (!u->cap_stream && hipStreamCreateWithFlags(&u->cap_stream, hipStreamNonBlocking) != hipSuccess) return SDN_E_LAUNCH;
        if (hipStreamBeginCapture(u->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return SDN_E_LAUNCH;
        rc = launch_ops(u, p, W, WS, L, T, O, PL, timestep, t_dev, false, (void*)u->cap_stream);   // records, does not run
        const hipError_t ec = hipStreamEndCapture(u->cap_stream, &graph);
        if (rc != SDN_OK || ec != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); return rc != SDN_OK ? rc : SDN_E_LAUNCH; }
        hipGraphExec_t exec = nullptr;
        const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess || !exec) return SDN_E_LAUNCH;
        it = u->graphs.emplace(key, exec).first;
      }
      u->kv_version = 0;                                       // the replay rewrites every K / V slot: nothing to reuse afterwards
      return hipGraphLaunch(it->second, hs) == hipSuccess ? SDN_OK : SDN_E_LAUNCH;
    }
  }
  // text K / V reuse (ordinary launches only: a captured graph holds a fixed op list)
  const bool declared = u->text_version != 0 && !u->is_mmdit && !u->is_vae && !u->is_clip && u->subbatch_bytes == 0;
  const bool skip = declared && !prof && u->kv_version == u->text_version && u->kv_batch == batch && u->kv_w == weights &&
                    u->kv_text == text && u->kv_ws == workspace;
  const int rc_l = launch_ops(u, p, W, WS, L, T, O, PL, timestep, nullptr, prof, stream, skip);
  if (rc_l == SDN_OK && declared) { u->kv_version = u->text_version; u->kv_batch = batch; u->kv_w = weights; u->kv_text = text; u->kv_ws = workspace; }
  else if (!declared) u->kv_version = 0;
  return rc_l;
}

void sdn_unet_set_text_version(sdn_unet* u, uint64_t version) {
  if (!u) return;
  u->text_version = version;
  if (version == 0) u->kv_version = 0;                         // undeclared: the cached K / V are dropped NOW, not at the next plain forward
}

void sdn_unet_profile_next(sdn_unet* u) { if (u) u->profile_next = true; }

// Configuration calls (not on the hot path) drop the cached graphs; a replay may still be in flight on whatever stream the
// caller launched it, so the device is drained first.
static void drop_graphs(sdn_unet* u) {
  if (u->graphs.empty()) return;
  (void)hipDeviceSynchronize();
  for (auto& kv : u->graphs) (void)hipGraphExecDestroy(kv.second);
  u->graphs.clear();
}

void sdn_unet_set_split_k(sdn_unet* u, int32_t on) {
  if (!u || u->split_k == (on != 0)) return;
  if (!u->is_mmdit && !u->is_vae && u->cfg.dtype >= 2) return;   // fp32-storage modes have no split-K form
  u->split_k = on != 0;
  drop_graphs(u);
  u->plans.clear();                                            // plans are rebuilt with / without partial buffers
}

void sdn_unet_set_graph_mode(sdn_unet* u, int32_t on) {
  if (!u) return;
  u->use_graph = on != 0;
  if (!on) drop_graphs(u);
}

// Undeclared tuning hook (tools/): size threshold of the transformer sub-batching; rebuilds the plans.
// Undeclared A/B hook: run the BasicTransformerBlock LayerNorms as separate kernels again (the derived regions stay).
// A/B and tests: dtype-3 plans on the f32-staging k_gemm_x3 everywhere (0) or with triple operands on the LDS-DMA tiles (1, default).
// The expanded weight regions stay registered either way (the manifest does not change); query the workspace size again.
extern "C" void sdn_debug_set_x3_expand(sdn_unet* u, int on) {
  if (!u || u->x3_expand == (on != 0)) return;
  u->x3_expand = on != 0;
  drop_graphs(u);
  u->plans.clear();
}

extern "C" void sdn_debug_set_res_pre(sdn_unet* u, int on) {
  if (!u || u->res_pre == (on != 0)) return;
  u->res_pre = on != 0;
  drop_graphs(u);
  u->plans.clear();
}

extern "C" void sdn_debug_set_ln_fold(sdn_unet* u, int on) {
  if (!u) return;
  u->ln_fold = on != 0;
  drop_graphs(u);
  u->plans.clear();
}

extern "C" void sdn_debug_set_ln_prepass_all(sdn_unet* u, int on) {
  if (!u) return;
  u->ln_prepass_all = on;
  drop_graphs(u);
  u->plans.clear();
}

extern "C" void sdn_debug_set_ffn_fuse(sdn_unet* u, int on) {   // one-launch GEGLU feed-forward (C = 320) on / off: A/B and equality tests
  if (!u) return;
  u->ffn_fuse = on != 0;
  drop_graphs(u);
  u->plans.clear();
}
extern "C" void sdn_debug_set_ff_fuse(sdn_unet* u, int on) {
  if (!u) return;
  u->ff_fuse = on != 0;
  drop_graphs(u);
  u->plans.clear();
}

extern "C" void sdn_debug_set_ffn_own_stats(sdn_unet* u, int on) {
  if (!u) return;
  u->ffn_own_stats = on != 0;
  drop_graphs(u);
  u->plans.clear();
}

extern "C" void sdn_debug_set_x3_pairs(sdn_unet* u, int on) {
  if (!u) return;
  u->x3_pairs = on != 0;
  drop_graphs(u);
  u->plans.clear();
}

extern "C" void sdn_debug_set_gn_fuse(sdn_unet* u, int on) {
  if (!u) return;
  u->gn_fuse = on != 0;
  drop_graphs(u);
  u->plans.clear();
}

extern "C" void sdn_debug_set_subbatch_bytes(sdn_unet* u, long long bytes) {
  if (!u || u->is_mmdit) return;
  u->subbatch_bytes = bytes;
  drop_graphs(u);
  u->plans.clear();
}

// Undeclared debug hook (tools/profile_ops.py): per-launch rows of the profiled forward, in plan order.
// out[i*6 + {0..5}] = {ms, flops, bytes, M, N, K}; labels[i*24..] = kernel label.  Returns the op count.
extern "C" int sdn_debug_profile_ops(sdn_unet* u, double* out, char* labels, int max_ops) {
  if (!u || !out || !labels || u->profiled_batch <= 0) return -1;
  Plan* p = get_plan(u, u->profiled_batch);
  int n = 0;
  for (size_t i = 0; i < p->ops.size() && n < max_ops; ++i, ++n) {
    if (hipEventSynchronize(u->ev[2 * i + 1]) != hipSuccess) return -2;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, u->ev[2 * i], u->ev[2 * i + 1]) != hipSuccess) return -2;
    const Op& o = p->ops[i];
    out[n * 6 + 0] = ms; out[n * 6 + 1] = o.flops; out[n * 6 + 2] = o.bytes;
    out[n * 6 + 3] = o.kind == OP_GEMM ? o.gd.M : (o.kind == OP_ATTN ? o.nq : o.rows);
    out[n * 6 + 4] = o.kind == OP_GEMM ? o.gd.N : (o.kind == OP_ATTN ? o.nk : o.c1);
    out[n * 6 + 5] = o.kind == OP_GEMM ? o.gd.K : (o.kind == OP_ATTN ? o.hd : o.c2);
    memcpy(labels + n * 24, o.label, 24);
  }
  return n;
}

int sdn_unet_profile_read(sdn_unet* u, sdn_profile_row* rows, int32_t max_rows) {
  if (!u || !rows || max_rows <= 0 || u->profiled_batch <= 0) return SDN_E_INVALID;
  Plan* p = get_plan(u, u->profiled_batch);
  if (u->ev.size() < 2 * p->ops.size()) return SDN_E_INVALID;
  int n = 0;
  for (size_t i = 0; i < p->ops.size(); ++i) {
    if (hipEventSynchronize(u->ev[2 * i + 1]) != hipSuccess) return SDN_E_LAUNCH;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, u->ev[2 * i], u->ev[2 * i + 1]) != hipSuccess) return SDN_E_LAUNCH;
    const Op& o = p->ops[i];
    int r = 0;
    for (; r < n; ++r) if (strcmp(rows[r].kernel, o.label) == 0) break;
    if (r == n) {
      if (n == max_rows) return SDN_E_INVALID;
      memset(&rows[r], 0, sizeof(rows[r]));
      snprintf(rows[r].kernel, sizeof(rows[r].kernel), "%s", o.label);
      ++n;
    }
    rows[r].launches += 1; rows[r].ms += ms; rows[r].flops += o.flops; rows[r].bytes += o.bytes;
  }
  return n;
}

}  // extern "C"
