// C-ABI implementation of the training step (include/streamformer_hip.h, "training step" section):
// flat fp32 parameter layout, bf16 working weights, forward with saved activations, staged backward,
// fused AdamW.  Reference: autograd through TimesformerMultiTaskingModelSigLIP.forward
// (modeling:1299-1354; layer :934-1004; head :1141-1154; embeddings :413-457) under the step
// semantics of tools/finetune_tools.py:395-573 and the optimizer grouping of optim_factory.py:59-104.
#include "sf_internal.h"
#include "sf_common.h"
#include "sf_switches.h"
#include "sf_train.h"
#include "sf_pool_head.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static const int kRank = 32;   // modeling:1280-1281

struct TParam {
  std::string name;
  int64_t shape[4];
  int ndim;
  size_t numel, off;
  bool trainable, decay;
};

struct TLin {                    // y = x W^T + b over the flat buffer
  int N = 0, K = 0;
  int pw = -1; size_t pw_off = 0;   // weight: param index + float offset inside it
  int pb = -1; size_t pb_off = 0;   // bias (pb < 0: none)
  int pla = -1, plb = -1;           // LoRA factors (spatial qkv / output.dense)
  int pgate = -1;                   // temporal_dense: forward weight = tanh(gate) * W
  bf16_t* w = nullptr;              // [N,K] working copy
  bf16_t* wT = nullptr;             // [K,N] for the input-gradient GEMM
  bf16_t* la_bf = nullptr;          // LoRA A [r,K] and B^T [r,N] (factor gradients through skinny GEMMs)
  bf16_t* lbT_bf = nullptr;
  float* bias_scaled = nullptr;     // tanh(gate) * b
  bool need_wT = true;
};

struct TLayer {
  int gate, ln_t_g, ln_t_b, ln_b_g, ln_b_b, ln_a_g, ln_a_b;
  TLin t_qkv, t_out, t_dense, s_qkv, s_out, up, down;
  size_t seg_off, seg_end;
  // temporal_dense o temporal_attention.output.dense as one projection (drop rates 0): W_f = tanh(g) W_d W_o [D, D], its transpose,
  // b_f = tanh(g) (W_d b_o + b_d); refreshed with the working weights (sf_launch_fuse_temporal)
  bf16_t* wf = nullptr; bf16_t* wfT = nullptr; float* bf = nullptr;
};

struct sf_trainer {
  sf_config cfg;
  int device;
  int D, I, L, heads, N, Kp, C, P;
  bool lora, freeze;
  std::vector<TParam> params;
  size_t total = 0, n_train = 0;
  int p_pos, p_time, p_probe, p_inw, p_inb, post_g, post_b, hln_g, hln_b;
  TLin patch, head_kv, head_out, fc1, fc2;
  std::vector<TLayer> layers;
  size_t emb_off, emb_end, tail_off, tail_end;
  // device state
  int* seg_end = nullptr;
  unsigned char* seg_decay = nullptr;
  unsigned char* seg_train = nullptr;
  int nseg = 0;
  bf16_t* arena = nullptr;
  float* farena = nullptr;          // scaled biases, head query, reduction scratch
  float* head_q = nullptr;
  float* head_u = nullptr;          // pooling head: U_h = Wk_h^T q_h, fp32 [16, D] (+ hi / lo bf16 planes), refreshed with the weights
  bf16_t* head_u_hi = nullptr; bf16_t* head_u_lo = nullptr;
  float* red_partial = nullptr;
  SfPrepJob* prep_jobs = nullptr;   // device table for sf_trainer_sync_weights
  int n_prep_jobs = 0, prep_tiles = 0;
  SfFuseJob* fuse_jobs = nullptr;   // device table of the fused temporal projections (one per layer)
  bool f_tfuse = false;             // the last forward ran the temporal branch's two projections as one (its backward follows)
  const float* params_dev = nullptr;
  int fB = 0, fT = 0;               // geometry of the last forward (0 = none)
  const float* dp_scales = nullptr; // drop_path factors of the next forward (device, caller-owned), nullptr = none
  int dp_B = 0, dp_T = 0;
  const float* f_dp = nullptr;      // the factors the last forward used: its backward applies the same ones
  // dropout of the NEXT forward (sf_trainer_set_dropout) and of the last one (its backward replays the same counter-based masks)
  float drop_hidden = 0.f, drop_attn = 0.f, f_drop_hidden = 0.f, f_drop_attn = 0.f;
  unsigned drop_seed = 0u, f_drop_seed = 0u;
  int n_extra = 0, extra_seg0 = 0;  // the scalar slots are the last n_extra trainable segments
  bool extra_steps_set = false;
  int extra_steps[64] = {};
  int* guard_flag = nullptr;          // non-finite guard (sf_trainer_set_nonfinite_guard): device int32[2], caller-owned
  const float* guard_loss = nullptr;  // optional device loss scalar checked next to the gradient's sum of squares
  float* guard_sumsq = nullptr;       // library-owned scalar the guard's own sum-of-squares pass writes
  // The rank-32 LoRA gradients of a layer (two projections, two small weight-gradient GEMMs + reductions per adapted Linear:
  // ~1.8 ms per step at 8 clips, all bandwidth- / latency-bound launches that depend on nothing downstream) run on a library-owned
  // side stream, forked from and joined into the caller's stream inside every layer: they fill the gaps of the MFMA-bound chain.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // D x D gradient algebra of the fused temporal projections (six small launches per layer): also on the side stream, joined one
  // layer LATER (its inputs G1 / cs alternate between two buffers by layer parity), and before sf_trainer_backward returns
  hipEvent_t ev_small[2] = {nullptr, nullptr};
  int small_pending = 0;              // bit p: ev_small[p] has been recorded and not yet waited for
  int side_state = 0;                 // 0 = not tried, 1 = available, -1 = unavailable (creation failed / SF_TRAIN_SIDE_STREAM=0)
};

static int add_param(sf_trainer* t, const std::string& name, std::initializer_list<int64_t> shape, bool trainable) {
  TParam p;
  p.name = name;
  p.ndim = (int)shape.size();
  p.numel = 1;
  int i = 0;
  for (int64_t d : shape) { p.shape[i++] = d; p.numel *= (size_t)d; }
  for (; i < 4; ++i) p.shape[i] = 1;
  p.off = 0;
  p.trainable = trainable;
  // optim_factory.py:72-77: 1-D parameters and "*.bias" are not decayed; everything else (incl. 0-dim) is
  const bool is_bias = name.size() >= 5 && name.compare(name.size() - 5, 5, ".bias") == 0;
  p.decay = !(p.ndim == 1 || is_bias);
  t->params.push_back(p);
  return (int)t->params.size() - 1;
}

static void lin_params(sf_trainer* t, TLin* l, const std::string& prefix, int N, int K, bool bias, bool trainable) {
  l->N = N; l->K = K;
  l->pw = add_param(t, prefix + ".weight", {N, K}, trainable);
  l->pb = bias ? add_param(t, prefix + ".bias", {N}, trainable) : -1;
}

static void free_trainer_device(sf_trainer* t) {
  if (t->seg_end) (void)hipFree(t->seg_end);
  if (t->seg_decay) (void)hipFree(t->seg_decay);
  if (t->seg_train) (void)hipFree(t->seg_train);
  if (t->arena) (void)hipFree(t->arena);
  if (t->farena) (void)hipFree(t->farena);
  if (t->prep_jobs) (void)hipFree(t->prep_jobs);
  if (t->fuse_jobs) (void)hipFree(t->fuse_jobs);
}

extern "C" int sf_trainer_create(const sf_config* cfg, int device, int freeze_spatial, int n_extra, sf_trainer** out) {
  if (!cfg || !out) return sf_set_err(SF_ERR_INVALID, "sf_trainer_create: null argument");
  const sf_config& c = *cfg;
  if (c.hidden_size % c.num_attention_heads || c.hidden_size / c.num_attention_heads != 64)
    return sf_set_err(SF_ERR_INVALID, "head_dim must be 64 (hidden %d, heads %d)", c.hidden_size, c.num_attention_heads);
  if (c.hidden_act != 0) return sf_set_err(SF_ERR_INVALID, "training supports hidden_act=gelu only");
  if (c.image_size % c.patch_size) return sf_set_err(SF_ERR_INVALID, "image_size %% patch_size != 0");
  if (n_extra < 0 || n_extra > 64) return sf_set_err(SF_ERR_INVALID, "n_extra out of range");
  if (hipSetDevice(device) != hipSuccess) return sf_set_err(SF_ERR_HIP, "hipSetDevice(%d) failed", device);
  sf_trainer* t = new sf_trainer();
  t->cfg = c; t->device = device;
  t->D = c.hidden_size; t->I = c.intermediate_size; t->L = c.num_hidden_layers; t->heads = c.num_attention_heads;
  t->C = c.num_channels; t->P = c.patch_size;
  t->N = (c.image_size / c.patch_size) * (c.image_size / c.patch_size);
  t->Kp = t->C * t->P * t->P;
  t->lora = c.add_lora_spatial != 0;
  t->freeze = freeze_spatial != 0;
  const int D = t->D, I = t->I;
  if (t->N > 224) { delete t; return sf_set_err(SF_ERR_INVALID, "%d patches per frame; kernels handle <= 224", t->N); }
  if ((D % 64) || (I % 64) || (t->Kp % 64)) { delete t; return sf_set_err(SF_ERR_INVALID, "hidden/intermediate/patch sizes must be multiples of 64"); }

  // ---- parameter list in model order (names = reference state_dict keys) -----------
  t->p_pos = add_param(t, "embeddings.position_embeddings", {1, t->N, D}, true);
  t->p_time = add_param(t, "embeddings.time_embeddings", {1, c.num_frames, D}, true);
  t->patch.N = D; t->patch.K = t->Kp; t->patch.need_wT = false;
  t->patch.pw = add_param(t, "embeddings.patch_embeddings.projection.weight", {D, t->C, t->P, t->P}, true);
  t->patch.pb = add_param(t, "embeddings.patch_embeddings.projection.bias", {D}, true);
  t->layers.resize(t->L);
  for (int i = 0; i < t->L; ++i) {
    TLayer& l = t->layers[i];
    const std::string p = "encoder.layer." + std::to_string(i) + ".";
    const bool sp_train = !t->freeze;
    l.gate = add_param(t, p + "temporal_attention_gating", {}, true);
    l.ln_t_g = add_param(t, p + "temporal_layernorm.weight", {D}, true);
    l.ln_t_b = add_param(t, p + "temporal_layernorm.bias", {D}, true);
    lin_params(t, &l.t_qkv, p + "temporal_attention.attention.qkv", 3 * D, D, c.qkv_bias != 0, true);
    lin_params(t, &l.t_out, p + "temporal_attention.output.dense", D, D, true, true);
    lin_params(t, &l.t_dense, p + "temporal_dense", D, D, true, true);
    l.t_dense.pgate = l.gate;
    l.ln_b_g = add_param(t, p + "layernorm_before.weight", {D}, true);
    l.ln_b_b = add_param(t, p + "layernorm_before.bias", {D}, true);
    lin_params(t, &l.s_qkv, p + "attention.attention.qkv", 3 * D, D, c.qkv_bias != 0, sp_train);
    if (t->lora) {
      l.s_qkv.pla = add_param(t, p + "attention.attention.qkv_lora_a.weight", {kRank, D}, true);
      l.s_qkv.plb = add_param(t, p + "attention.attention.qkv_lora_b.weight", {3 * D, kRank}, true);
    }
    lin_params(t, &l.s_out, p + "attention.output.dense", D, D, true, sp_train);
    if (t->lora) {
      l.s_out.pla = add_param(t, p + "attention.output.dense_lora_a.weight", {kRank, D}, true);
      l.s_out.plb = add_param(t, p + "attention.output.dense_lora_b.weight", {D, kRank}, true);
    }
    l.ln_a_g = add_param(t, p + "layernorm_after.weight", {D}, true);
    l.ln_a_b = add_param(t, p + "layernorm_after.bias", {D}, true);
    lin_params(t, &l.up, p + "intermediate.dense", I, D, true, true);
    lin_params(t, &l.down, p + "output.dense", D, I, true, true);
  }
  t->post_g = add_param(t, "post_layernorm.weight", {D}, true);
  t->post_b = add_param(t, "post_layernorm.bias", {D}, true);
  t->p_probe = add_param(t, "head.probe", {1, 1, D}, true);
  t->p_inw = add_param(t, "head.attention.in_proj_weight", {3 * D, D}, true);
  t->p_inb = add_param(t, "head.attention.in_proj_bias", {3 * D}, true);
  t->head_kv.N = 2 * D; t->head_kv.K = D;
  t->head_kv.pw = t->p_inw; t->head_kv.pw_off = (size_t)D * D;
  t->head_kv.pb = t->p_inb; t->head_kv.pb_off = (size_t)D;
  lin_params(t, &t->head_out, "head.attention.out_proj", D, D, true, true);
  t->hln_g = add_param(t, "head.layernorm.weight", {D}, true);
  t->hln_b = add_param(t, "head.layernorm.bias", {D}, true);
  lin_params(t, &t->fc1, "head.mlp.fc1", I, D, true, true);
  lin_params(t, &t->fc2, "head.mlp.fc2", D, I, true, true);
  for (int i = 0; i < n_extra; ++i) add_param(t, "extra." + std::to_string(i), {}, true);

  // ---- offsets: trainable prefix in model order, frozen tail; 64-float alignment ---------------------
  size_t off = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (TParam& p : t->params) {
      if (p.trainable != (pass == 0)) continue;
      p.off = off;
      off += (p.numel + 63) & ~(size_t)63;
    }
    if (pass == 0) t->n_train = off;
  }
  t->total = off;
  auto seg_end_of = [&](int idx) { const TParam& p = t->params[idx]; return p.off + ((p.numel + 63) & ~(size_t)63); };
  t->emb_off = t->params[t->p_pos].off;
  t->emb_end = seg_end_of(t->patch.pb);
  for (int i = 0; i < t->L; ++i) {
    t->layers[i].seg_off = t->params[t->layers[i].gate].off;
    t->layers[i].seg_end = seg_end_of(t->layers[i].down.pb);
  }
  t->tail_off = t->params[t->post_g].off;
  t->tail_end = t->n_train;

  // ---- segment table for the optimizer ----------------------------------------------------------------
  {
    std::vector<const TParam*> order;
    for (const TParam& p : t->params) order.push_back(&p);
    std::vector<int> ends(order.size());
    std::vector<unsigned char> dec(order.size()), tr(order.size());
    // params are already offset-sorted within each pass; build by offset
    std::vector<int> idx(order.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
    for (size_t i = 1; i < idx.size(); ++i)
      for (size_t j = i; j > 0 && order[idx[j]]->off < order[idx[j - 1]]->off; --j) std::swap(idx[j], idx[j - 1]);
    if (t->total >= ((size_t)1 << 31)) { delete t; return sf_set_err(SF_ERR_INVALID, "model too large for int32 segment offsets"); }
    for (size_t i = 0; i < idx.size(); ++i) {
      const TParam* p = order[idx[i]];
      ends[i] = (int)(p->off + ((p->numel + 63) & ~(size_t)63));
      dec[i] = p->decay; tr[i] = p->trainable;
    }
    t->nseg = (int)idx.size();
    int ntr = 0;
    for (const TParam& p : t->params) ntr += p.trainable ? 1 : 0;
    t->n_extra = n_extra;
    t->extra_seg0 = ntr - n_extra;   // trainable segments come first in offset order, the extras last among them
    if (hipMalloc(&t->seg_end, ends.size() * sizeof(int)) != hipSuccess || hipMalloc(&t->seg_decay, dec.size()) != hipSuccess ||
        hipMalloc(&t->seg_train, tr.size()) != hipSuccess) {
      free_trainer_device(t); delete t;
      return sf_set_err(SF_ERR_HIP, "hipMalloc failed (segment table)");
    }
    (void)hipMemcpy(t->seg_end, ends.data(), ends.size() * sizeof(int), hipMemcpyHostToDevice);
    (void)hipMemcpy(t->seg_decay, dec.data(), dec.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(t->seg_train, tr.data(), tr.size(), hipMemcpyHostToDevice);
  }

  // ---- bf16 working-weight arena -----------------------------------------------------------------------
  {
    std::vector<TLin*> lins = {&t->patch, &t->head_kv, &t->head_out, &t->fc1, &t->fc2};
    for (TLayer& l : t->layers) for (TLin* x : {&l.t_qkv, &l.t_out, &l.t_dense, &l.s_qkv, &l.s_out, &l.up, &l.down}) lins.push_back(x);
    size_t nb = 0, nf = 0;
    for (TLin* x : lins) {
      nb += ((size_t)x->N * x->K + 127) & ~(size_t)127;
      if (x->need_wT) nb += ((size_t)x->N * x->K + 127) & ~(size_t)127;
      if (x->pla >= 0) nb += (((size_t)kRank * x->K + 127) & ~(size_t)127) + (((size_t)kRank * x->N + 127) & ~(size_t)127);
      if (x->pgate >= 0) nf += ((size_t)x->N + 63) & ~(size_t)63;
    }
    nf += (size_t)D + 64 + 2048 + (size_t)16 * D;          // head query + reduction scratch + the head's folded key projection
    nb += (size_t)2 * 16 * D;
    nb += (size_t)t->L * 2 * D * D;                        // fused temporal projections: wf + wfT per layer
    nf += (size_t)t->L * D;                                //                             + b_f
    if (hipMalloc(&t->arena, nb * sizeof(bf16_t)) != hipSuccess || hipMalloc(&t->farena, nf * sizeof(float)) != hipSuccess) {
      free_trainer_device(t); delete t;
      return sf_set_err(SF_ERR_HIP, "hipMalloc failed (working weights, %zu bytes)", nb * 2);
    }
    bf16_t* bp = t->arena;
    float* fp = t->farena;
    for (TLin* x : lins) {
      const size_t n = ((size_t)x->N * x->K + 127) & ~(size_t)127;
      x->w = bp; bp += n;
      if (x->need_wT) { x->wT = bp; bp += n; }
      if (x->pla >= 0) {
        x->la_bf = bp; bp += ((size_t)kRank * x->K + 127) & ~(size_t)127;
        x->lbT_bf = bp; bp += ((size_t)kRank * x->N + 127) & ~(size_t)127;
      }
      if (x->pgate >= 0) { x->bias_scaled = fp; fp += ((size_t)x->N + 63) & ~(size_t)63; }
    }
    t->head_q = fp; fp += (size_t)D + 64;
    t->head_u = fp; fp += (size_t)16 * D;
    t->head_u_hi = bp; bp += (size_t)16 * D;
    t->head_u_lo = bp; bp += (size_t)16 * D;
    for (TLayer& l : t->layers) {
      l.wf = bp; bp += (size_t)D * D;
      l.wfT = bp; bp += (size_t)D * D;
      l.bf = fp; fp += (size_t)D;
    }
    t->red_partial = fp;
    // one-launch weight refresh: job table with offsets into the flat parameter buffer
    std::vector<SfPrepJob> jobs;
    int tiles = 0;
    auto push = [&](long w_off, long la, long lb, int rank, long gate, long bias, bf16_t* w_bf, bf16_t* wT_bf, float* bias_out, int N, int K) {
      SfPrepJob j;
      j.w_off = w_off; j.la_off = la; j.lb_off = lb; j.gate_off = gate; j.bias_off = bias;
      j.w_bf = w_bf; j.wT_bf = wT_bf; j.bias_out = bias_out; j.N = N; j.K = K; j.rank = rank; j.tile0 = tiles;
      tiles += ((N + SF_PREP_TILE - 1) / SF_PREP_TILE) * ((K + SF_PREP_TILE - 1) / SF_PREP_TILE);
      jobs.push_back(j);
    };
    auto off = [&](int idx, size_t extra = 0) -> long { return idx < 0 ? -1 : (long)(t->params[idx].off + extra); };
    for (TLin* x : lins) {
      push(off(x->pw, x->pw_off), off(x->pla), off(x->plb), kRank, off(x->pgate), x->pgate >= 0 ? off(x->pb, x->pb_off) : -1, x->w, x->wT,
           x->bias_scaled, x->N, x->K);
      if (x->pla >= 0) {
        push(off(x->pla), -1, -1, 0, -1, -1, x->la_bf, nullptr, nullptr, kRank, x->K);
        push(off(x->plb), -1, -1, 0, -1, -1, nullptr, x->lbT_bf, nullptr, x->N, kRank);
      }
    }
    t->n_prep_jobs = (int)jobs.size(); t->prep_tiles = tiles;
    if (hipMalloc(&t->prep_jobs, jobs.size() * sizeof(SfPrepJob)) != hipSuccess) {
      free_trainer_device(t); delete t;
      return sf_set_err(SF_ERR_HIP, "hipMalloc failed (prep table)");
    }
    (void)hipMemcpy(t->prep_jobs, jobs.data(), jobs.size() * sizeof(SfPrepJob), hipMemcpyHostToDevice);
    std::vector<SfFuseJob> fj;
    for (TLayer& l : t->layers) {
      SfFuseJob j;
      j.wd = l.t_dense.w; j.woT = l.t_out.wT; j.wf = l.wf; j.wfT = l.wfT; j.bf = l.bf;
      j.wd_off = off(l.t_dense.pw); j.bo_off = off(l.t_out.pb); j.bd_off = off(l.t_dense.pb); j.gate_off = off(l.gate);
      fj.push_back(j);
    }
    if (hipMalloc(&t->fuse_jobs, fj.size() * sizeof(SfFuseJob)) != hipSuccess) {
      free_trainer_device(t); delete t;
      return sf_set_err(SF_ERR_HIP, "hipMalloc failed (fuse table)");
    }
    (void)hipMemcpy(t->fuse_jobs, fj.data(), fj.size() * sizeof(SfFuseJob), hipMemcpyHostToDevice);
  }
  *out = t;
  return SF_OK;
}

extern "C" void sf_trainer_destroy(sf_trainer* t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  free_trainer_device(t);
  if (t->guard_sumsq) (void)hipFree(t->guard_sumsq);
  if (t->side) (void)hipStreamDestroy(t->side);
  if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
  if (t->ev_join) (void)hipEventDestroy(t->ev_join);
  for (hipEvent_t e : t->ev_small) if (e) (void)hipEventDestroy(e);
  delete t;
}

extern "C" int sf_trainer_num_params(const sf_trainer* t) { return t ? (int)t->params.size() : 0; }

extern "C" int sf_trainer_param_info(const sf_trainer* t, int index, char* name_out, int name_cap, int64_t* offset_out,
                                     int64_t* numel_out, int64_t* shape_out, int* ndim_out, int* trainable_out,
                                     int* decay_out) {
  if (!t || index < 0 || index >= (int)t->params.size()) return sf_set_err(SF_ERR_INVALID, "param index out of range");
  const TParam& p = t->params[index];
  if (name_out && name_cap > 0) snprintf(name_out, (size_t)name_cap, "%s", p.name.c_str());
  if (offset_out) *offset_out = (int64_t)p.off;
  if (numel_out) *numel_out = (int64_t)p.numel;
  if (shape_out) for (int i = 0; i < 4; ++i) shape_out[i] = p.shape[i];
  if (ndim_out) *ndim_out = p.ndim;
  if (trainable_out) *trainable_out = p.trainable;
  if (decay_out) *decay_out = p.decay;
  return SF_OK;
}

extern "C" int sf_trainer_total_floats(const sf_trainer* t, int64_t* total_out, int64_t* trainable_out) {
  if (!t) return sf_set_err(SF_ERR_INVALID, "null handle");
  if (total_out) *total_out = (int64_t)t->total;
  if (trainable_out) *trainable_out = (int64_t)t->n_train;
  return SF_OK;
}

extern "C" int sf_trainer_num_stages(const sf_trainer* t) { return t ? t->L + 2 : 0; }

extern "C" int sf_trainer_stage_range(const sf_trainer* t, int stage, int64_t* offset_out, int64_t* numel_out) {
  if (!t || stage < 0 || stage > t->L + 1) return sf_set_err(SF_ERR_INVALID, "stage out of range");
  size_t a, b;
  if (stage == 0) { a = t->tail_off; b = t->tail_end; }
  else if (stage == t->L + 1) { a = t->emb_off; b = t->emb_end; }
  else { const TLayer& l = t->layers[t->L - stage]; a = l.seg_off; b = l.seg_end; }
  if (offset_out) *offset_out = (int64_t)a;
  if (numel_out) *numel_out = (int64_t)(b - a);
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// working weights
// ------------------------------------------------------------------------------------------------
static inline const float* PP(const sf_trainer* t, const float* base, int idx, size_t extra = 0) {
  return idx < 0 ? nullptr : base + t->params[idx].off + extra;
}
static inline float* GG(const sf_trainer* t, float* grads, int idx, size_t extra = 0) {
  return (idx < 0 || !t->params[idx].trainable) ? nullptr : grads + t->params[idx].off + extra;
}
static inline const float* lin_bias(const sf_trainer* t, const TLin& l) {
  return l.pgate >= 0 ? l.bias_scaled : PP(t, t->params_dev, l.pb, l.pb_off);
}

extern "C" int sf_trainer_sync_weights(sf_trainer* t, const float* params_dev, sf_stream stream) {
  if (!t || !params_dev) return sf_set_err(SF_ERR_INVALID, "null argument");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(t->device));
  t->params_dev = params_dev;
  HIP_TRY(sf_launch_prep_weights_batched(params_dev, t->prep_jobs, t->n_prep_jobs, t->prep_tiles, s));
  // nn.MultiheadAttention scales q by head_dim^-0.5 after the in-projection (modeling:1145-1149)
  HIP_TRY(sf_launch_head_query(PP(t, params_dev, t->p_probe), PP(t, params_dev, t->p_inw), PP(t, params_dev, t->p_inb), 0.125f,
                               t->head_q, t->D, s));
  // the temporal branch's two projections as one (used by forwards without drop_path / hidden dropout): from the fresh bf16 copies
  HIP_TRY(sf_launch_fuse_temporal(params_dev, t->fuse_jobs, t->L, t->D, s));
  // the keys of the pooling head only meet that one query: U_h = Wk_h^T q_h (sf_pool_head.hip)
  HIP_TRY(sf_launch_pool_u(PP(t, params_dev, t->p_inw, (size_t)t->D * t->D), t->head_q, t->head_u, t->head_u_hi, t->head_u_lo, t->heads, t->D, s));
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
struct TCarver {
  char* base;
  size_t off = 0;
  explicit TCarver(void* b) : base((char*)b) {}
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct TSavedLayer {
  float *h1, *h2;
  bf16_t *ln_t, *tqkv, *ctx_t, *t_out, *ln_b, *sqkv, *ctx_s, *ln_a, *pre, *act;
  float* lse_s;      // spatial attention log-sum-exp [F, heads, N]
};
struct TWs {
  // saved by the forward
  bf16_t* patches; float* te_rows;
  std::vector<float*> h;               // L+1 residual snapshots
  std::vector<TSavedLayer> sl;
  bf16_t *xn, *pc, *hn, *hm_pre, *hm;
  float* attn_out;
  float *pz, *pprobs, *pml, *pzpart;   // pooling head: z_h = sum_n p_hn x_n [F, heads, D], the raw scores [F, heads, N], {max, sum} and partial sums per token split
  // backward scratch
  bf16_t* d_ln_bf;
  float *g, *d_ln, *wg_partial, *dw_scratch, *cs, *ln_partial, *cs_partial, *s_tn;
  bf16_t *g_bf, *d_wide, *d_ctx, *d_tout, *lora_u, *lora_v;
  float *wg_partial_side, *cs_partial_side;          // the side stream's own scratch (LoRA gradients, see sf_trainer::side)
  float* dw_scratch2; bf16_t* g1_bf;                 // fused temporal projections: g^T t_out [D, D] fp32 and bf16(g^T ctx) [D, D]
  float *g1_alt, *cs_alt;                            // second G1 / cs buffers: layers alternate, the side stream reads one layer behind
  bf16_t *lora_u_side, *lora_v_side;
  bf16_t *g_bf1, *g_bf2, *d_wide_s, *d_wide_t;       // a layer's weight-gradient operands stay intact until its grouped launch
  float *gh, *d_hn, *d_pc, *dq_total, *pdz, *pdu;
  bf16_t* pds;                         // pooling head: score gradients [M, 32] (the dY operand of dU = ds^T x)
  bf16_t *gh_bf, *d_hm;
  size_t bytes;
};

static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

static TWs tcarve(const sf_trainer* t, void* base, int B, int T) {
  TWs w;
  TCarver c(base);
  const size_t D = t->D, I = t->I, N = t->N;
  const size_t M = (size_t)B * T * N, F = (size_t)B * T;
  w.patches = c.take<bf16_t>(M * t->Kp);
  w.te_rows = c.take<float>((size_t)T * D);
  w.h.resize(t->L + 1);
  w.sl.resize(t->L);
  for (int i = 0; i <= t->L; ++i) w.h[i] = c.take<float>(M * D);
  for (int i = 0; i < t->L; ++i) {
    TSavedLayer& s = w.sl[i];
    s.h1 = c.take<float>(M * D); s.h2 = c.take<float>(M * D);
    s.ln_t = c.take<bf16_t>(M * D); s.tqkv = c.take<bf16_t>(M * 3 * D); s.ctx_t = c.take<bf16_t>(M * D);
    s.t_out = c.take<bf16_t>(M * D);
    s.ln_b = c.take<bf16_t>(M * D); s.sqkv = c.take<bf16_t>(M * 3 * D); s.ctx_s = c.take<bf16_t>(M * D);
    s.ln_a = c.take<bf16_t>(M * D); s.pre = c.take<bf16_t>(M * I); s.act = c.take<bf16_t>(M * I);
    s.lse_s = c.take<float>(F * (size_t)t->heads * N);
  }
  w.xn = c.take<bf16_t>(M * D); w.pc = c.take<bf16_t>(F * D);
  w.pz = c.take<float>(F * (size_t)t->heads * D); w.pprobs = c.take<float>(F * (size_t)t->heads * N);
  w.pml = c.take<float>(sf_pool_ml_floats((int)F, (int)N, t->heads));
  w.pzpart = c.take<float>(sf_pool_z_floats((int)F, (int)N, t->heads, (int)D));
  w.attn_out = c.take<float>(F * D); w.hn = c.take<bf16_t>(F * D);
  w.hm_pre = c.take<bf16_t>(F * I); w.hm = c.take<bf16_t>(F * I);
  // scratch
  w.g = c.take<float>(M * D); w.d_ln = c.take<float>(M * D);
  w.d_ln_bf = reinterpret_cast<bf16_t*>(w.d_ln);          // per-layer LayerNorm input gradients travel as bf16
  w.g_bf = c.take<bf16_t>(M * D);
  w.d_wide = c.take<bf16_t>(M * max_sz(I, 3 * D));
  w.d_ctx = c.take<bf16_t>(M * D); w.d_tout = c.take<bf16_t>(M * D);
  w.g_bf1 = c.take<bf16_t>(M * D); w.g_bf2 = c.take<bf16_t>(M * D);
  w.d_wide_s = c.take<bf16_t>(M * 3 * D); w.d_wide_t = c.take<bf16_t>(M * 3 * D);
  w.lora_u = c.take<bf16_t>(M * kRank); w.lora_v = c.take<bf16_t>(M * kRank);
  w.lora_u_side = c.take<bf16_t>(M * kRank); w.lora_v_side = c.take<bf16_t>(M * kRank);
  size_t wp = 0;
  const int Mi = (int)M, Fi = (int)F, Di = t->D, Ii = t->I;
  wp = max_sz(wp, sf_wgrad_partial_floats(Mi, Di, Ii)); wp = max_sz(wp, sf_wgrad_partial_floats(Mi, Ii, Di));
  wp = max_sz(wp, sf_wgrad_partial_floats(Mi, 3 * Di, Di)); wp = max_sz(wp, sf_wgrad_partial_floats(Mi, Di, Di));
  wp = max_sz(wp, sf_wgrad_partial_floats(Mi, 2 * Di, Di)); wp = max_sz(wp, sf_wgrad_partial_floats(Mi, Di, t->Kp));
  wp = max_sz(wp, sf_wgrad_partial_floats(Fi, Di, Ii)); wp = max_sz(wp, sf_wgrad_partial_floats(Fi, Ii, Di));
  wp = max_sz(wp, sf_wgrad_partial_floats(Fi, Di, Di));
  wp = max_sz(wp, sf_wgrad_partial_floats(Mi, 3 * Di, kRank)); wp = max_sz(wp, sf_wgrad_partial_floats(Mi, kRank, Di));
  {     // one layer's Linears in one grouped launch (backward_layer)
    int tiles = 0, n1 = 0;
    const int dims[7][2] = {{Di, Ii}, {Ii, Di}, {Di, Di}, {3 * Di, Di}, {Di, Di}, {Di, Di}, {3 * Di, Di}};
    for (const auto& d : dims)
      if (sf_wgrad_groupable(Mi, d[0], d[1])) { tiles += (d[0] / 256) * (d[1] / 256); n1 += d[0]; }
    for (int n = 1; n <= tiles; ++n) wp = max_sz(wp, sf_wgrad_group_partial_floats(Mi, n, n1));
  }
  w.wg_partial = c.take<float>(wp);
  {
    size_t ws2 = 0;
    ws2 = max_sz(ws2, sf_wgrad_partial_floats(Mi, 3 * Di, kRank)); ws2 = max_sz(ws2, sf_wgrad_partial_floats(Mi, kRank, Di));
    ws2 = max_sz(ws2, sf_wgrad_partial_floats(Mi, Di, kRank));
    ws2 = max_sz(ws2, sf_wgrad_partial_floats(Di, Di, Di));      // temporal_fused_grads on the side stream: dW_o = (tanh(g) W_d)^T G1, M = D
    w.wg_partial_side = c.take<float>(ws2);
  }
  w.dw_scratch = c.take<float>((size_t)3 * D * D);
  w.dw_scratch2 = c.take<float>((size_t)D * D); w.g1_bf = c.take<bf16_t>((size_t)D * D);
  w.g1_alt = c.take<float>((size_t)D * D); w.cs_alt = c.take<float>(max_sz(I, 3 * D));
  w.cs = c.take<float>(max_sz(I, 3 * D));
  w.ln_partial = c.take<float>(sf_ln_bwd_partial_floats(t->D));
  w.cs_partial = c.take<float>(sf_colsum_partial_floats((int)max_sz(I, 3 * D)));
  w.cs_partial_side = c.take<float>(sf_colsum_partial_floats((int)max_sz(I, 3 * D)));
  w.s_tn = c.take<float>((size_t)T * N * D);
  w.gh = c.take<float>(F * D); w.d_hn = c.take<float>(F * D); w.d_pc = c.take<float>(F * D);
  w.dq_total = c.take<float>(D);
  w.pdz = c.take<float>(F * (size_t)t->heads * D); w.pdu = c.take<float>((size_t)32 * D); w.pds = c.take<bf16_t>(M * 32);
  w.gh_bf = c.take<bf16_t>(F * D); w.d_hm = c.take<bf16_t>(F * I);
  w.bytes = (c.off + 255) & ~(size_t)255;
  return w;
}

static int check_bt(const sf_trainer* t, int B, int T) {
  if (!t) return sf_set_err(SF_ERR_INVALID, "null handle");
  if (B <= 0 || T <= 0) return sf_set_err(SF_ERR_INVALID, "bad geometry B=%d T=%d", B, T);
  if (T > t->cfg.num_frames) return sf_set_err(SF_ERR_INVALID, "training needs T <= config.num_frames (%d > %d)", T, t->cfg.num_frames);
  if (T > 32) return sf_set_err(SF_ERR_INVALID, "temporal attention backward handles T <= 32 (got %d)", T);
  if ((size_t)B * T * t->N * (size_t)(t->I > 3 * t->D ? t->I : 3 * t->D) * 2 >= ((size_t)1 << 32))
    return sf_set_err(SF_ERR_INVALID, "batch too large for 32-bit buffer offsets");
  return SF_OK;
}

extern "C" int sf_trainer_workspace_bytes(const sf_trainer* t, int B, int T, size_t* out) {
  int rc = check_bt(t, B, T);
  if (rc) return rc;
  if (!out) return sf_set_err(SF_ERR_INVALID, "null out");
  *out = tcarve(t, nullptr, B, T).bytes;
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// GEMM helpers
// ------------------------------------------------------------------------------------------------
static SfGemmArgs tgemm_args(const bf16_t* a, const bf16_t* w, const float* bias, int M, int N, int K, int epi, float* out_f32,
                            bf16_t* out_bf, const float* resid) {
  SfGemmArgs g;
  memset(&g, 0, sizeof(g));
  g.a_hi = a; g.w_hi = w; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.epi = epi; g.alpha = 1.f; g.resid = resid;
  g.out_f32 = out_f32; g.out_hi = epi == SF_EPI_RESID_F32 ? nullptr : out_bf; g.ldc = N;
  return g;
}
static hipError_t tgemm(const bf16_t* a, const bf16_t* w, const float* bias, int M, int N, int K, int epi, hipStream_t s,
                        float* out_f32, bf16_t* out_bf, const float* resid = nullptr) {
  return sf_launch_gemm(tgemm_args(a, w, bias, M, N, K, epi, out_f32, out_bf, resid), false, s);
}
// pre = x W^T + b and act = gelu(pre): one launch where the 256^2 kernel takes the shape, else GEMM + GELU pass
static hipError_t lin_fwd_gelu(const sf_trainer* t, const TLin& l, const bf16_t* x, int M, hipStream_t s, bf16_t* pre, bf16_t* act);
// d_pre = (dy W) * gelu'(pre): same
static hipError_t lin_dgrad_dgelu(const TLin& l, const bf16_t* dy, int M, hipStream_t s, bf16_t* d_pre, const bf16_t* pre);
// y = x W^T + b
static hipError_t lin_fwd(const sf_trainer* t, const TLin& l, const bf16_t* x, int M, int epi, hipStream_t s, float* out_f32,
                          bf16_t* out_bf, const float* resid = nullptr) {
  return tgemm(x, l.w, lin_bias(t, l), M, l.N, l.K, epi, s, out_f32, out_bf, resid);
}
// dx = dy W   (dy [M,N] -> dx [M,K]); the forward-scaled weight is used as is
static hipError_t lin_dgrad(const TLin& l, const bf16_t* dy, int M, hipStream_t s, float* out_f32, bf16_t* out_bf) {
  return tgemm(dy, l.wT, nullptr, M, l.K, l.N, out_f32 ? SF_EPI_F32 : SF_EPI_BF16, s, out_f32, out_bf);
}

static hipError_t lin_fwd_gelu(const sf_trainer* t, const TLin& l, const bf16_t* x, int M, hipStream_t s, bf16_t* pre, bf16_t* act) {
  SfGemmArgs g = tgemm_args(x, l.w, lin_bias(t, l), M, l.N, l.K, SF_EPI_BF16, nullptr, pre, nullptr);
  g.aux_mode = 1; g.aux = act;
  if (sf_gemm256_aux_supported(g)) return sf_launch_gemm(g, false, s);
  g.aux_mode = 0; g.aux = nullptr;
  hipError_t e = sf_launch_gemm(g, false, s);
  return e != hipSuccess ? e : sf_launch_gelu_fwd(pre, act, (size_t)M * l.N, s);
}
static hipError_t lin_dgrad_dgelu(const TLin& l, const bf16_t* dy, int M, hipStream_t s, bf16_t* d_pre, const bf16_t* pre) {
  SfGemmArgs g = tgemm_args(dy, l.wT, nullptr, M, l.K, l.N, SF_EPI_BF16, nullptr, d_pre, nullptr);
  g.aux_mode = 2; g.aux = const_cast<bf16_t*>(pre);
  if (sf_gemm256_aux_supported(g)) return sf_launch_gemm(g, false, s);
  g.aux_mode = 0; g.aux = nullptr;
  hipError_t e = sf_launch_gemm(g, false, s);
  return e != hipSuccess ? e : sf_launch_gelu_bwd(d_pre, pre, (size_t)M * l.K, s);
}

// ------------------------------------------------------------------------------------------------
// forward (activations kept)
// ------------------------------------------------------------------------------------------------
extern "C" int sf_trainer_forward(sf_trainer* t, const void* pixels, int pixel_dtype, int B, int T, float* last_hidden,
                                  float* pooler, void* workspace, size_t workspace_bytes, sf_stream stream) {
  int rc = check_bt(t, B, T);
  if (rc) return rc;
  if (!t->params_dev) return sf_set_err(SF_ERR_STATE, "sf_trainer_sync_weights has not been called");
  if (!pixels || !workspace || !pooler) return sf_set_err(SF_ERR_INVALID, "null argument");
  if (pixel_dtype != SF_F32 && pixel_dtype != SF_BF16 && pixel_dtype != SF_U8)
    return sf_set_err(SF_ERR_INVALID, "pixels must be fp32, bf16 or uint8 (uint8: (x/255 - 0.5)/0.5 fused)");
  HIP_TRY(hipSetDevice(t->device));
  hipStream_t s = (hipStream_t)stream;
  const TWs ws = tcarve(t, workspace, B, T);
  if (workspace_bytes < ws.bytes) return sf_set_err(SF_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, ws.bytes);
  const sf_config& c = t->cfg;
  const int D = t->D, I = t->I, N = t->N, heads = t->heads;
  const int M = B * T * N, F = B * T;
  const float eps = c.layer_norm_eps;
  const float* P0 = t->params_dev;
  t->fB = 0;

  SfRowIndex idx;
  idx.n = T;
  for (int i = 0; i < T; ++i) idx.idx[i] = i;             // modeling:436-439 (T <= num_frames)
  HIP_TRY(sf_launch_gather_rows(PP(t, P0, t->p_time), ws.te_rows, idx, D, s));
  HIP_TRY(sf_launch_patchify(pixels, pixel_dtype == SF_U8 ? 2 : (pixel_dtype == SF_BF16 ? 1 : 0), ws.patches, nullptr, F, c.num_channels, c.image_size, c.image_size,
                             c.patch_size, s));
  {
    SfGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a_hi = ws.patches; g.w_hi = t->patch.w; g.bias = PP(t, P0, t->patch.pb);
    g.M = M; g.N = D; g.K = t->Kp; g.epi = SF_EPI_EMBED_F32;
    g.pos = PP(t, P0, t->p_pos); g.time_rows = ws.te_rows; g.Np = N; g.Tn = T;
    g.out_f32 = ws.h[0]; g.ldc = D;
    if (t->drop_hidden > 0.f) {
      // pos_drop(patches + pos) then time_drop(. + time) (modeling:374, 378): the GEMM adds a zero time table, one elementwise pass does the rest
      HIP_TRY(hipMemsetAsync(ws.g, 0, (size_t)T * D * sizeof(float), s));
      g.time_rows = ws.g;
      HIP_TRY(sf_launch_gemm(g, false, s));
      HIP_TRY(sf_launch_embed_dropout(ws.h[0], ws.te_rows, M, D, T, N, sf_drop_make(t->drop_hidden, t->drop_seed, (unsigned)t->L * 8u),
                                      sf_drop_make(t->drop_hidden, t->drop_seed, (unsigned)t->L * 8u + 1u), s));
    } else {
      HIP_TRY(sf_launch_gemm(g, false, s));
    }
  }
  const float scale = 0.125f;
  const bool hd = t->drop_hidden > 0.f;
  auto site = [&](int li, int k) { return sf_drop_make(t->drop_hidden, t->drop_seed, (unsigned)(li * 8 + k)); };
  const bool ad = t->drop_attn > 0.f;
  if (ad && (T > 16 || N > 224)) return sf_set_err(SF_ERR_INVALID, "attention dropout needs clips of <= 16 frames and <= 224 patches per frame");
  auto asite = [&](int li, int k) { return sf_drop_make(t->drop_attn, t->drop_seed, (unsigned)(li * 8 + k)); };
  if (t->dp_scales && (t->dp_B != B || t->dp_T != T))
    return sf_set_err(SF_ERR_INVALID, "drop_path factors were set for B=%d T=%d, the forward runs B=%d T=%d", t->dp_B, t->dp_T, B, T);
  const float* dp = t->dp_scales;
  const size_t dp_per_layer = (size_t)B * N + (size_t)B * T + (size_t)B;
  // the temporal branch's two projections as one: only without drop_path / hidden dropout (both sit between them);
  // SF_TRAIN_UNFUSED_TEMPORAL keeps the two launches (A/B, and the path the drop rates use)
  const bool tfuse = !dp && !hd && sf_sw(SW_TRAIN_UNFUSED_TEMPORAL) == nullptr;
  for (int li = 0; li < t->L; ++li) {
    const TLayer& l = t->layers[li];
    const TSavedLayer& sv = ws.sl[li];
    const float* h = ws.h[li];
    // temporal attention (modeling:937-958)
    HIP_TRY(sf_launch_layernorm(h, PP(t, P0, l.ln_t_g), PP(t, P0, l.ln_t_b), nullptr, sv.ln_t, nullptr, M, D, eps, s));
    HIP_TRY(lin_fwd(t, l.t_qkv, sv.ln_t, M, SF_EPI_BF16, s, nullptr, sv.tqkv));
    {
      SfAttnArgs a;
      memset(&a, 0, sizeof(a));
      a.q = sv.tqkv; a.k = sv.tqkv + D; a.v = sv.tqkv + 2 * D;
      a.row_pitch_q = 3 * D; a.row_pitch_kv = 3 * D; a.heads = heads; a.scale = scale;
      a.N = N; a.B = B; a.Tq = T; a.Tk = T; a.Tcap = T; a.t_past = 0; a.causal = c.enable_causal_temporal;
      a.Tq_cap = T; a.q_t0 = 0; a.ctx_hi = sv.ctx_t; a.D = D;
      if (ad) a.drop = asite(li, 4);
      HIP_TRY(sf_launch_temporal_attention(a, false, s));
    }
    if (tfuse) {
      // h1 = h + ctx W_f^T + b_f: output.dense and temporal_dense (modeling:947-958) have nothing between them at drop rates 0
      HIP_TRY(tgemm(sv.ctx_t, l.wf, l.bf, M, D, D, SF_EPI_RESID_F32, s, sv.h1, nullptr, h));
    } else {
    HIP_TRY(lin_fwd(t, l.t_out, sv.ctx_t, M, SF_EPI_BF16, s, nullptr, sv.t_out));
    // drop_path (modeling:949) sits between the attention output and temporal_dense: the saved t_out IS the dropped tensor
    // hidden dropout of the temporal SelfOutput (modeling:761) rides on the same pass
    if (dp || hd) HIP_TRY(sf_launch_rowscale_bf16(sv.t_out, sv.t_out, dp ? dp + (size_t)li * dp_per_layer : nullptr, M, D, 0, T, N, s, site(li, 0)));
    HIP_TRY(lin_fwd(t, l.t_dense, sv.t_out, M, SF_EPI_RESID_F32, s, sv.h1, nullptr, h));      // h1 = h + tanh(g) * dense(.)
    }
    // spatial attention (modeling:962-996)
    HIP_TRY(sf_launch_layernorm(sv.h1, PP(t, P0, l.ln_b_g), PP(t, P0, l.ln_b_b), nullptr, sv.ln_b, nullptr, M, D, eps, s));
    HIP_TRY(lin_fwd(t, l.s_qkv, sv.ln_b, M, SF_EPI_BF16, s, nullptr, sv.sqkv));
    {
      SfAttnArgs a;
      memset(&a, 0, sizeof(a));
      a.q = sv.sqkv; a.k = sv.sqkv + D; a.v = sv.sqkv + 2 * D;
      a.row_pitch_q = 3 * D; a.row_pitch_kv = 3 * D; a.heads = heads; a.scale = scale;
      a.N = N; a.frames = F; a.ctx_hi = sv.ctx_s; a.D = D; a.lse2_out = sv.lse_s;
      if (ad) a.drop = asite(li, 5);
      HIP_TRY(sf_launch_spatial_attention(a, false, s));
    }
    if (dp || hd) {     // h2 = h1 + drop_path(dropout(out(ctx))) (modeling:752 / 761, 980): the branch leaves the GEMM as fp32, the residual add applies the factors
      HIP_TRY(lin_fwd(t, l.s_out, sv.ctx_s, M, SF_EPI_F32, s, ws.g, nullptr));
      HIP_TRY(sf_launch_resid_rowscale(sv.h2, sv.h1, ws.g, dp ? dp + (size_t)li * dp_per_layer + (size_t)B * N : nullptr, M, D, 1, T, N, s, site(li, 1)));
    } else {
      HIP_TRY(lin_fwd(t, l.s_out, sv.ctx_s, M, SF_EPI_RESID_F32, s, sv.h2, nullptr, sv.h1));
    }
    // MLP (modeling:997-1000)
    HIP_TRY(sf_launch_layernorm(sv.h2, PP(t, P0, l.ln_a_g), PP(t, P0, l.ln_a_b), nullptr, sv.ln_a, nullptr, M, D, eps, s));
    HIP_TRY(lin_fwd_gelu(t, l.up, sv.ln_a, M, s, sv.pre, sv.act));
    if (hd) HIP_TRY(sf_launch_rowscale_bf16(sv.act, sv.act, nullptr, M, I, 0, T, N, s, site(li, 2)));      // dropout behind the activation (modeling:822): the saved act IS the dropped tensor
    if (dp || hd) {     // out = h2 + drop_path(dropout(mlp)) (modeling:835, 1000)
      HIP_TRY(lin_fwd(t, l.down, sv.act, M, SF_EPI_F32, s, ws.g, nullptr));
      HIP_TRY(sf_launch_resid_rowscale(ws.h[li + 1], sv.h2, ws.g, dp ? dp + (size_t)li * dp_per_layer + (size_t)B * N + (size_t)B * T : nullptr, M, D, 2, T, N, s,
                                       site(li, 3)));
    } else {
      HIP_TRY(lin_fwd(t, l.down, sv.act, M, SF_EPI_RESID_F32, s, ws.h[li + 1], nullptr, sv.h2));
    }
  }
  // post LayerNorm + pooling head (modeling:1330-1340, 1141-1154)
  // The probe attention reads the fp32 tokens and never projects them to k / v (sf_pool_head.hip): scores = x . U, z_h = sum_n p_hn x_n,
  // ctx_h = Wv_h z_h + bv_h.  The caller's last_hidden_state (or, without one, the backward's scratch) holds the fp32 rows; the
  // backward itself works from the bf16 copy ws.xn, so the caller may do with its tensor what it likes.
  float* xf = last_hidden ? last_hidden : ws.g;
  HIP_TRY(sf_launch_layernorm(ws.h[t->L], PP(t, P0, t->post_g), PP(t, P0, t->post_b), xf, ws.xn, nullptr, M, D, eps, s));
  {
    SfPoolArgs pa;
    memset(&pa, 0, sizeof(pa));
    // token splits as in inference (a frame's tokens over S workgroups): the raw scores and {max, sum} per split are kept, the backward
    // finishes the softmax itself; the combined, normalised sums z land in ws.pz (directly when S == 1)
    const int S = sf_pool_splits(F, N, heads);
    pa.x = xf; pa.u_hi = t->head_u_hi; pa.u_lo = t->head_u_lo; pa.zpart = S == 1 ? ws.pz : ws.pzpart; pa.probs = ws.pprobs; pa.probs_raw = 1; pa.ml = ws.pml;
    pa.F = F; pa.N = N; pa.heads = heads; pa.D = D; pa.S = S; pa.normalize = S == 1;
    HIP_TRY(sf_launch_pool_probe(pa, s));
    SfPoolCtxArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.zpart = pa.zpart; ca.ml = ws.pml; ca.z_out = S == 1 ? nullptr : ws.pz;
    ca.wv = PP(t, P0, t->p_inw, (size_t)2 * D * D); ca.ldw = D; ca.bv = PP(t, P0, t->p_inb, (size_t)2 * D);
    ca.ctx_hi = ws.pc; ca.F = F; ca.heads = heads; ca.D = D; ca.S = S;
    HIP_TRY(sf_launch_pool_ctx(ca, s));
  }
  HIP_TRY(lin_fwd(t, t->head_out, ws.pc, F, SF_EPI_F32, s, ws.attn_out, nullptr));
  HIP_TRY(sf_launch_layernorm(ws.attn_out, PP(t, P0, t->hln_g), PP(t, P0, t->hln_b), nullptr, ws.hn, nullptr, F, D, eps, s));
  HIP_TRY(lin_fwd(t, t->fc1, ws.hn, F, SF_EPI_BF16, s, nullptr, ws.hm_pre));
  HIP_TRY(sf_launch_gelu_fwd(ws.hm_pre, ws.hm, (size_t)F * I, s));
  HIP_TRY(lin_fwd(t, t->fc2, ws.hm, F, SF_EPI_RESID_F32, s, pooler, nullptr, ws.attn_out));
  t->fB = B; t->fT = T; t->f_dp = dp; t->f_tfuse = tfuse;
  t->f_drop_hidden = t->drop_hidden; t->f_drop_attn = t->drop_attn; t->f_drop_seed = t->drop_seed;
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct BwdCtx {
  const sf_trainer* t;
  const TWs* ws;
  float* grads;
  hipStream_t s;
  bool on_side = false;               // this context launches on the side stream and uses the side scratch
  bf16_t* lora_u() const { return on_side ? ws->lora_u_side : ws->lora_u; }
  bf16_t* lora_v() const { return on_side ? ws->lora_v_side : ws->lora_v; }
  float* wg_partial() const { return on_side ? ws->wg_partial_side : ws->wg_partial; }
  float* cs_partial() const { return on_side ? ws->cs_partial_side : ws->cs_partial; }
};

// weight + bias gradients of one Linear: dW (+)= dy^T x, db += colsum(dy); LoRA factors from dW_eff
static hipError_t lin_wgrad(const BwdCtx& c, const TLin& l, const bf16_t* dy, const bf16_t* x, int M) {
  const sf_trainer* t = c.t;
  float* gw = GG(t, c.grads, l.pw, l.pw_off);
  float* gb = GG(t, c.grads, l.pb, l.pb_off);
  SfWgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = dy; a.ldy = l.N; a.x = x; a.ldx = l.K; a.M = M; a.N1 = l.N; a.N2 = l.K; a.ldo = l.K; a.alpha = 1.f;
  a.partial = c.wg_partial();
  hipError_t e = hipSuccess;
  if (l.pla >= 0) {
    // W_eff = W + B A (modeling:541-545):  dB = dy^T (x A^T),  dA = (dy B)^T x  — two rank-32 projections
    // and two skinny weight-gradient GEMMs instead of the full [N,K] one (the base weight is frozen)
    if ((e = tgemm(x, l.la_bf, nullptr, M, kRank, l.K, SF_EPI_BF16, c.s, nullptr, c.lora_u())) != hipSuccess) return e;
    if ((e = tgemm(dy, l.lbT_bf, nullptr, M, kRank, l.N, SF_EPI_BF16, c.s, nullptr, c.lora_v())) != hipSuccess) return e;
    SfWgradArgs b = a;
    b.dy = dy; b.ldy = l.N; b.x = c.lora_u(); b.ldx = kRank; b.N1 = l.N; b.N2 = kRank; b.ldo = kRank;
    b.out = GG(t, c.grads, l.plb); b.accumulate = 1;
    if (b.out && (e = sf_launch_wgrad(b, c.s)) != hipSuccess) return e;
    b.dy = c.lora_v(); b.ldy = kRank; b.x = x; b.ldx = l.K; b.N1 = kRank; b.N2 = l.K; b.ldo = l.K;
    b.out = GG(t, c.grads, l.pla);
    if (b.out && (e = sf_launch_wgrad(b, c.s)) != hipSuccess) return e;
  }
  if (gw) {
    a.out = gw; a.accumulate = 1;
    a.dbias = gb; a.dbias_scratch = c.cs_partial();      // bias gradient rides on the same launch
    e = sf_launch_wgrad(a, c.s);
  } else if (gb) {
    e = sf_launch_colsum_bf16(dy, M, l.N, l.N, 1.f, gb, 1, c.cs_partial(), c.s);
  }
  return e;
}

static int backward_head(const BwdCtx& c, const float* d_pooler, const float* d_lhs, int B, int T) {
  const sf_trainer* t = c.t;
  const TWs& ws = *c.ws;
  hipStream_t s = c.s;
  const int D = t->D, I = t->I, N = t->N;
  const int M = B * T * N, F = B * T;
  const float eps = t->cfg.layer_norm_eps;
  const float* P0 = t->params_dev;
  // pooler = attn_out + fc2(gelu(fc1(LN(attn_out))))
  HIP_TRY(sf_launch_split(d_pooler, ws.gh_bf, nullptr, (size_t)F * D, s));
  HIP_TRY(lin_dgrad(t->fc2, ws.gh_bf, F, s, nullptr, ws.d_hm));
  HIP_TRY(lin_wgrad(c, t->fc2, ws.gh_bf, ws.hm, F));
  HIP_TRY(sf_launch_gelu_bwd(ws.d_hm, ws.hm_pre, (size_t)F * I, s));
  HIP_TRY(lin_dgrad(t->fc1, ws.d_hm, F, s, ws.d_hn, nullptr));
  HIP_TRY(lin_wgrad(c, t->fc1, ws.d_hm, ws.hn, F));
  HIP_TRY(sf_launch_ln_bwd(ws.attn_out, ws.d_hn, 0, PP(t, P0, t->hln_g), d_pooler, ws.gh, ws.gh_bf, GG(t, c.grads, t->hln_g), GG(t, c.grads, t->hln_b),
                           ws.ln_partial, F, D, eps, s));
  // attn_out = out_proj(ctx)   (gh_bf = bf16(gh) written by the LayerNorm backward)
  HIP_TRY(lin_dgrad(t->head_out, ws.gh_bf, F, s, ws.d_pc, nullptr));
  HIP_TRY(lin_wgrad(c, t->head_out, ws.gh_bf, ws.pc, F));
  // probe attention over the N tokens of every frame
  // ctx_h = Wv_h z_h + bv_h: dz, dWv, dbv (the value rows are in_proj rows [2D, 3D))
  HIP_TRY(sf_launch_pool_ctx_bwd(ws.d_pc, t->head_kv.wT, 2 * D, D, ws.pz, ws.pdz, GG(t, c.grads, t->p_inw, (size_t)2 * D * D), D,
                                 GG(t, c.grads, t->p_inb, (size_t)2 * D), F, t->heads, D, s));
  // p = softmax(x . U), z = p x: dx (+ the gradient that arrives through last_hidden_state) and the score gradients ds
  {
    SfPoolBwdArgs pb;
    memset(&pb, 0, sizeof(pb));
    pb.x_bf = ws.xn; pb.probs = ws.pprobs; pb.probs_raw = 1; pb.ml = ws.pml; pb.ml_splits = sf_pool_splits(F, N, t->heads); pb.z = ws.pz; pb.dz = ws.pdz; pb.u = t->head_u; pb.d_lhs = d_lhs; pb.dx = ws.d_ln; pb.ds_bf = ws.pds;
    pb.F = F; pb.N = N; pb.heads = t->heads; pb.D = D;
    HIP_TRY(sf_launch_pool_probe_bwd(pb, s));
  }
  {   // dU = ds^T x over all token rows ([32, D], rows >= heads zero), then U_h = Wk_h^T q_h: dWk_h += q_h dU_h^T, dq_h = Wk_h dU_h.
      // The key bias gets no gradient: its term q_h . bk_h is constant over the keys and cancels in the softmax.
    SfWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = ws.pds; a.ldy = 32; a.x = ws.xn; a.ldx = D; a.M = M; a.N1 = 32; a.N2 = D; a.out = ws.pdu; a.ldo = D; a.alpha = 1.f;
    a.partial = ws.wg_partial;
    HIP_TRY(sf_launch_wgrad(a, s));
    HIP_TRY(sf_launch_pool_u_bwd(ws.pdu, PP(t, P0, t->p_inw, (size_t)D * D), t->head_q, GG(t, c.grads, t->p_inw, (size_t)D * D), ws.dq_total, D, s));
  }
  HIP_TRY(sf_launch_head_query_bwd(ws.dq_total, PP(t, P0, t->p_probe), PP(t, P0, t->p_inw), 0.125f, GG(t, c.grads, t->p_inw),
                                   GG(t, c.grads, t->p_inb), GG(t, c.grads, t->p_probe), D, s));
  // post_layernorm: g = dLN(h_L)
  HIP_TRY(sf_launch_ln_bwd(ws.h[t->L], ws.d_ln, 0, PP(t, P0, t->post_g), nullptr, ws.g, ws.g_bf, GG(t, c.grads, t->post_g), GG(t, c.grads, t->post_b),
                           ws.ln_partial, M, D, eps, s));
  return SF_OK;
}

// one encoder layer's weight gradients, batched: a Linear whose [N, K] is made of 256^2 tiles (and is trained densely, no
// LoRA factors) is queued here with its operands and launched together with the others at the end of the layer
struct LayerWgrads {
  SfWgradGroup g;
  bool on;
};
static hipError_t lin_wgrad_queued(const BwdCtx& c, LayerWgrads& q, const TLin& l, const bf16_t* dy, const bf16_t* x, int M) {
  float* gw = GG(c.t, c.grads, l.pw, l.pw_off);
  if (!q.on || l.pla >= 0 || !gw || !sf_wgrad_groupable(M, l.N, l.K) || q.g.njobs >= SF_WG_MAX_JOBS) return lin_wgrad(c, l, dy, x, M);
  SfWgradJob& J = q.g.job[q.g.njobs++];
  memset(&J, 0, sizeof(J));
  J.dy = dy; J.x = x; J.ldy = l.N; J.ldx = l.K; J.N1 = l.N; J.N2 = l.K; J.ldo = l.K; J.alpha = 1.f; J.accumulate = 1;
  J.out = gw; J.dbias = GG(c.t, c.grads, l.pb, l.pb_off);
  return hipSuccess;
}

// Gradients of the fused temporal projections from G1 = g^T ctx (ws.dw_scratch, fp32 [D, D]) and cs = colsum g (ws.cs):
//   G = g^T t_out = G1 W_o^T + cs b_o^T  ->  dW_d += tanh(g) G, db_d += tanh(g) cs, dgate += (1 - tanh^2)(<G, W_d> + <cs, b_d>)
//   dW_o += (tanh(g) W_d)^T G1,  db_o += (tanh(g) W_d)^T cs                    (all D x D; bf16 operands like every backward GEMM)
static hipError_t temporal_fused_grads(const BwdCtx& c, const TLayer& l, int D, const float* g1, const float* cs) {
  const sf_trainer* t = c.t;
  const TWs& ws = *c.ws;
  const float* P0 = t->params_dev;
  hipStream_t s = c.s;
  hipError_t e;
  if ((e = sf_launch_split(g1, ws.g1_bf, nullptr, (size_t)D * D, s)) != hipSuccess) return e;
  if ((e = tgemm(ws.g1_bf, l.t_out.w, nullptr, D, D, D, SF_EPI_F32, s, ws.dw_scratch2, nullptr)) != hipSuccess) return e;
  if ((e = sf_launch_gate_grad(ws.dw_scratch2, cs, PP(t, P0, l.t_dense.pw), PP(t, P0, l.t_dense.pb), PP(t, P0, l.gate),
                               GG(t, c.grads, l.t_dense.pw), GG(t, c.grads, l.t_dense.pb), GG(t, c.grads, l.gate), t->red_partial, D, D, s,
                               PP(t, P0, l.t_out.pb))) != hipSuccess) return e;
  if (float* gwo = GG(t, c.grads, l.t_out.pw)) {
    SfWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = l.t_dense.w; a.ldy = D; a.x = ws.g1_bf; a.ldx = D; a.M = D; a.N1 = D; a.N2 = D; a.out = gwo; a.ldo = D; a.accumulate = 1; a.alpha = 1.f;
    a.partial = c.wg_partial();
    if ((e = sf_launch_wgrad(a, s)) != hipSuccess) return e;
  }
  if (float* gbo = GG(t, c.grads, l.t_out.pb))
    if ((e = sf_launch_matvec_t_bf16(l.t_dense.w, D, cs, gbo, D, D, s)) != hipSuccess) return e;
  return hipSuccess;
}

// the side stream of the LoRA gradients: created on first use; SF_TRAIN_SIDE_STREAM=0 keeps everything on the caller's stream (A/B)
static bool side_stream_ready(sf_trainer* t) {
  if (t->side_state == 0) {
    const char* e = sf_sw(SW_TRAIN_SIDE_STREAM);
    t->side_state = -1;
    if (!(e && e[0] == '0') && hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&t->ev_small[0], hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&t->ev_small[1], hipEventDisableTiming) == hipSuccess)
      t->side_state = 1;
  }
  return t->side_state == 1;
}
// weight gradients of a LoRA-adapted Linear, forked onto the side stream behind everything the caller's stream has enqueued so
// far (its operands are complete there); the caller joins at the end of the layer (side_join)
static hipError_t lin_wgrad_side(const BwdCtx& c, const TLin& l, const bf16_t* dy, const bf16_t* x, int M, bool* forked) {
  sf_trainer* t = const_cast<sf_trainer*>(c.t);
  // only the rank-32 factor gradients go to the side stream: ws.wg_partial_side is sized for those shapes.  A LoRA-adapted Linear whose
  // base weight is NOT frozen (add_lora_spatial without frozen_spatial: scripts/pretrain_streamformer.sh:32-33) also needs the full
  // [N, K] gradient and stays on the caller's stream with the full-size scratch.
  if (l.pla < 0 || GG(c.t, c.grads, l.pw, l.pw_off) != nullptr || !side_stream_ready(t)) return lin_wgrad(c, l, dy, x, M);
  hipError_t e;
  if ((e = hipEventRecord(t->ev_fork, c.s)) != hipSuccess) return e;
  if ((e = hipStreamWaitEvent(t->side, t->ev_fork, 0)) != hipSuccess) return e;
  BwdCtx cs = c;
  cs.s = t->side; cs.on_side = true;
  *forked = true;
  return lin_wgrad(cs, l, dy, x, M);
}
static hipError_t side_join(const BwdCtx& c) {
  sf_trainer* t = const_cast<sf_trainer*>(c.t);
  hipError_t e = hipEventRecord(t->ev_join, t->side);
  return e != hipSuccess ? e : hipStreamWaitEvent(c.s, t->ev_join, 0);
}

static int backward_layer(const BwdCtx& c, int li, int B, int T) {
  const sf_trainer* t = c.t;
  const TWs& ws = *c.ws;
  hipStream_t s = c.s;
  const TLayer& l = t->layers[li];
  const TSavedLayer& sv = ws.sl[li];
  const int D = t->D, N = t->N;
  const int M = B * T * N, F = B * T;
  const float eps = t->cfg.layer_norm_eps;
  const float* P0 = t->params_dev;

  // drop_path: the gradient entering a dropped branch carries the branch's factor (0 or 1 / keep per sample group)
  const float* dp = t->f_dp ? t->f_dp + (size_t)li * ((size_t)B * N + (size_t)B * T + (size_t)B) : nullptr;
  const bool ungrouped = sf_sw(SW_WGRAD_UNGROUPED) != nullptr;
  LayerWgrads q;
  memset(&q.g, 0, sizeof(q.g));
  q.g.M = M; q.g.partial = ws.wg_partial;
  const bool hd = t->f_drop_hidden > 0.f;
  const int I = t->I;
  auto site = [&](int k) { return sf_drop_make(t->f_drop_hidden, t->f_drop_seed, (unsigned)(li * 8 + k)); };
  q.on = !dp && !hd && !ungrouped;  // the drop_path / dropout copies reuse d_ctx / d_tout inside the layer: immediate launches there
  const bool side_ok = q.on;        // same condition: the side stream's operands must stay untouched until the end of the layer
  bool forked = false;
  // g (fp32) and its bf16 copy are both written by the LayerNorm backward that produced them; the bf16 copy rotates through
  // g_bf -> g_bf1 -> g_bf2 -> g_bf inside the layer and the three attention / MLP gradients have their own wide buffers, so
  // that every queued weight gradient still finds its operands at the end of the layer
  // ---- MLP: out = h2 + down(gelu(up(LN_a(h2)))) --------------------------------------------------------
  const bf16_t* gy = ws.g_bf;
  if (dp || hd) { HIP_TRY(sf_launch_rowscale_bf16(ws.g_bf, ws.d_ctx, dp ? dp + (size_t)B * N + (size_t)B * T : nullptr, M, D, 2, T, N, s, site(3))); gy = ws.d_ctx; }
  HIP_TRY(lin_dgrad_dgelu(l.down, gy, M, s, ws.d_wide, sv.pre));            // d pre = (g W_down) * gelu'(pre)  [M,I]
  if (hd) HIP_TRY(sf_launch_rowscale_bf16(ws.d_wide, ws.d_wide, nullptr, M, I, 0, T, N, s, site(2)));      // ... through the activation's dropout mask (elementwise factors commute)
  HIP_TRY(lin_wgrad_queued(c, q, l.down, gy, sv.act, M));
  HIP_TRY(lin_dgrad(l.up, ws.d_wide, M, s, nullptr, ws.d_ln_bf));
  HIP_TRY(lin_wgrad_queued(c, q, l.up, ws.d_wide, sv.ln_a, M));
  HIP_TRY(sf_launch_ln_bwd(sv.h2, ws.d_ln_bf, 1, PP(t, P0, l.ln_a_g), ws.g, ws.g, ws.g_bf1, GG(t, c.grads, l.ln_a_g), GG(t, c.grads, l.ln_a_b),
                           ws.ln_partial, M, D, eps, s));
  // ---- spatial: h2 = h1 + out(attn(qkv(LN_b(h1)))) ---------------------------------------------------------
  gy = ws.g_bf1;
  if (dp || hd) { HIP_TRY(sf_launch_rowscale_bf16(ws.g_bf1, ws.d_tout, dp ? dp + (size_t)B * N : nullptr, M, D, 1, T, N, s, site(1))); gy = ws.d_tout; }
  HIP_TRY(lin_dgrad(l.s_out, gy, M, s, nullptr, ws.d_ctx));
  if (side_ok && l.s_out.pla >= 0) HIP_TRY(lin_wgrad_side(c, l.s_out, gy, sv.ctx_s, M, &forked));
  else HIP_TRY(lin_wgrad_queued(c, q, l.s_out, gy, sv.ctx_s, M));
  {
    SfAttnBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.qkv = sv.sqkv; a.ld_qkv = 3 * D; a.o = sv.ctx_s; a.ld_o = D; a.d_o = ws.d_ctx; a.d_qkv = ws.d_wide_s;
    a.heads = t->heads; a.D = D; a.scale = 0.125f; a.L = N; a.nseq = F; a.seq_rows = 1; a.causal = 0;
    a.lse2 = sv.lse_s;
    if (t->f_drop_attn > 0.f) a.drop = sf_drop_make(t->f_drop_attn, t->f_drop_seed, (unsigned)(li * 8 + 5));
    HIP_TRY(sf_launch_spatial_attention_bwd(a, s));
  }
  if (side_ok && l.s_qkv.pla >= 0) HIP_TRY(lin_wgrad_side(c, l.s_qkv, ws.d_wide_s, sv.ln_b, M, &forked));
  else HIP_TRY(lin_wgrad_queued(c, q, l.s_qkv, ws.d_wide_s, sv.ln_b, M));
  HIP_TRY(lin_dgrad(l.s_qkv, ws.d_wide_s, M, s, nullptr, ws.d_ln_bf));
  HIP_TRY(sf_launch_ln_bwd(sv.h1, ws.d_ln_bf, 1, PP(t, P0, l.ln_b_g), ws.g, ws.g, ws.g_bf2, GG(t, c.grads, l.ln_b_g), GG(t, c.grads, l.ln_b_b),
                           ws.ln_partial, M, D, eps, s));
  // ---- temporal: h1 = h + tanh(gate) * dense(out(attn(qkv(LN_t(h))))) ----------------------------------------
  const bool tfuse = t->f_tfuse;
  bool dense_queued = false;
  const int par = li & 1;
  float* const g1buf = (tfuse && par) ? ws.g1_alt : ws.dw_scratch;      // fused path: G1 / cs alternate by layer parity
  float* const csbuf = (tfuse && par) ? ws.cs_alt : ws.cs;
  HIP_TRY(hipMemsetAsync(csbuf, 0, (size_t)D * sizeof(float), s));
  if (tfuse) {
    // forward ran h1 = h + ctx W_f^T + b_f with W_f = tanh(g) W_d W_o.  One input-gradient GEMM, d_ctx = g W_f, and ONE token-
    // contracting GEMM, G1 = g^T ctx [D, D] (+ cs = colsum g): everything else is D x D algebra after the grouped launch —
    //   g^T t_out = G1 W_o^T + cs b_o^T (-> dW_d, db_d, dgate as before),  dW_o = (tanh(g) W_d)^T G1,  db_o = (tanh(g) W_d)^T cs
    HIP_TRY(tgemm(ws.g_bf2, l.wfT, nullptr, M, D, D, SF_EPI_BF16, s, nullptr, ws.d_ctx));
    SfWgradJob* J = nullptr;
    SfWgradGroup lone;
    if (q.on && sf_wgrad_groupable(M, D, D) && q.g.njobs < SF_WG_MAX_JOBS) { J = &q.g.job[q.g.njobs++]; dense_queued = true; }
    else { memset(&lone, 0, sizeof(lone)); lone.njobs = 1; lone.M = M; lone.partial = ws.wg_partial; J = &lone.job[0]; }
    memset(J, 0, sizeof(*J));
    J->dy = ws.g_bf2; J->x = sv.ctx_t; J->ldy = D; J->ldx = D; J->N1 = D; J->N2 = D; J->ldo = D; J->alpha = 1.f; J->accumulate = 0;
    J->out = g1buf; J->dbias = csbuf;
    if (!dense_queued) {
      if (sf_wgrad_groupable(M, D, D)) HIP_TRY(sf_launch_wgrad_group(lone, s));
      else {
        SfWgradArgs a;
        memset(&a, 0, sizeof(a));
        a.dy = ws.g_bf2; a.ldy = D; a.x = sv.ctx_t; a.ldx = D; a.M = M; a.N1 = D; a.N2 = D; a.ldo = D; a.alpha = 1.f;
        a.partial = ws.wg_partial; a.out = g1buf; a.accumulate = 0; a.dbias = csbuf; a.dbias_scratch = ws.cs_partial;
        HIP_TRY(sf_launch_wgrad(a, s));
      }
      HIP_TRY(temporal_fused_grads(c, l, D, g1buf, csbuf));
    }
  } else {
  HIP_TRY(lin_dgrad(l.t_dense, ws.g_bf2, M, s, nullptr, ws.d_tout));               // wT already carries tanh(gate)
  // unscaled G = g^T t_out and column sums -> dW, db, dgate (see sf_launch_gate_grad, after the grouped launch)
  dense_queued = q.on && sf_wgrad_groupable(M, D, D) && q.g.njobs < SF_WG_MAX_JOBS;
  if (dense_queued) {
    SfWgradJob& J = q.g.job[q.g.njobs++];
    memset(&J, 0, sizeof(J));
    J.dy = ws.g_bf2; J.x = sv.t_out; J.ldy = D; J.ldx = D; J.N1 = D; J.N2 = D; J.ldo = D; J.alpha = 1.f; J.accumulate = 0;
    J.out = ws.dw_scratch; J.dbias = ws.cs;
  } else {
    SfWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = ws.g_bf2; a.ldy = D; a.x = sv.t_out; a.ldx = D; a.M = M; a.N1 = D; a.N2 = D; a.ldo = D; a.alpha = 1.f;
    a.partial = ws.wg_partial; a.out = ws.dw_scratch; a.accumulate = 0;
    a.dbias = ws.cs; a.dbias_scratch = ws.cs_partial;
    HIP_TRY(sf_launch_wgrad(a, s));
    HIP_TRY(sf_launch_gate_grad(ws.dw_scratch, ws.cs, PP(t, P0, l.t_dense.pw), PP(t, P0, l.t_dense.pb), PP(t, P0, l.gate),
                                GG(t, c.grads, l.t_dense.pw), GG(t, c.grads, l.t_dense.pb), GG(t, c.grads, l.gate), t->red_partial, D, D, s));
  }
  if (dp || hd) HIP_TRY(sf_launch_rowscale_bf16(ws.d_tout, ws.d_tout, dp, M, D, 0, T, N, s, site(0)));     // through the drop_path / dropout in front of temporal_dense
  HIP_TRY(lin_dgrad(l.t_out, ws.d_tout, M, s, nullptr, ws.d_ctx));
  HIP_TRY(lin_wgrad_queued(c, q, l.t_out, ws.d_tout, sv.ctx_t, M));
  }
  {
    SfAttnBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.qkv = sv.tqkv; a.ld_qkv = 3 * D; a.o = sv.ctx_t; a.ld_o = D; a.d_o = ws.d_ctx; a.d_qkv = ws.d_wide_t;
    a.heads = t->heads; a.D = D; a.scale = 0.125f; a.L = T; a.nseq = B * N; a.seq_rows = N;
    a.causal = t->cfg.enable_causal_temporal;
    if (t->f_drop_attn > 0.f) a.drop = sf_drop_make(t->f_drop_attn, t->f_drop_seed, (unsigned)(li * 8 + 4));
    HIP_TRY(sf_launch_temporal_attention_bwd(a, s));
  }
  HIP_TRY(lin_wgrad_queued(c, q, l.t_qkv, ws.d_wide_t, sv.ln_t, M));
  HIP_TRY(lin_dgrad(l.t_qkv, ws.d_wide_t, M, s, nullptr, ws.d_ln_bf));
  if (q.g.njobs > 0) HIP_TRY(sf_launch_wgrad_group(q.g, s));
  bool small_forked = false;
  if (dense_queued && tfuse) {
    sf_trainer* tm = const_cast<sf_trainer*>(t);
    if (side_ok && side_stream_ready(tm)) {
      // on the side stream, behind the grouped launch; joined one layer later (small_join) — the LoRA work enqueued before it is
      // marked complete first (ev_join), so that the end-of-layer join below does not wait for these launches
      if (forked) HIP_TRY(hipEventRecord(tm->ev_join, tm->side));
      HIP_TRY(hipEventRecord(tm->ev_fork, s));
      HIP_TRY(hipStreamWaitEvent(tm->side, tm->ev_fork, 0));
      BwdCtx cs2 = c;
      cs2.s = tm->side; cs2.on_side = true;
      HIP_TRY(temporal_fused_grads(cs2, l, D, g1buf, csbuf));
      HIP_TRY(hipEventRecord(tm->ev_small[par], tm->side));
      tm->small_pending |= 1 << par;
      small_forked = true;
    } else {
      HIP_TRY(temporal_fused_grads(c, l, D, g1buf, csbuf));
    }
  }
  if (dense_queued && !tfuse)
    HIP_TRY(sf_launch_gate_grad(ws.dw_scratch, ws.cs, PP(t, P0, l.t_dense.pw), PP(t, P0, l.t_dense.pb), PP(t, P0, l.gate),
                                GG(t, c.grads, l.t_dense.pw), GG(t, c.grads, l.t_dense.pb), GG(t, c.grads, l.gate), t->red_partial, D, D, s));
  HIP_TRY(sf_launch_ln_bwd(ws.h[li], ws.d_ln_bf, 1, PP(t, P0, l.ln_t_g), ws.g, ws.g, ws.g_bf, GG(t, c.grads, l.ln_t_g), GG(t, c.grads, l.ln_t_b),
                           ws.ln_partial, M, D, eps, s));
  if (forked) {
    sf_trainer* tm = const_cast<sf_trainer*>(t);
    if (small_forked) HIP_TRY(hipStreamWaitEvent(s, tm->ev_join, 0));      // recorded above, in front of the small launches
    else HIP_TRY(side_join(c));
  }
  {   // the PREVIOUS layer's D x D algebra (other parity) must be done before the next layer reuses its G1 / cs buffers
    sf_trainer* tm = const_cast<sf_trainer*>(t);
    const int other = par ^ 1;
    if (tm->small_pending & (1 << other)) {
      HIP_TRY(hipStreamWaitEvent(s, tm->ev_small[other], 0));
      tm->small_pending &= ~(1 << other);
    }
  }
  return SF_OK;
}

static int backward_embeddings(const BwdCtx& c, int B, int T) {
  const sf_trainer* t = c.t;
  const TWs& ws = *c.ws;
  hipStream_t s = c.s;
  const int D = t->D, N = t->N;
  const int M = B * T * N;
  // h0 = patches W^T + b + pos[n] + time[t]   (modeling:336-350, 413-457)
  if (t->f_drop_hidden > 0.f) {
    // h0 = m_time o (m_pos o (patches W^T + b + pos) + time): the time table sees m_time o g, everything else m_pos o m_time o g
    HIP_TRY(sf_launch_dropout_f32(ws.g, nullptr, (size_t)M * D, sf_drop_make(t->f_drop_hidden, t->f_drop_seed, (unsigned)t->L * 8u + 1u), s));
    HIP_TRY(sf_launch_sum_rows(ws.g, ws.s_tn, T * N, T * N, 1, 0, B, (long)T * N, D, 0, s));
    if (float* gt = GG(t, c.grads, t->p_time)) HIP_TRY(sf_launch_sum_rows(ws.s_tn, gt, T, T, N, 0, N, 1, D, 1, s));
    HIP_TRY(sf_launch_dropout_f32(ws.g, ws.g_bf, (size_t)M * D, sf_drop_make(t->f_drop_hidden, t->f_drop_seed, (unsigned)t->L * 8u), s));
    HIP_TRY(lin_wgrad(c, t->patch, ws.g_bf, ws.patches, M));
    HIP_TRY(sf_launch_sum_rows(ws.g, ws.s_tn, T * N, T * N, 1, 0, B, (long)T * N, D, 0, s));
    if (float* gp = GG(t, c.grads, t->p_pos)) HIP_TRY(sf_launch_sum_rows(ws.s_tn, gp, N, N, 1, 0, T, N, D, 1, s));
    return SF_OK;
  }
  HIP_TRY(lin_wgrad(c, t->patch, ws.g_bf, ws.patches, M));
  HIP_TRY(sf_launch_sum_rows(ws.g, ws.s_tn, T * N, T * N, 1, 0, B, (long)T * N, D, 0, s));       // sum over batch
  if (float* gp = GG(t, c.grads, t->p_pos)) HIP_TRY(sf_launch_sum_rows(ws.s_tn, gp, N, N, 1, 0, T, N, D, 1, s));
  if (float* gt = GG(t, c.grads, t->p_time)) HIP_TRY(sf_launch_sum_rows(ws.s_tn, gt, T, T, N, 0, N, 1, D, 1, s));
  return SF_OK;
}

extern "C" int sf_trainer_backward(sf_trainer* t, const float* d_pooler, const float* d_lhs, float* grads, int stage_first,
                                   int stage_last, void* workspace, size_t workspace_bytes, sf_stream stream) {
  if (!t || !grads || !workspace) return sf_set_err(SF_ERR_INVALID, "null argument");
  if (!t->fB) return sf_set_err(SF_ERR_STATE, "sf_trainer_backward needs a preceding sf_trainer_forward");
  if (stage_first < 0 || stage_last > t->L + 1 || stage_first > stage_last) return sf_set_err(SF_ERR_INVALID, "bad stage range");
  if (stage_first == 0 && !d_pooler) return sf_set_err(SF_ERR_INVALID, "stage 0 needs d_pooler");
  HIP_TRY(hipSetDevice(t->device));
  const int B = t->fB, T = t->fT;
  const TWs ws = tcarve(t, workspace, B, T);
  if (workspace_bytes < ws.bytes) return sf_set_err(SF_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, ws.bytes);
  BwdCtx c{t, &ws, grads, (hipStream_t)stream};
  for (int st = stage_first; st <= stage_last; ++st) {
    int rc;
    if (st == 0) rc = backward_head(c, d_pooler, d_lhs, B, T);
    else if (st == t->L + 1) rc = backward_embeddings(c, B, T);
    else rc = backward_layer(c, t->L - st, B, T);
    if (rc) return rc;
  }
  // every gradient slice of the stages just run must be complete in the caller's stream order when this call returns (the
  // caller all-reduces them): join what is still on the side stream
  for (int p = 0; p < 2; ++p)
    if (t->small_pending & (1 << p)) {
      HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, t->ev_small[p], 0));
      t->small_pending &= ~(1 << p);
    }
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// optimizer
// ------------------------------------------------------------------------------------------------
extern "C" int sf_trainer_adamw_step(sf_trainer* t, float* params, float* grads, float* m, float* v, int step, float lr,
                                     float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                     const float* grad_sumsq_dev, float clip_norm, int zero_grads, sf_stream stream) {
  if (!t || !params || !grads || !m || !v) return sf_set_err(SF_ERR_INVALID, "null argument");
  if (step < 1) return sf_set_err(SF_ERR_INVALID, "step counts from 1");
  HIP_TRY(hipSetDevice(t->device));
  SfAdamWArgs a;
  a.p = params; a.g = grads; a.m = m; a.v = v; a.n = t->n_train;
  a.seg_end = t->seg_end; a.seg_decay = t->seg_decay; a.seg_train = t->seg_train; a.nseg = t->nseg;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  a.bias_correction1 = 1.0f - powf(beta1, (float)step);
  a.bias_correction2 = 1.0f - powf(beta2, (float)step);
  a.grad_scale = grad_scale;
  a.clip_sumsq = grad_sumsq_dev; a.clip_norm = clip_norm; a.zero_grads = zero_grads;
  a.extra_seg0 = t->extra_seg0;
  a.n_extra = t->extra_steps_set ? t->n_extra : 0;
  for (int i = 0; i < 64; ++i) a.extra_steps[i] = t->extra_steps[i];
  if (grad_sumsq_dev && !(clip_norm > 0.f)) return sf_set_err(SF_ERR_INVALID, "clip_norm must be positive");
  a.guard_flag = t->guard_flag; a.guard_sumsq = nullptr; a.guard_loss = t->guard_loss;
  if (t->guard_flag) {
    // the clip pass already holds sum g^2 of this gradient; otherwise one deterministic pass over the trainable prefix (~0.1 ms)
    if (grad_sumsq_dev) a.guard_sumsq = grad_sumsq_dev;
    else {
      if (!t->guard_sumsq) HIP_TRY(hipMalloc((void**)&t->guard_sumsq, sizeof(float)));
      HIP_TRY(sf_launch_sumsq(grads, t->n_train, t->guard_sumsq, t->red_partial, (hipStream_t)stream));
      a.guard_sumsq = t->guard_sumsq;
    }
  }
  HIP_TRY(sf_launch_adamw(a, (hipStream_t)stream));
  return SF_OK;
}

extern "C" int sf_trainer_set_dropout(sf_trainer* t, float hidden_p, float attn_p, uint32_t seed) {
  if (!t) return sf_set_err(SF_ERR_INVALID, "null argument");
  if (!(hidden_p >= 0.f && hidden_p < 1.f) || !(attn_p >= 0.f && attn_p < 1.f)) return sf_set_err(SF_ERR_INVALID, "dropout probabilities must be in [0, 1)");
  t->drop_hidden = hidden_p; t->drop_attn = attn_p; t->drop_seed = seed;
  return SF_OK;
}

extern "C" int sf_trainer_set_nonfinite_guard(sf_trainer* t, int32_t* flag_dev, const float* loss_dev) {
  if (!t) return sf_set_err(SF_ERR_INVALID, "null argument");
  t->guard_flag = (int*)flag_dev;
  t->guard_loss = flag_dev ? loss_dev : nullptr;
  return SF_OK;
}

extern "C" int sf_trainer_set_drop_path(sf_trainer* t, const float* scales_dev, int B, int T) {
  if (!t) return sf_set_err(SF_ERR_INVALID, "null argument");
  if (scales_dev && (B <= 0 || T <= 0)) return sf_set_err(SF_ERR_INVALID, "bad geometry B=%d T=%d", B, T);
  t->dp_scales = scales_dev; t->dp_B = B; t->dp_T = T;
  return SF_OK;
}

extern "C" int sf_trainer_set_extra_steps(sf_trainer* t, const int32_t* steps, int n) {
  if (!t) return sf_set_err(SF_ERR_INVALID, "null argument");
  if (!steps || n == 0) { t->extra_steps_set = false; return SF_OK; }
  if (n != t->n_extra) return sf_set_err(SF_ERR_INVALID, "sf_trainer_set_extra_steps: %d entries for %d slots", n, t->n_extra);
  for (int i = 0; i < n; ++i) {
    if (steps[i] < 0) return sf_set_err(SF_ERR_INVALID, "negative step count");
    t->extra_steps[i] = steps[i];
  }
  t->extra_steps_set = true;
  return SF_OK;
}

extern "C" int sf_trainer_grad_sumsq(sf_trainer* t, const float* grads, float* out, sf_stream stream) {
  if (!t || !grads || !out) return sf_set_err(SF_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(t->device));
  HIP_TRY(sf_launch_sumsq(grads, t->n_train, out, t->red_partial, (hipStream_t)stream));
  return SF_OK;
}

// ------------------------------------------------------------------------------------------------
// single backward operators (parity tests)
// ------------------------------------------------------------------------------------------------
extern "C" int sf_op_wgrad(const void* dy, int ldy, const void* x, int ldx, int M, int N1, int N2, float alpha, int accumulate,
                           float* out, int ldo, float* dbias, sf_stream stream) {
  if (!dy || !x || !out) return sf_set_err(SF_ERR_INVALID, "null argument");
  float* partial = nullptr;
  HIP_TRY(hipMalloc(&partial, (sf_wgrad_partial_floats(M, N1, N2) + sf_colsum_partial_floats(N1)) * sizeof(float)));
  SfWgradArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = (const bf16_t*)dy; a.ldy = ldy; a.x = (const bf16_t*)x; a.ldx = ldx; a.M = M; a.N1 = N1; a.N2 = N2;
  a.out = out; a.ldo = ldo; a.accumulate = accumulate; a.alpha = alpha; a.partial = partial;
  a.dbias = dbias; a.dbias_scratch = partial + sf_wgrad_partial_floats(M, N1, N2);
  hipError_t e = sf_launch_wgrad(a, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(partial);
  if (e != hipSuccess) return sf_set_err(SF_ERR_HIP, "sf_op_wgrad: %s", hipGetErrorString(e));
  return SF_OK;
}

extern "C" int sf_op_attention_bwd(const void* qkv, const void* o, const void* d_o, void* d_qkv, int layout, int nseq, int L,
                                   int seq_rows, int heads, int causal, sf_stream stream) {
  if (!qkv || !o || !d_o || !d_qkv) return sf_set_err(SF_ERR_INVALID, "null argument");
  SfAttnBwdArgs a;
  memset(&a, 0, sizeof(a));
  const int D = heads * 64;
  a.qkv = (const bf16_t*)qkv; a.ld_qkv = 3 * D; a.o = (const bf16_t*)o; a.ld_o = D; a.d_o = (const bf16_t*)d_o;
  a.d_qkv = (bf16_t*)d_qkv; a.heads = heads; a.D = D; a.scale = 0.125f; a.L = L; a.nseq = nseq; a.seq_rows = seq_rows;
  a.causal = causal;
  HIP_TRY(layout == 0 ? sf_launch_spatial_attention_bwd(a, (hipStream_t)stream) : sf_launch_temporal_attention_bwd(a, (hipStream_t)stream));
  return SF_OK;
}

extern "C" int sf_op_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* g_in, float* dx,
                                   float* d_gamma, float* d_beta, int rows, int D, float eps, sf_stream stream) {
  if (!x || !dy || !gamma || !dx) return sf_set_err(SF_ERR_INVALID, "null argument");
  float* partial = nullptr;
  HIP_TRY(hipMalloc(&partial, sf_ln_bwd_partial_floats(D) * sizeof(float)));
  hipError_t e = sf_launch_ln_bwd(x, dy, 0, gamma, g_in, dx, nullptr, d_gamma, d_beta, partial, rows, D, eps, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(partial);
  if (e != hipSuccess) return sf_set_err(SF_ERR_HIP, "sf_op_layernorm_bwd: %s", hipGetErrorString(e));
  return SF_OK;
}
