// Kernel launchers of the training step (backward pass + optimizer) — SURVEY.md §8 f-1.
// Reference semantics: torch autograd through modeling_timesformer_siglip.py:934-1004 (layer),
// :1141-1154 (pooling head), :413-457 (embeddings); AdamW as run_finetuning_multi_task.py:429-433 /
// optim_factory.py:59-104 configure it.
#pragma once
#include "sf_common.h"

// ------------------------------------------------------------------------------------------------
// weight-gradient GEMM ("TN"):  C[N1,N2] (+)= sum_m dY[m,N1] * X[m,N2]
// Both operands are token-major bf16 exactly as the forward/backward kernels leave them; the
// contraction runs over token rows, fragments come out of row-major LDS tiles through
// ds_read_b64_tr_b16.  Split over M with fp32 partials + a deterministic reduction.
// ------------------------------------------------------------------------------------------------
struct SfWgradArgs {
  const bf16_t* dy; int ldy;       // [M, ldy], columns [0, N1)
  const bf16_t* x; int ldx;        // [M, ldx], columns [0, N2)
  int M, N1, N2;
  float* out; int ldo;             // [N1, ldo] fp32
  int accumulate;                  // out += result (else out = result)
  float alpha;                     // result scaled by alpha
  float* partial;                  // scratch, >= sf_wgrad_partial_floats(...) floats
  float* dbias;                    // optional: dbias[N1] += alpha * column sums of dY (bias gradient)
  float* dbias_scratch;            // >= sf_colsum_partial_floats(N1) floats (used when the tile kernel cannot fuse it)
};
size_t sf_wgrad_partial_floats(int M, int N1, int N2);
hipError_t sf_launch_wgrad(const SfWgradArgs& a, hipStream_t s);

// Several weight gradients over the SAME token rows in one launch (the Linears of one encoder layer): their 256 x 256
// tiles fill the chip together, so the token range is split 2-ways instead of 7...28-ways per projection — 5x fewer
// launches and ~10x fewer fp32 partial bytes per layer.  Every job needs N1 % 256 == 0 and N2 % 256 == 0.
#define SF_WG_MAX_JOBS 8
struct SfWgradJob {
  const bf16_t* dy; const bf16_t* x;   // [M, ldy] / [M, ldx]
  float* out; float* dbias;            // [N1, ldo] fp32; optional bias gradient (+= alpha * column sums of dY)
  int ldy, ldx, N1, N2, ldo, accumulate;
  float alpha;
  int tile0, tiles2;                   // filled by the launcher: first tile of the job, tiles along N2
  unsigned part_off, bias_off;         // filled by the launcher: float offsets inside one split's partial / bias block
};
struct SfWgradGroup {
  SfWgradJob job[SF_WG_MAX_JOBS];
  int njobs, M;
  float* partial;                      // >= sf_wgrad_group_partial_floats(M, sum of tiles, sum of N1) floats
};
bool sf_wgrad_groupable(int M, int N1, int N2);
size_t sf_wgrad_group_partial_floats(int M, int ntiles, int sum_n1);
hipError_t sf_launch_wgrad_group(SfWgradGroup& g, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// attention backward (softmax(QK^T * scale [+ causal mask]) V), bf16 storage, fp32 math
// ------------------------------------------------------------------------------------------------
struct SfAttnBwdArgs {
  const bf16_t* qkv; int ld_qkv;   // token rows [rows, 3D]: q | k | v column blocks
  const bf16_t* o;  int ld_o;      // forward context [rows, D]
  const bf16_t* d_o;               // gradient wrt context [rows, D]   (same pitch as o)
  bf16_t* d_qkv;                   // [rows, 3D] (same pitch as qkv)
  int heads, D;
  float scale;
  int L;                           // sequence length (spatial: patches per frame; temporal: frames)
  int nseq;                        // spatial: frames; temporal: B * N sequences
  int seq_rows;                    // temporal only: N (token row of (b, t, n) = (b*L + t)*N + n)
  int causal;
  const float* lse2;               // spatial, optional: [nseq, heads, L] log-sum-exp (base 2) saved by the forward kernel;
                                   // without it the backward recomputes the row statistics (phase A)
  int lab;                         // timing lab (SF_ATTN_BWD_LAB): 1 no phase B, 2 no phase C, 4 no output stores
  SfDrop drop;                     // attention-probability dropout the forward applied (on = 0: none); element index ((seq * heads + h) * L + q) * L + k
};
hipError_t sf_launch_spatial_attention_bwd(const SfAttnBwdArgs& a, hipStream_t s);    // L <= 224
hipError_t sf_launch_temporal_attention_bwd(const SfAttnBwdArgs& a, hipStream_t s);   // L <= 32
// (pooling head backward: sf_pool_head.h)

// ------------------------------------------------------------------------------------------------
// row-wise / elementwise
// ------------------------------------------------------------------------------------------------
// act = gelu(pre)   (erf form, modeling:819-824)
hipError_t sf_launch_gelu_fwd(const bf16_t* pre, bf16_t* act, size_t n, hipStream_t s);
// drop_path factors per sample group (mode 0 temporal (b, n), 1 spatial (b, t), 2 MLP (b)) and / or an elementwise dropout mask;
// scales == nullptr: no drop_path factor.  See sf_train_kernels.hip
hipError_t sf_launch_rowscale_bf16(const bf16_t* in, bf16_t* out, const float* scales, int rows, int D, int mode, int T, int N, hipStream_t s,
                                   SfDrop drop = SfDrop{0u, 0u, 0u, 1.f});
hipError_t sf_launch_resid_rowscale(float* out, const float* resid, const float* y, const float* scales, int rows, int D, int mode, int T, int N,
                                    hipStream_t s, SfDrop drop = SfDrop{0u, 0u, 0u, 1.f});
// embeddings with dropout (modeling:374, 378): h = m_time o (m_pos o h + time[t]) in place; h arrives as patches W^T + b + pos[n]
hipError_t sf_launch_embed_dropout(float* h, const float* time_rows, int M, int D, int T, int N, SfDrop pos_drop, SfDrop time_drop, hipStream_t s);
// g = m o g in place on fp32 rows, optional bf16 copy of the result (embedding backward)
hipError_t sf_launch_dropout_f32(float* g, bf16_t* g_bf, size_t n, SfDrop drop, hipStream_t s);
// d = d * gelu'(pre)   in place
hipError_t sf_launch_gelu_bwd(bf16_t* d, const bf16_t* pre, size_t n, hipStream_t s);
// LayerNorm backward over rows of x (statistics recomputed): g_out = (g_in ? g_in : 0) + dL/dx, optionally
// also as a bf16 copy (the A operand of the next input-gradient GEMM);
// d_gamma += sum_rows dy * xhat, d_beta += sum_rows dy.   partial: >= sf_ln_bwd_partial_floats(D)
size_t sf_ln_bwd_partial_floats(int D);
hipError_t sf_launch_ln_bwd(const float* x, const void* dy /* fp32 or bf16 [rows, D] */, int dy_is_bf16, const float* gamma,
                            const float* g_in, float* g_out, bf16_t* g_out_bf, float* d_gamma, float* d_beta, float* partial, int rows, int D, float eps,
                            hipStream_t s);
// out[c] += alpha * sum_r x[r, c]   (bias gradients); partial >= sf_colsum_partial_floats(cols)
size_t sf_colsum_partial_floats(int cols);
hipError_t sf_launch_colsum_bf16(const bf16_t* x, int rows, int cols, int ld, float alpha, float* out,
                                 int accumulate, float* partial, hipStream_t s);
// out[o, :] (+)= sum_{r < R} in[(o % n_a) * stride_a + (o / n_a) * stride_b + r * stride_r, :]  (fp32 rows of D)
hipError_t sf_launch_sum_rows(const float* in, float* out, int n_out, int n_a, long stride_a, long stride_b,
                              int R, long stride_r, int D, int accumulate, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// weights: fp32 master -> bf16 working copies
// ------------------------------------------------------------------------------------------------
// per weight: w_eff = scale * (w + lora_b * lora_a) [N,K] -> w_bf [N,K] and wT_bf [K,N] (either may be null);
// scale = tanh(*gate) when a gate is given; bias_out = scale * bias.  All weights of the model go in ONE
// launch through a job table; offsets are floats from `base` (-1 = absent)
#define SF_PREP_TILE 64      // weight-refresh tile edge (sf_prep_weights_batched_kernel)
struct SfPrepJob {
  long w_off, la_off, lb_off, gate_off, bias_off;
  bf16_t* w_bf; bf16_t* wT_bf; float* bias_out;
  int N, K, rank;
  int tile0;                 // first workgroup of this job (32x32 tiles, k fastest)
};
hipError_t sf_launch_prep_weights_batched(const float* base, const SfPrepJob* jobs_dev, int njobs, int total_tiles,
                                          hipStream_t s);
// pooling-head query: q[D] = (probe * Wq^T + bq) * scale     (modeling:1145-1149 with nn.MultiheadAttention)
hipError_t sf_launch_head_query(const float* probe, const float* wq, const float* bq, float scale, float* q, int D,
                                hipStream_t s);
// ... and its backward: dWq += scale * dq (x) probe, dbq += scale * dq, dprobe += scale * Wq^T dq
hipError_t sf_launch_head_query_bwd(const float* dq, const float* probe, const float* wq, float scale, float* d_wq,
                                    float* d_bq, float* d_probe, int D, hipStream_t s);
// temporal gate: h1 = h + tanh(g) * (t_out W^T + b).  G = unscaled dW, cs = unscaled db (colsum of dL/dh1):
//   dW += tanh(g) G, db += tanh(g) cs, dgate += (1 - tanh(g)^2) * (<G, W> + <cs, b>)
//   r1 != nullptr: G is taken as G + cs (x) r1 (the bias term of the fused temporal projections)
hipError_t sf_launch_gate_grad(const float* G, const float* cs, const float* w, const float* b, const float* gate,
                               float* d_w, float* d_b, float* d_gate, float* partial /* >= 128 floats */, int N, int K,
                               hipStream_t s, const float* r1 = nullptr);
// temporal_dense o temporal_attention.output.dense as one projection of the training step (sf_train_kernels.hip): per layer
//   wf [D, D] = (tanh(g) W_d) W_o as bf16, wfT its transpose, bf [D] = tanh(g) (W_d b_o + b_d)
struct SfFuseJob {
  const bf16_t* wd; const bf16_t* woT;     // bf16 working copies: tanh(g) W_d [D, D], W_o^T [D, D]
  bf16_t* wf; bf16_t* wfT; float* bf;
  long wd_off, bo_off, bd_off, gate_off;   // fp32 parameter offsets (floats) for the bias
};
hipError_t sf_launch_fuse_temporal(const float* base, const SfFuseJob* jobs_dev, int layers, int D, hipStream_t s);
// out[k] += sum_i w[i, k] v[i]  (w bf16 [rows, ld])
hipError_t sf_launch_matvec_t_bf16(const bf16_t* w, int ld, const float* v, float* out, int rows, int cols, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// optimizer
// ------------------------------------------------------------------------------------------------
struct SfAdamWArgs {
  float* p; float* g; float* m; float* v;         // flat fp32 [n]
  size_t n;
  const int* seg_end;        // device: exclusive end offset of each segment (ascending), nseg entries
  const unsigned char* seg_decay;   // device: 1 = weight decay applies
  const unsigned char* seg_train;   // device: 1 = trainable (others untouched)
  int nseg;
  float lr, beta1, beta2, eps, weight_decay;
  float bias_correction1, bias_correction2;   // 1 - beta^t
  float grad_scale;                            // g is multiplied by this first (averaging: 1 / world)
  // clip_grad_norm_ without a host round trip: when clip_sumsq != nullptr the kernel reads sum g^2 of the (summed)
  // gradient from device memory, total_norm = sqrt(sum) * grad_scale, and multiplies g by min(1, clip_norm / (total_norm + 1e-6))
  const float* clip_sumsq; float clip_norm;
  int zero_grads;                              // 1: g is cleared by the same pass (optimizer.zero_grad fused)
  // the last n_extra trainable segments are scalar slots with their own step counts (0 = skipped this step):
  // torch.optim.AdamW keeps `step` per parameter and skips parameters without a gradient
  int extra_seg0, n_extra;                     // n_extra = 0: no per-slot handling
  int extra_steps[64];
  // non-finite guard (tools/finetune_tools.py:533-541 stops on a non-finite loss; utils.py:515-551 skips the step on inf grads):
  // guard_flag != nullptr -> the kernel reads sum g^2 (guard_sumsq) and, when given, the loss scalar; if either is inf / NaN the
  // whole update is skipped (p, m, v untouched; g still cleared when zero_grads) and guard_flag = {1 (sticky), skipped steps + 1}.
  // Nothing synchronises with the host: the flag is read at the caller's next host touch.
  int* guard_flag; const float* guard_sumsq; const float* guard_loss;
};
hipError_t sf_launch_adamw(const SfAdamWArgs& a, hipStream_t s);
// out[0] = sum g^2 (deterministic two-stage); partial >= 1024 floats
hipError_t sf_launch_sumsq(const float* g, size_t n, float* out, float* partial, hipStream_t s);
