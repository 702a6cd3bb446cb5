// Attention-pooling head without the key / value projections of the tokens (sf_pool_head.hip; modeling:1141-1154).
#pragma once
#include "sf_common.h"

// U_h = Wk_h^T q_h: fp32 [16, D] and hi / lo bf16 planes [16, D] (rows >= heads zero).  wk = in_proj_weight rows [D, 2D),
// q = the projected, scaled probe [D]
hipError_t sf_launch_pool_u(const float* wk, const float* q, float* u, bf16_t* u_hi, bf16_t* u_lo, int heads, int D, hipStream_t s);

struct SfPoolArgs {
  const float* x;                          // [F * N, D] fp32: the normalised tokens (post_layernorm output)
  const float* const* x_ind;               // optional: x is read from device memory (the streamed frame's caller tensor)
  const bf16_t* u_hi; const bf16_t* u_lo;  // [16, D]
  float* zpart;                            // [F, S, heads, D] weighted token sums of each split (normalised when S == 1)
  float* ml;                               // [F, S, heads, 2] {max score, sum of exp} per split (needed when S > 1)
  float* probs;                            // optional (S == 1): softmax probabilities [F, heads, N] fp32, kept for the backward
  int probs_raw;                           // probs holds the RAW scores and ml the {max, sum} per head: the backward finishes the softmax itself
  int F, N, heads, D, S, normalize;
};
int sf_pool_splits(int F, int N, int heads);                       // token splits per frame the launcher wants for this shape
size_t sf_pool_z_floats(int F, int N, int heads, int D);
size_t sf_pool_ml_floats(int F, int N, int heads);
hipError_t sf_launch_pool_probe(const SfPoolArgs& a, hipStream_t s);

struct SfPoolCtxArgs {
  const float* zpart; const float* ml;     // as written by sf_launch_pool_probe
  const float* wv; int ldw;                // value projection rows [D][ldw] fp32 (in_proj_weight rows [2D, 3D))
  const float* bv;                         // [D]
  float* z_out;                            // optional, S > 1: combined normalised sums [F, heads, D]
  float* ctx_f32; bf16_t* ctx_hi; bf16_t* ctx_lo;      // [F, D] outputs (any subset)
  int F, heads, D, S;
};
hipError_t sf_launch_pool_ctx(const SfPoolCtxArgs& a, hipStream_t s);

// Generic widths (head_dim != 64 or D > 1024): the whole probe attention of a frame in one fp32 workgroup (sf_pool_head.hip, bottom)
struct SfPoolGenArgs {
  const float* x; const float* const* x_ind;     // as SfPoolArgs
  const float* u;                                // [heads, D] fp32 U_h = Wk_h^T q_h (q scaled by 1 / sqrt(head_dim))
  const float* wv; int ldw; const float* bv;     // value projection rows [D][ldw], bias [D]
  float* ctx_f32; bf16_t* ctx_hi; bf16_t* ctx_lo;  // [F, D] outputs (any subset)
  float* scratch;                                // sf_pool_generic_scratch_floats(F, N, heads, D) floats: scores [F, heads, N] + z [F, heads, D]
  float* scores; float* z;                       // set by the launcher
  int F, N, heads, hd, D;
};
bool sf_pool_generic_supported(int N, int heads, int D);
size_t sf_pool_generic_scratch_floats(int F, int N, int heads, int D);
hipError_t sf_launch_pool_generic(const SfPoolGenArgs& a, hipStream_t s);

// One-to-four-row Linear of the head's per-frame tail (a streamed frame: F = streams <= 4): y = act(LN?(x) W^T + b) (+ resid) with fp32
// activations and hi + lo bf16 weights multiplied out in fp32 FMAs; one wave per output column, the whole K range in flight.
struct SfRowLinArgs {
  const float* x; int ldx;                 // [F, K] fp32
  const float* ln_g; const float* ln_b; float ln_eps;      // optional: LayerNorm of the rows first
  const bf16_t* w_hi; const bf16_t* w_lo;  // [N, K]
  const float* bias;                       // [N] or null
  const float* resid; int ldr;             // optional [F, N]
  float* out; int ldo;                     // [F, N] fp32
  float* const* out_ind;                   // optional: the destination is read from device memory (replaces out)
  int F, N, K, act;                        // act: -1 none, else the hidden_act code (0 erf-gelu, 1 tanh-gelu, 2 relu)
};
bool sf_rowlin_supported(int F, int N, int K);
hipError_t sf_launch_rowlin(const SfRowLinArgs& a, hipStream_t s);

// backward of ctx = Wv z + bv: dz [F, heads, D]; dwv / dbv accumulate (+=), either may be null.  wT = transposed bf16 working copy
// [D][ldt] whose value columns start at col0
hipError_t sf_launch_pool_ctx_bwd(const float* dctx, const bf16_t* wT, int ldt, int col0, const float* z, float* dz, float* dwv, int ldw,
                                  float* dbv, int F, int heads, int D, hipStream_t s);
struct SfPoolBwdArgs {
  const bf16_t* x_bf;                      // [F * N, D] bf16 normalised tokens
  const float* probs;                      // [F, heads, N] probabilities, or (probs_raw) raw scores with ml = {max, sum} [F, heads, 2]
  const float* ml; int probs_raw;          // ml [F, ml_splits, heads, 2] as the forward's token splits wrote it
  int ml_splits;
  const float* z; const float* dz;         // [F, heads, D]
  const float* u;                          // [16, D] fp32
  const float* d_lhs;                      // optional [F * N, D]: gradient arriving through last_hidden_state, added to dx
  float* dx;                               // [F * N, D] fp32 gradient wrt the normalised tokens (written)
  bf16_t* ds_bf;                           // [F * N, 32] bf16: score gradients, the dY operand of dU = ds^T x (columns >= heads zero)
  int F, N, heads, D;
};
hipError_t sf_launch_pool_probe_bwd(const SfPoolBwdArgs& a, hipStream_t s);
// dWk += q dU^T (null: skipped), dq = Wk dU;  du [>= heads, D] fp32
hipError_t sf_launch_pool_u_bwd(const float* du, const float* wk, const float* q, float* dwk, float* dq, int D, hipStream_t s);
