/*
 * streamformer_hip.h — C ABI of the MI355X (gfx950) StreamFormer encoder hot path.
 *
 * The reference has no FFI seam: the path is a torch.nn.Module,
 * TimesformerMultiTaskingModelSigLIP (reference models/modeling_timesformer_siglip.py:1241-1354).
 * This header is the boundary a binding would use instead of that module's forward; the Python
 * mirror of the module (streamformer_amd/modeling.py) binds it with ctypes, INTEGRATION.md shows
 * the stub.  Each entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns 0 on
 *     success or a negative sf_status; sf_last_error() gives the message of the last failure on the
 *     calling thread.
 *   - All device buffers passed in (pixels, outputs, workspace) are CALLER-allocated and
 *     caller-owned; the library never frees them and never synchronises the stream.  Weights and
 *     KV-caches are library-owned device memory.
 *   - All work is enqueued on the caller's hipStream_t (pass torch.cuda.current_stream()).
 *   - A handle is bound to one device and may be used from one thread at a time.
 *   - Layout: activations are FRAME-major [B, T, N, D] row-major (token row = (b*T + t)*N + n),
 *     not the reference's patch-major (B, N*T, D) (modeling:452-457).
 */
#ifndef STREAMFORMER_HIP_H
#define STREAMFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sf_encoder sf_encoder;   /* opaque: config + packed weights on one device        */
typedef struct sf_cache sf_cache;       /* opaque: temporal K/V cache of one stream (all layers) */
typedef void* sf_stream;                /* hipStream_t                                          */

typedef enum {
  SF_OK = 0,
  SF_ERR_INVALID = -1,      /* bad argument / unsupported configuration                      */
  SF_ERR_STATE = -2,        /* call order (e.g. forward before finalize, missing weight)     */
  SF_ERR_HIP = -3,          /* HIP runtime error (message carries hipGetErrorString)         */
  SF_ERR_WORKSPACE = -4,    /* workspace too small                                           */
  SF_ERR_UNKNOWN_KEY = -5,  /* sf_load_tensor: not a weight of this model (ignored keys OK)  */
  SF_ERR_CAPACITY = -6      /* streaming past the cache / time-embedding capacity            */
} sf_status;

typedef enum { SF_F32 = 0, SF_BF16 = 1, SF_F16 = 2, SF_F64 = 3, SF_U8 = 4 } sf_dtype;

/* Arithmetic mode of the matrix products (reported as "dtype" by bench.py):
 *   SF_COMPUTE_BF16   : bf16 operands, fp32 accumulate, one MFMA pass            (throughput)
 *   SF_COMPUTE_BF16X3 : x = xh+xl, w = wh+wl, xh*wh + xh*wl + xl*wh, fp32 acc   (fp32-accurate)
 * Residual stream, LayerNorm statistics, softmax and GELU are fp32 in both.                   */
typedef enum { SF_COMPUTE_BF16 = 0, SF_COMPUTE_BF16X3 = 1 } sf_compute;

/* Plain-old-data mirror of StreamformerConfig (reference models/configuration_streamformer.py:90-135). */
typedef struct {
  int32_t image_size, patch_size, num_channels, num_frames;
  int32_t hidden_size, num_hidden_layers, num_attention_heads, intermediate_size;
  int32_t hidden_act;              /* 0 = "gelu" (exact erf); 1 = "gelu_new"/tanh; 2 = "relu" */
  int32_t qkv_bias;                /* bool */
  int32_t enable_causal_temporal;  /* bool: causal (modeling:887) vs bidirectional (:894)     */
  int32_t add_lora_spatial;        /* bool: expect *_lora_{a,b}.weight (rank 32, modeling:1280) */
  float layer_norm_eps;
} sf_config;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* replaces TimesformerMultiTaskingModelSigLIP.__init__ (modeling:1244-1258) */
int sf_create(const sf_config* cfg, int device, sf_encoder** out);
void sf_destroy(sf_encoder* enc);
const char* sf_last_error(void);
int sf_abi_version(void);

/* ---- weights: replaces from_pretrained()/load_state_dict() (modeling:1066-1075) -------------
 * `key` is the reference state_dict key, without any "timesformer." prefix.
 * The host buffer is borrowed only for the duration of the call.  Keys that are not weights of
 * this model return SF_ERR_UNKNOWN_KEY (callers may ignore it for buffers such as "...mask").   */
int sf_load_tensor(sf_encoder* enc, const char* key, const void* host_ptr, int dtype,
                   const int64_t* shape, int ndim);
/* Packs weights for the kernels and uploads them.  merge_lora: fold W += B*A (inference).
 * fuse_temporal_proj: fold temporal_dense o temporal_attention.output.dense into one matrix
 * (modeling:947-954 are two Linear layers with nothing in between).                            */
int sf_finalize_weights(sf_encoder* enc, int compute, int merge_lora, int fuse_temporal_proj);
/* number of weight tensors still missing (0 when complete); names via sf_last_error()          */
int sf_missing_weights(sf_encoder* enc);
/* TimesformerImageProcessor's arithmetic for SF_U8 frames (vqa_enc:1400-1447): y = (x * rescale - mean[c]) / std[c].
 * Defaults: mean = std = 0.5, rescale = 1/255.  The bicubic resize stays with the caller.          */
int sf_set_pixel_normalization(sf_encoder* enc, const float* mean, const float* std, int channels, float rescale);

/* ---- full-clip forward: replaces .forward(pixel_values) (modeling:1299-1354) -----------------
 * pixels_dev         [B,T,C,H,W] contiguous, dtype pixel_dtype: SF_F32 / SF_BF16 = normalised frames;
 *                    SF_U8 = raw frames, the image processor's rescale + normalize (see
 *                    sf_set_pixel_normalization) is fused into the patch-extraction kernel
 * last_hidden_dev    fp32 [B,T,N,D]   (post-LayerNorm tokens, modeling:1330,1342-1346)
 * pooler_dev         fp32 [B,T,D]     (modeling:1338-1340)
 * hidden_states_dev  NULL, or fp32 [L+1,B,T,N,D]: the input of every layer + the last output
 *                    (modeling:1031-1051), FRAME-major (the Python mirror permutes views)
 * pos_dev            NULL to use the loaded position table (needs H==W==image_size), else an
 *                    fp32 [N',D] table already resized on the host (modeling:380-411)
 */
int sf_workspace_bytes(sf_encoder* enc, int B, int T, int H, int W, size_t* out);
int sf_forward(sf_encoder* enc, const void* pixels_dev, int pixel_dtype, int B, int T, int H, int W,
               float* last_hidden_dev, float* pooler_dev, float* hidden_states_dev,
               const float* pos_dev, void* workspace_dev, size_t workspace_bytes, sf_stream stream);

/* Measurement hook: ONE forward (same arguments as sf_forward, no hidden states) with HIP events around the launches of four kernel
 * classes INSIDE it — 0: N = 768 residual projection at K = hidden_size, 1: the same at K = intermediate_size, 2: spatial attention,
 * 3: temporal attention.  Synchronises the stream.  out_ms_host[2 c] = mean milliseconds per launch of class c (the event pair also
 * spans the launch boundary in front of the kernel), out_ms_host[2 c + 1] = number of launches.  bench.py's `roofline` uses it.   */
int sf_forward_profile(sf_encoder* enc, const void* pixels_dev, int pixel_dtype, int B, int T, int H, int W,
                       float* last_hidden_dev, float* pooler_dev, void* workspace_dev, size_t workspace_bytes,
                       sf_stream stream, float* out_ms_host);

/* same, additionally returning the spatial attention probabilities of every layer
 * (output_attentions=True, modeling:703-716, 1052-1057): attentions_dev fp32 [L, B*T, heads, N, N]
 * (any N the attention kernels take: above 224 patches the streamed-keys kernel makes a second sweep).  */
int sf_forward_attentions(sf_encoder* enc, const void* pixels_dev, int pixel_dtype, int B, int T, int H, int W,
                          float* last_hidden_dev, float* pooler_dev, float* hidden_states_dev,
                          float* attentions_dev, const float* pos_dev, void* workspace_dev,
                          size_t workspace_bytes, sf_stream stream);

/* ---- the forward in three stages, for callers that interleave their own modules ------------------
 * (reference sub-module users: `blk(x, T, output_attentions=False)[0]` over encoder.layer in the
 * ViT-Adapter interaction blocks, models/modeling_timesformer_siglip_adapter.py:424-425; the
 * embeddings -> encoder -> post_layernorm -> head sequence of the video classifier,
 * downstream/AR/models/modeling_timesformer_video_classification.py:121-134).
 * hidden_dev: the caller's residual stream, fp32 FRAME-major [B,T,N,D] (the reference's patch-major
 * (B, N*T, D) is a permuted view of it); sf_layers updates it in place.
 * attentions_dev (optional): fp32 [layer_end-layer_begin, B*T, heads, N, N].  Workspace: sf_workspace_bytes. */
int sf_embed(sf_encoder* enc, const void* pixels_dev, int pixel_dtype, int B, int T, int H, int W,
             float* hidden_out_dev, const float* pos_dev, void* workspace_dev, size_t workspace_bytes,
             sf_stream stream);                                         /* TimesformerEmbeddingsSigLIP.forward, modeling:413-457 */
int sf_layers(sf_encoder* enc, float* hidden_dev, int B, int T, int H, int W, int layer_begin, int layer_end,
              float* attentions_dev, void* workspace_dev, size_t workspace_bytes, sf_stream stream);
                                                                        /* TimesformerLayerSigLIP.forward x (end-begin), modeling:934-1004 */
int sf_post_head(sf_encoder* enc, float* hidden_dev, int B, int T, int H, int W, float* last_hidden_dev,
                 float* pooler_dev, void* workspace_dev, size_t workspace_bytes, sf_stream stream);
                                                                        /* post_layernorm + head, modeling:1330-1340, 1141-1154;
                                                                           last_hidden_dev == NULL: head alone on normalised tokens */

/* ---- streaming forward with a temporal KV-cache ----------------------------------------------
 * replaces forward(..., past_key_values, use_cache=True) of the VideoQA copy
 * (reference downstream/VideoQA/llava/model/multimodal_encoder/timesformer_encoder.py:1316-1392;
 * cache update :517-518, offset causal mask :522-546, time-embedding offset :328-369).
 * The cache holds, per layer, the temporal K/V rows of every frame seen so far.                 */
int sf_cache_create(sf_encoder* enc, int B, int max_frames, int H, int W, sf_cache** out);
int sf_cache_reset(sf_cache* cache);            /* TimesformerVisionTower.clear_cache (:1528) */
int sf_cache_length(const sf_cache* cache);
/* Bounded-memory policy of a stream that outlives the cache (choose while the cache is empty): 0 (default) = stop at capacity
 * with SF_ERR_CAPACITY — the reference raises at config.num_frames (timesformer_encoder.py:343-348); 1 = SLIDING WINDOW: once
 * `max_frames` frames are cached every further single-frame call overwrites the oldest one, its temporal query sees the last
 * `max_frames` frames, and frames past the time-embedding table reuse its last row.  An extension beyond the reference
 * (SURVEY.md section 8 f-2 asks for a bounded-memory policy); sf_cache_length keeps counting the frames seen.                 */
int sf_cache_set_policy(sf_cache* cache, int policy);     /* DynamicCache.get_seq_length()              */
size_t sf_cache_bytes(const sf_cache* cache);
void sf_cache_destroy(sf_cache* cache);
int sf_stream_workspace_bytes(sf_encoder* enc, const sf_cache* cache, int T_new, size_t* out);
/* hidden_states_dev: NULL, or fp32 [L+1, B, T_new, N, D] — the new frames' input to every layer + the last output
 * (output_hidden_states=True together with use_cache=True: the vision tower's call form, vqa_enc:1536).
 * A cache is tied to the weight packing it was created against: after another sf_finalize_weights on the same
 * handle (or a new handle at a recycled address) sf_forward_stream returns SF_ERR_STATE instead of touching it. */
int sf_forward_stream(sf_encoder* enc, sf_cache* cache, const void* pixels_dev, int pixel_dtype,
                      int T_new, float* last_hidden_dev, float* pooler_dev, float* hidden_states_dev,
                      const float* pos_dev, void* workspace_dev, size_t workspace_bytes, sf_stream stream);
/* The same call with output_attentions (timesformer_encoder.py:494, 557, 633, 659, 720-754): attentions_dev receives the
 * spatial attention probabilities of the NEW frames, [L, B * T_new, heads, N, N] fp32, as sf_forward_attentions
 * returns them for whole clips.  Runs the launches eagerly (no graph replay).                                            */
int sf_forward_stream_attentions(sf_encoder* enc, sf_cache* cache, const void* pixels_dev, int pixel_dtype, int T_new,
                                 float* last_hidden_dev, float* pooler_dev, float* hidden_states_dev, float* attentions_dev,
                                 const float* pos_dev, void* workspace_dev, size_t workspace_bytes, sf_stream stream);

/* ---- single operators (each is one kernel of the path; used by the parity tests) ----------- */
/* nn.LayerNorm(D, eps) rows (modeling:860-865,878-880,1251): x fp32 [rows,D] -> y fp32 [rows,D] */
int sf_op_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev,
                    int rows, int D, float eps, sf_stream stream);
/* nn.Linear (+ optional exact-erf GELU, + optional residual): y = act(x W^T + b) [+ alpha*() + r]
 * x fp32 [M,K], w fp32 [N,K], b fp32 [N] or NULL, resid fp32 [M,N] or NULL, y fp32 [M,N].
 * Operands are rounded/split on device exactly as the encoder does for `compute`.               */
int sf_op_linear(const float* x_dev, const float* w_dev, const float* b_dev, const float* resid_dev,
                 float alpha, int gelu, float* y_dev, int M, int N, int K, int compute,
                 void* workspace_dev, size_t workspace_bytes, sf_stream stream);
size_t sf_op_linear_workspace_bytes(int M, int N, int K);
/* softmax(q k^T / sqrt(d)) v per head over `groups` independent sequences
 * (modeling:688-717 spatial; :575-615 temporal with causal=1 and past offset).
 * qkv fp32 [groups, Lq|Lk, 3*D] packed as the qkv Linear emits it; ctx fp32 [groups, Lq, D].
 * Spatial: seq stride = rows are contiguous tokens.  Temporal goes through the same entry with
 * `row_stride` (in rows) between consecutive sequence positions.                                */
int sf_op_attention(const float* qkv_dev, float* ctx_dev, int groups, int L, int heads, int head_dim,
                    int causal, int temporal_layout, int N_tokens, int compute,
                    void* workspace_dev, size_t workspace_bytes, sf_stream stream);
size_t sf_op_attention_workspace_bytes(int groups, int L, int heads, int head_dim);

/* ---- loss heads of the multitask pre-training step (BASELINE config #3) ---------------------
 * retrieval: TimesformerVideoRetrievalHead.forward + SigLipLoss._loss (modeling:2324-2351,221-237)
 * localization: TimesformerUniversalLocalizationHead.forward (modeling:2238-2282)
 * pooler fp32 [B,T,D]; text fp32 [Bt,D] (un-normalised); label_emb fp32 [L,D]; labels int32 [B,T]
 * (-1 = background).  pos_offset: column of `text` that is row 0's positive (rank*B when `text` is
 * the all-gathered [world*B, D] table, modeling:250-280; -1 = negatives only).
 * logit_scale_dev / logit_bias_dev: DEVICE pointers to one fp32 each (the heads' parameters, modeling:1363-1364;
 * in a training step they point into the flat parameter buffer, so no host round trip sits between
 * forward and backward).  workspace_dev: caller-owned scratch of sf_loss_workspace_bytes(B, T) bytes
 * (per-row partial sums, reduced in a fixed order: the losses are bit-reproducible).
 * There is no limit on Bt (the text table is walked in chunks); D <= 2048, L <= 4096 (SF_ERR_CAPACITY).
 * Outputs: loss_dev fp32 [1]; grad_pooler_dev fp32 [B,T,D] or NULL;
 * grad_scalars_dev fp32 [2] = d loss / d (logit_scale, logit_bias) or NULL.                     */
size_t sf_loss_workspace_bytes(int B, int T);
int sf_retrieval_loss(const float* pooler_dev, const float* text_dev, int B, int T, int D, int Bt,
                      int pos_offset, const float* logit_scale_dev, const float* logit_bias_dev,
                      float* loss_dev, float* grad_pooler_dev, float* grad_scalars_dev,
                      void* workspace_dev, size_t workspace_bytes, sf_stream stream);
int sf_localization_loss(const float* pooler_dev, const float* label_emb_dev, const int32_t* labels_dev,
                         int B, int T, int D, int L, const float* logit_scale_dev, const float* logit_bias_dev,
                         float* loss_dev, float* grad_pooler_dev, float* grad_scalars_dev,
                         void* workspace_dev, size_t workspace_bytes, sf_stream stream);

/* ---- training step (BASELINE configs #3 / #4) ------------------------------
 * Replaces, for one micro-batch: the autograd graph of TimesformerMultiTaskingModelSigLIP.forward
 * (modeling:1299-1354) as driven by train_one_epoch_multi_task (tools/finetune_tools.py:395-573:
 * forward -> task-head loss -> loss/update_freq -> backward -> optimizer step every update_freq
 * micro-steps) and torch.optim.AdamW as optim_factory.py:59-104 configures it (no weight decay for
 * 1-D parameters and "*.bias"; frozen parameters skipped).
 *
 * State layout: ALL parameters live in ONE caller-owned flat fp32 device buffer (so the gradient
 * buffer of the same layout can be all-reduced by RCCL in a few large slices); each
 * segment starts on a multiple of 64 floats.  Trainable parameters come first, in model order
 * (embeddings, layers 0..L-1, post_layernorm, head, extra scalars), frozen ones (the spatial
 * qkv / output.dense base weights when freeze_spatial=1, modeling:1284-1297) after them.
 * Names are the reference state_dict keys; `extra.<i>` are caller-defined trainable
 * scalars (task heads' logit_scale / logit_bias, modeling:1363-1364).
 * Compute mode is SF_COMPUTE_BF16 (bf16 MFMA operands, fp32 accumulation, fp32 master weights,
 * fp32 residual stream and its gradient).                                                        */
typedef struct sf_trainer sf_trainer;
int sf_trainer_create(const sf_config* cfg, int device, int freeze_spatial, int n_extra, sf_trainer** out);
void sf_trainer_destroy(sf_trainer* tr);
int sf_trainer_num_params(const sf_trainer* tr);
/* shape_out: up to 4 dims; trainable/decay: the optim_factory.py:70-77 grouping                 */
int sf_trainer_param_info(const sf_trainer* tr, int index, char* name_out, int name_cap, int64_t* offset_out,
                          int64_t* numel_out, int64_t* shape_out, int* ndim_out, int* trainable_out,
                          int* decay_out);
/* total floats of the flat buffer, and the length of its trainable prefix                        */
int sf_trainer_total_floats(const sf_trainer* tr, int64_t* total_out, int64_t* trainable_out);
/* backward runs in stages so the caller can all-reduce finished gradient slices while earlier
 * layers are still being differentiated: stage 0 = pooling head + post_layernorm,
 * stage 1+k = layer L-1-k, stage L+1 = embeddings.  The slice [offset, offset+numel) of the
 * gradient buffer is final when the stage returns.                                              */
int sf_trainer_num_stages(const sf_trainer* tr);
int sf_trainer_stage_range(const sf_trainer* tr, int stage, int64_t* offset_out, int64_t* numel_out);
/* fp32 master -> bf16 working weights (row-major and transposed copies, LoRA merged as
 * W + B A (modeling:541-545), temporal_dense scaled by tanh(gate) (modeling:954-958)).  Call after
 * every optimizer step; params_dev must stay valid until the next call.                          */
int sf_trainer_sync_weights(sf_trainer* tr, const float* params_dev, sf_stream stream);
int sf_trainer_workspace_bytes(const sf_trainer* tr, int B, int T, size_t* out);
/* drop_path (stochastic depth, modeling:460-486, 846-856) of the forwards that follow: scales_dev = DEVICE array, per layer
 * [B*N temporal | B*T spatial | B MLP] factors (0 = branch dropped for that sample group, 1 / keep_prob = kept), L layers back to
 * back; the reference draws one Bernoulli per dim-0 entry of the tensor each branch returns ((B*N,T,D), (B*T,N,D), (B,N*T,D)).
 * The backward of a forward applies the factors that forward used.  NULL = none (eval, or drop_path_rate 0).  The array is
 * caller-owned and must stay valid until the matching backward has run.                                                        */
int sf_trainer_set_drop_path(sf_trainer* tr, const float* scales_dev, int B, int T);
/* Dropout of the forwards that follow (config.hidden_dropout_prob / attention_probs_dropout_prob; reference sites modeling:374, 378
 * (position / time embeddings), 752 / 761 (both SelfOutput projections), 822 (MLP activation), 835 (MLP output), 556 / 603 / 669 / 705
 * (attention probabilities)).  Masks are COUNTER-BASED: element idx of site k is kept iff hash(idx, hash(k, seed)) < keep * 2^32 and
 * scaled by 1 / keep, so no mask tensor exists and the backward of a forward replays the masks from the seed that forward used
 * (the CPU oracle evaluates the same integer hash).  0 / 0 switches dropout off (eval).                                           */
int sf_trainer_set_dropout(sf_trainer* tr, float hidden_p, float attention_p, uint32_t seed);
/* Non-finite guard of the optimizer step (tools/finetune_tools.py:533-541 stops the run on a non-finite loss; the GradScaler of
 * utils.py:515-551 skips a step whose gradients hold inf / NaN).  flag_dev = DEVICE int32[2], caller-owned and zero-initialised:
 * every sf_trainer_adamw_step that follows checks sum g^2 of the gradient (and *loss_dev, a device float, when not NULL) ON THE
 * DEVICE; if either is inf / NaN the update is skipped as a whole (parameters and moments untouched, gradients still cleared
 * when zero_grads) and flag_dev = {1 (sticky), number of skipped steps}.  No host synchronisation: the caller reads the flag at
 * its next host touch.  flag_dev = NULL switches the guard off.                                                              */
int sf_trainer_set_nonfinite_guard(sf_trainer* tr, int32_t* flag_dev, const float* loss_dev);
/* forward with every activation the backward needs kept in the workspace                         */
int sf_trainer_forward(sf_trainer* tr, const void* pixels_dev, int pixel_dtype, int B, int T,
                       float* last_hidden_dev, float* pooler_dev, void* workspace_dev,
                       size_t workspace_bytes, sf_stream stream);
/* grads_dev += d loss / d params for stages [stage_first, stage_last]; d_pooler_dev fp32 [B,T,D],
 * d_last_hidden_dev fp32 [B,T,N,D] or NULL.  Must follow sf_trainer_forward on the same workspace. */
int sf_trainer_backward(sf_trainer* tr, const float* d_pooler_dev, const float* d_last_hidden_dev,
                        float* grads_dev, int stage_first, int stage_last, void* workspace_dev,
                        size_t workspace_bytes, sf_stream stream);
/* torch.optim.AdamW update of the trainable prefix; `step` counts from 1; grad_scale multiplies
 * the gradient first (1/world for averaging).  grad_sumsq_dev (optional, DEVICE pointer to the
 * output of sf_trainer_grad_sumsq): torch.nn.utils.clip_grad_norm_(max_norm = clip_norm) applied
 * inside the kernel — total_norm = sqrt(sum) * grad_scale, coefficient min(1, clip_norm / (total_norm + 1e-6)) —
 * so clipping needs no host synchronisation.  zero_grads != 0: grads_dev is cleared by the same pass
 * (optimizer.zero_grad(), tools/finetune_tools.py:566).                                            */
int sf_trainer_adamw_step(sf_trainer* tr, float* params_dev, float* grads_dev, float* exp_avg_dev,
                          float* exp_avg_sq_dev, int step, float lr, float beta1, float beta2, float eps,
                          float weight_decay, float grad_scale, const float* grad_sumsq_dev, float clip_norm,
                          int zero_grads, sf_stream stream);
/* Per-slot step counts of the `n_extra` scalar slots ("extra.<i>": the task heads' logit_scale / logit_bias) for
 * the adamw steps that follow: steps_host[i] > 0 = update slot i with bias corrections of that step count,
 * 0 = leave slot i untouched (its gradient is still cleared).  torch.optim.AdamW skips parameters whose .grad is
 * None and counts `step` per parameter: a head whose task was not scheduled in an accumulation window gets
 * neither weight decay nor moment decay (tools/finetune_tools.py:560-570 + zero_grad(set_to_none)).  n = 0
 * (or steps_host NULL) restores the default: every slot follows the step passed to sf_trainer_adamw_step.     */
int sf_trainer_set_extra_steps(sf_trainer* tr, const int32_t* steps_host, int n);
/* out_dev[0] = sum of squares of the trainable gradient prefix (for clip_grad_norm_)             */
int sf_trainer_grad_sumsq(sf_trainer* tr, const float* grads_dev, float* out_dev, sf_stream stream);

/* single backward operators (parity tests).  bf16 tensors are raw uint16 device buffers.         */
/* C[N1,N2] = alpha * dY^T X  (+ C) : dy [M,ldy], x [M,ldx] bf16; out fp32 [N1,ldo];
 * dbias_dev (optional) fp32 [N1] += alpha * column sums of dY (the bias gradient of the same Linear) */
int sf_op_wgrad(const void* dy_dev, int ldy, const void* x_dev, int ldx, int M, int N1, int N2, float alpha,
                int accumulate, float* out_dev, int ldo, float* dbias_dev, sf_stream stream);
/* attention backward; layout 0 = spatial (nseq sequences of L consecutive token rows),
 * 1 = temporal (token row of (b, t, n) = (b*L + t)*seq_rows + n, nseq = B*seq_rows).
 * qkv/d_qkv bf16 [rows, 3D], o/d_o bf16 [rows, D].                                               */
int sf_op_attention_bwd(const void* qkv_dev, const void* o_dev, const void* d_o_dev, void* d_qkv_dev, int layout,
                        int nseq, int L, int seq_rows, int heads, int causal, sf_stream stream);
/* LayerNorm backward: dx = g_in + dLN(x; dy), d_gamma/d_beta accumulated                          */
int sf_op_layernorm_bwd(const float* x_dev, const float* dy_dev, const float* gamma_dev, const float* g_in_dev,
                        float* dx_dev, float* d_gamma_dev, float* d_beta_dev, int rows, int D, float eps,
                        sf_stream stream);

/* ---- introspection for bench/roofline ------------------------------------------------------- */
/* Enqueue `iters` back-to-back launches of the dominant GEMM (the MLP up-projection shape of the
 * loaded model at M rows) between two HIP events on `stream` and return the mean launch time.   */
int sf_bench_gemm(sf_encoder* enc, int M, int which, int iters, void* workspace_dev,
                  size_t workspace_bytes, sf_stream stream, float* mean_ms_out, double* flops_out);

/* Same for the attention kernels at the loaded model's shape: which = 0 spatial (B*T frames of N
 * tokens), 1 temporal (B*N sequences of T frames).  bytes_out = algorithmic bytes per launch
 * (read q,k,v once + write ctx once, in the storage type of the compute mode).                  */
int sf_bench_attention(sf_encoder* enc, int B, int T, int which, int iters, void* workspace_dev,
                       size_t workspace_bytes, sf_stream stream, float* mean_ms_out, double* bytes_out,
                       double* flops_out);

/* Environment switches (A/B and tuning knobs of the measurements; csrc/sf_switches.h holds the one table): the library reads them
 * once, at first use.  sf_reload_switches() re-reads the environment (tests flip a switch inside a process);
 * sf_switch_info(i, 0) / (i, 1) return name / description of switch i, NULL past the end.                                      */
void sf_reload_switches(void);
const char* sf_switch_info(int index, int what);

/* Launch floor of this device: `launches` dependent EMPTY kernels (256 workgroups of 256 threads) captured into one hipGraph and
 * replayed `iters` times; mean microseconds per launch.  What a chain of dependent launches costs when the kernels do nothing —
 * the yardstick next to the streamed frame's ~100-launch graph (bench.py `streaming.launch_floor`).                              */
int sf_bench_launch_floor(int device, int launches, int iters, sf_stream stream, float* us_per_launch_out);

#ifdef __cplusplus
}
#endif
#endif /* STREAMFORMER_HIP_H */
