"""CPU ORACLE — test infrastructure, NOT product code.

A plain-torch (fp32 or fp64, CPU) restatement of the reference's
``TimesformerMultiTaskingModelSigLIP`` forward (``/root/reference/models/
modeling_timesformer_siglip.py:1299-1354``), its KV-cached streaming variant
(``/root/reference/downstream/VideoQA/llava/model/multimodal_encoder/timesformer_encoder.py``,
"vqa_enc" below) and the two loss heads BASELINE.json config #3 names.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product package ``streamformer_amd`` never does (and fails loudly without its HIP library).

How it is pinned: the reference has no tests or golden vectors for this path, so
``oracle/make_golden.py`` imports the reference in the build container, loads the SAME seeded
state_dict into both, asserts this restatement equals the reference (fp32, <=2e-5 max-abs;
fp64 restatement vs fp32 reference agrees to fp32 round-off) and writes the reference's outputs to
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` re-checks the restatement against those
files on every run, so the oracle that travels to the GPU box is the one the reference vouched for.

The arithmetic is torch ATen CPU (addmm/bmm/softmax/erf-gelu/layer_norm) exactly like the
reference's (torch pin 2.5.1 there, 2.10.0 here); what is restated is the op *sequence* and data
layout.  The layout is deliberately different from the reference's: the residual stream is kept
frame-major ``[B, T, N, D]`` (the reference keeps patch-major ``(B, N*T, D)``, modeling:452-457),
which is how the HIP path stores it; ``hidden_states`` are permuted back on request.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------------
def _lin(x: Tensor, sd: Dict[str, Tensor], name: str) -> Tensor:
    b = sd.get(name + ".bias")
    return F.linear(x, sd[name + ".weight"], b)


def _ln(x: Tensor, sd: Dict[str, Tensor], name: str, eps: float) -> Tensor:
    # nn.LayerNorm(D, eps=layer_norm_eps): modeling:860-865, 878-880, 1251, 1138
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _act(cfg, x: Tensor) -> Tensor:
    # ACT2FN[config.hidden_act] (modeling:814-817).  "gelu" is the exact erf form.
    a = cfg.hidden_act
    if a == "gelu":
        return F.gelu(x)
    if a in ("gelu_new", "gelu_pytorch_tanh"):
        return F.gelu(x, approximate="tanh")
    if a == "relu":
        return F.relu(x)
    if a == "selu":
        return F.selu(x)
    raise ValueError(f"unsupported hidden_act {a!r}")


def _lora_lin(x: Tensor, sd: Dict[str, Tensor], name: str, lora_name: str) -> Tensor:
    """``dense(x) + lora_b(lora_a(x))`` with no scaling factor: modeling:536-551, 649-664, 748-754."""
    y = _lin(x, sd, name)
    a = sd.get(lora_name + "_lora_a.weight")
    if a is not None:
        y = y + F.linear(F.linear(x, a), sd[lora_name + "_lora_b.weight"])
    return y


def _mha(q: Tensor, k: Tensor, v: Tensor, heads: int, mask: Optional[Tensor], prob_mask: Optional[Tensor] = None) -> Tensor:
    """softmax((q k^T) * d^-0.5 [+ mask]) v per head.  q:[G,Lq,D] k,v:[G,Lk,D] -> [G,Lq,D].

    Follows modeling:577-609 / 690-711: scores are scaled AFTER the matmul, masked positions are
    filled with -inf (modeling:599-601) before the softmax.
    """
    G, Lq, D = q.shape
    Lk = k.shape[1]
    d = D // heads
    qh = q.reshape(G, Lq, heads, d).transpose(1, 2)
    kh = k.reshape(G, Lk, heads, d).transpose(1, 2)
    vh = v.reshape(G, Lk, heads, d).transpose(1, 2)
    s = (qh @ kh.transpose(-2, -1)) * (d ** -0.5)
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    p = s.softmax(dim=-1)
    pd = p if prob_mask is None else p * prob_mask          # attn_drop (modeling:556, 603, 669, 705): [G, heads, Lq, Lk] factors
    return (pd @ vh).transpose(1, 2).reshape(G, Lq, D), p


# --------------------------------------------------------------------------------------------
# embeddings  (modeling:336-350, 380-457;  vqa_enc:307-375 for the streaming offset)
# --------------------------------------------------------------------------------------------
def time_embedding_rows(sd, cfg, t_past: int, t_new: int, streaming: bool, clamp: bool = False) -> Tensor:
    """Rows of the time-embedding table used for frames ``t_past .. t_past+t_new-1`` -> [t_new, D].

    Full clip (modeling:435-450): slice when T < num_frames, ``interpolate(mode="nearest")`` when
    T > num_frames (index map floor(t * num_frames / T)).  Streaming (vqa_enc:328-369): direct
    rows while ``t_past+t_new <= num_frames``; the reference raises past that, so
    does this restatement.
    """
    te = sd["embeddings.time_embeddings"][0]  # [num_frames, D]
    nf = te.shape[0]
    total = t_past + t_new
    if streaming and clamp:      # sliding-window extension (NOT in the reference): frames past the table reuse its last row
        return te[torch.arange(t_past, total).clamp(max=nf - 1)]
    if streaming:
        if total > nf:
            raise ValueError(
                f"streaming past config.num_frames={nf} time-embedding rows (needed {total}); the "
                "reference raises here too (vqa_enc:343-348)")
        return te[t_past:total]
    assert t_past == 0
    if t_new <= nf:
        return te[:t_new]
    idx = torch.floor(torch.arange(t_new, dtype=torch.float64) * (nf / t_new)).long().clamp_(max=nf - 1)
    return te[idx]


def position_embedding(sd, cfg, H: int, W: int) -> Tensor:
    """[N', D] position table for an H x W input (modeling:380-411)."""
    pe = sd["embeddings.position_embeddings"]  # [1, N, D]
    N, D = pe.shape[1], pe.shape[2]
    P = cfg.patch_size
    npatch = (H // P) * (W // P)
    if npatch == N and W == H:
        return pe[0]
    M = int(math.sqrt(N))
    assert N == M * M
    w0, h0 = W // P, H // P
    grid = pe.float().reshape(1, M, M, D).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(w0, h0), mode="bicubic", antialias=True)
    return grid.permute(0, 2, 3, 1).reshape(-1, D).to(pe.dtype)


def patchify(pixels: Tensor, P: int) -> Tensor:
    """[B,T,C,H,W] -> [B,T,N,C*P*P] with columns in (c, ph, pw) order and n = row*(W/P)+col.

    Conv2d(k=P, s=P) (modeling:329-334, 344-348) is exactly this matrix times
    ``weight.reshape(D, C*P*P)^T``.
    """
    B, T, C, H, W = pixels.shape
    gh, gw = H // P, W // P
    x = pixels[..., : gh * P, : gw * P].reshape(B, T, C, gh, P, gw, P)
    return x.permute(0, 1, 3, 5, 2, 4, 6).reshape(B, T, gh * gw, C * P * P)


# --------------------------------------------------------------------------------------------
# dropout masks (training): the reference draws them with torch's Philox stream (nn.Dropout at modeling:374, 378, 556, 603, 669, 705,
# 752, 761, 822, 835), which no other implementation can replay; here — as in the build's kernels (streamformer_amd/csrc/sf_train.h:
# sf_drop_hash / sf_drop_make) — a mask is a pure function of (seed, site, element index), so both sides of a test use the same draw.
# --------------------------------------------------------------------------------------------
_M32 = 0xFFFFFFFF


def _drop_hash(idx: Tensor, key: int) -> Tensor:
    x = (idx * 0x9E3779B1 + key) & _M32          # int64 products wrap modulo 2^64: the low 32 bits are exact
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & _M32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & _M32
    return x ^ (x >> 16)


def dropout_mask(shape, p: float, seed: int, site: int, dtype=torch.float32) -> Tensor:
    """0 or 1 / keep per element of a frame-major tensor of ``shape`` (flat C-order index), site keys as in sf_train.h."""
    n = 1
    for d in shape:
        n *= int(d)
    key = int(_drop_hash(torch.tensor([site], dtype=torch.int64), int(seed) & _M32)[0])
    keep = 1.0 - float(p)
    thresh = min(int(keep * 4294967296.0), 4294967295)
    h = _drop_hash(torch.arange(n, dtype=torch.int64), key)
    scale = torch.tensor(1.0 / keep, dtype=torch.float32)
    return ((h < thresh).to(torch.float32) * scale).reshape(*shape).to(dtype)


def embeddings(sd, cfg, pixels: Tensor, t_past: int = 0, streaming: bool = False, clamp_time: bool = False,
               dropout: Optional[tuple] = None) -> Tensor:
    """``dropout`` = (seed, hidden_p, attention_p) in training: pos_drop / time_drop of modeling:374, 378 (sites L*8, L*8 + 1)."""
    B, T, C, H, W = pixels.shape
    w = sd["embeddings.patch_embeddings.projection.weight"]
    x = patchify(pixels, cfg.patch_size) @ w.reshape(w.shape[0], -1).t()
    x = x + sd["embeddings.patch_embeddings.projection.bias"]
    x = x + position_embedding(sd, cfg, H, W)[None, None]          # broadcast over B, T
    hp = dropout[1] if dropout is not None else 0.0
    if hp > 0:
        x = x * dropout_mask(x.shape, hp, dropout[0], cfg.num_hidden_layers * 8, x.dtype)
    if cfg.attention_type != "space_only":
        x = x + time_embedding_rows(sd, cfg, t_past, T, streaming, clamp_time)[None, :, None, :]
        if hp > 0:
            x = x * dropout_mask(x.shape, hp, dropout[0], cfg.num_hidden_layers * 8 + 1, x.dtype)
    return x  # [B, T, N, D]


# --------------------------------------------------------------------------------------------
# one encoder layer (modeling:900-1004), frame-major
# --------------------------------------------------------------------------------------------
def layer_forward(sd, cfg, i: int, h: Tensor, kv: Optional[Dict[str, Tensor]] = None,
                  collect: Optional[dict] = None, window: Optional[int] = None, drop_path: Optional[Tensor] = None,
                  dropout: Optional[tuple] = None) -> Tensor:
    """h: [B, T, N, D].  ``kv`` (streaming): dict with 'k','v' tensors [B, T_past, N, D] or empty.
    ``drop_path`` (training): this layer's factors [B*N + B*T + B] — 0 or 1/keep per dim-0 entry of the tensor each residual
    branch returns in the reference ((B*N,T,D) temporal, (B*T,N,D) spatial, (B,N*T,D) MLP; modeling:460-486, 949, 980, 1000);
    the reference draws them with torch.rand, here they are an input so that both sides of a test use the same draw."""
    B, T, N, D = h.shape
    heads = cfg.num_attention_heads
    eps = cfg.layer_norm_eps
    p = f"encoder.layer.{i}."
    # dropout = (seed, hidden_p, attention_p): sites i*8 + {0 temporal SelfOutput, 1 spatial SelfOutput, 2 MLP activation, 3 MLP output,
    # 4 temporal probabilities, 5 spatial probabilities}; element index = flat index of the frame-major tensor (sf_train.h)
    d_seed, d_hid, d_att = dropout if dropout is not None else (0, 0.0, 0.0)
    hmask = (lambda x, k: x * dropout_mask(x.shape, d_hid, d_seed, i * 8 + k, x.dtype)) if d_hid > 0 else (lambda x, k: x)

    if cfg.attention_type != "divided_space_time":
        # The other two TimeSformer modes (modeling:914-933).  StreamFormer only ever instantiates the divided branch (modeling:934-1004)
        # and the HIP path refuses these two; they are restated for the forward only (fixture F14), as the checker of a later round.
        #   space_only: the embeddings stay [B*T, N, D] (no time embedding, modeling:424) -> attention over the N patches of a frame
        #   joint_space_time: [B, N*T, D] -> attention over all tokens of a clip (their order does not matter to the result)
        # both: h = h + attention(layernorm_before(h)); h = h + mlp(layernorm_after(h)); the temporal parameters are unused
        if cfg.attention_type not in ("space_only", "joint_space_time"):
            raise NotImplementedError(cfg.attention_type)
        if kv is not None or drop_path is not None or dropout is not None:
            raise NotImplementedError(f"{cfg.attention_type}: forward only (no cache, no drop rates)")
        G, S = (B * T, N) if cfg.attention_type == "space_only" else (B, T * N)
        xs = _ln(h, sd, p + "layernorm_before", eps).reshape(G, S, D)
        qkv = _lora_lin(xs, sd, p + "attention.attention.qkv", p + "attention.attention.qkv")
        q, k, v = qkv.split(D, dim=-1)
        ctx, probs = _mha(q, k, v, heads, None, None)
        xs = _lora_lin(ctx, sd, p + "attention.output.dense", p + "attention.output.dense").reshape(B, T, N, D)
        h2 = h + xs
        out = h2 + _lin(_act(cfg, _lin(_ln(h2, sd, p + "layernorm_after", eps), sd, p + "intermediate.dense")), sd, p + "output.dense")
        if collect is not None:
            collect.setdefault("h1", []).append(h)
            collect.setdefault("h2", []).append(h2)
            collect.setdefault("attentions", []).append(probs)
        return out

    # ---- temporal attention over T for each (b, n): modeling:937-958 ------------------------
    xt = _ln(h, sd, p + "temporal_layernorm", eps)                  # [B,T,N,D]
    qkv = _lin(xt, sd, p + "temporal_attention.attention.qkv")
    q, k, v = qkv.split(D, dim=-1)
    t_past = 0
    if kv is not None:
        if "k" in kv:                                                # vqa_enc:517-518 cache.update
            t_past = kv["k"].shape[1]
            k = torch.cat([kv["k"], k], dim=1)
            v = torch.cat([kv["v"], v], dim=1)
        if window is not None and k.shape[1] > window:          # sliding-window extension: the oldest frames leave the cache;
            drop = k.shape[1] - window                          # the new queries keep their distance to the keys that stay
            k, v = k[:, drop:], v[:, drop:]
            t_past -= drop
        kv["k"], kv["v"] = k, v
    Tk = k.shape[1]
    to_bn = lambda z: z.permute(0, 2, 1, 3).reshape(B * N, z.shape[1], D)
    mask = None
    if cfg.enable_causal_temporal:                                   # modeling:594-601; vqa_enc:533-537
        qi = torch.arange(T)[:, None] + t_past
        mask = torch.arange(Tk)[None, :] <= qi                      # [T, Tk] True = keep
    pm = dropout_mask((B * N, heads, T, Tk), d_att, d_seed, i * 8 + 4, h.dtype) if d_att > 0 else None
    ctx, _ = _mha(to_bn(q), to_bn(k), to_bn(v), heads, mask, pm)
    ctx = ctx.reshape(B, N, T, D).permute(0, 2, 1, 3)
    att_t = hmask(_lin(ctx, sd, p + "temporal_attention.output.dense"), 0)          # SelfOutput dropout (modeling:761)
    if drop_path is not None:                                        # modeling:949: between the attention output and temporal_dense
        att_t = att_t * drop_path[:B * N].reshape(B, 1, N, 1).to(att_t.dtype)
    res_t = _lin(att_t, sd, p + "temporal_dense")
    h1 = h + torch.tanh(sd[p + "temporal_attention_gating"]) * res_t   # modeling:955-958

    # ---- spatial attention over N for each (b, t): modeling:962-996 --------------------------
    xs = _ln(h1, sd, p + "layernorm_before", eps).reshape(B * T, N, D)
    qkv = _lora_lin(xs, sd, p + "attention.attention.qkv", p + "attention.attention.qkv")
    q, k, v = qkv.split(D, dim=-1)
    pm = dropout_mask((B * T, heads, N, N), d_att, d_seed, i * 8 + 5, h.dtype) if d_att > 0 else None
    ctx, probs = _mha(q, k, v, heads, None, pm)
    xs = _lora_lin(ctx, sd, p + "attention.output.dense", p + "attention.output.dense")
    xs = hmask(xs.reshape(B, T, N, D), 1)                                          # modeling:752 / 761
    if drop_path is not None:                                        # modeling:980
        xs = xs * drop_path[B * N:B * N + B * T].reshape(B, T, 1, 1).to(xs.dtype)
    h2 = h1 + xs                                                     # residual onto h1 (modeling:993-996)

    # ---- MLP: modeling:997-1000, 819-837 ---------------------------------------------------------
    y = hmask(_lin(hmask(_act(cfg, _lin(_ln(h2, sd, p + "layernorm_after", eps), sd, p + "intermediate.dense")), 2),
                   sd, p + "output.dense"), 3)                                     # modeling:822, 835
    if drop_path is not None:                                        # modeling:1000
        y = y * drop_path[B * N + B * T:].reshape(B, 1, 1, 1).to(y.dtype)
    out = h2 + y
    if collect is not None:
        collect.setdefault("h1", []).append(h1)
        collect.setdefault("h2", []).append(h2)
        collect.setdefault("attentions", []).append(probs)
    return out


# --------------------------------------------------------------------------------------------
# pooling head (modeling:1141-1154; nn.MultiheadAttention with packed in_proj [q;k;v])
# --------------------------------------------------------------------------------------------
def pooling_head(sd, cfg, x: Tensor) -> Tensor:
    """x: [G, N, D] (post-LN tokens of G frames) -> [G, D]."""
    G, N, D = x.shape
    heads = cfg.num_attention_heads
    w, b = sd["head.attention.in_proj_weight"], sd["head.attention.in_proj_bias"]
    probe = sd["head.probe"].reshape(1, 1, D).expand(G, 1, D)
    q = F.linear(probe, w[:D], b[:D])
    k = F.linear(x, w[D:2 * D], b[D:2 * D])
    v = F.linear(x, w[2 * D:], b[2 * D:])
    ctx, _ = _mha(q, k, v, heads, None)
    a = _lin(ctx, sd, "head.attention.out_proj")                      # [G,1,D]
    y = _ln(a, sd, "head.layernorm", cfg.layer_norm_eps)
    y = _lin(_act(cfg, _lin(y, sd, "head.mlp.fc1")), sd, "head.mlp.fc2")
    return (a + y)[:, 0]


# --------------------------------------------------------------------------------------------
# whole forward
# --------------------------------------------------------------------------------------------
def cast_state_dict(sd: Dict[str, Tensor], dtype: torch.dtype) -> Dict[str, Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def to_patch_major(h: Tensor) -> Tensor:
    """[B,T,N,D] -> the reference's (B, N*T, D) token order (token = n*T + t, modeling:452-457)."""
    B, T, N, D = h.shape
    return h.permute(0, 2, 1, 3).reshape(B, N * T, D)


@torch.no_grad()
def forward(sd, cfg, pixels: Tensor, output_hidden_states: bool = False,
            collect: Optional[dict] = None, cache: Optional[List[dict]] = None, window: Optional[int] = None) -> Dict[str, Tensor]:
    """Inference entry: :func:`forward_graph` under ``torch.no_grad()``."""
    with torch.no_grad():
        return forward_graph(sd, cfg, pixels, output_hidden_states, collect, cache, window)


def forward_graph(sd, cfg, pixels: Tensor, output_hidden_states: bool = False,
                  collect: Optional[dict] = None, cache: Optional[List[dict]] = None, window: Optional[int] = None,
                  drop_path: Optional[Tensor] = None, dropout: Optional[tuple] = None) -> Dict[str, Tensor]:
    """Full-clip forward (``cache is None``) or one streaming call (``cache`` = list of per-layer dicts).

    Returns ``last_hidden_state [B,T,N,D]``, ``pooler_output [B,T,D]`` and, on request,
    ``hidden_states``: L+1 tensors in the reference's patch-major ``(B, N*T, D)`` order
    (modeling:1031-1051, 1352).
    """
    pixels = pixels.to(next(iter(sd.values())).dtype)
    streaming = cache is not None
    t_past = 0
    if streaming and cache and "k" in cache[0]:
        t_past = cache[0]["k"].shape[1]
    if window is not None:
        # ``window`` = the build's sliding-window cache policy (an EXTENSION, the reference raises at num_frames,
        # vqa_enc:343-348): absolute frame count kept next to the truncated K / V, time rows clamped to the table
        t_past = cache[0].get("seen", 0)
        cache[0]["seen"] = t_past + pixels.shape[1]
    h = embeddings(sd, cfg, pixels, t_past=t_past, streaming=streaming, clamp_time=window is not None, dropout=dropout)
    if collect is not None:
        collect["embeddings"] = h
    hs = []
    for i in range(cfg.num_hidden_layers):
        if output_hidden_states:
            hs.append(to_patch_major(h))
        h = layer_forward(sd, cfg, i, h, kv=(cache[i] if streaming else None), collect=collect, window=window,
                          drop_path=None if drop_path is None else drop_path[i], dropout=dropout)
        if collect is not None:
            collect.setdefault("layer_out", []).append(h)
    if output_hidden_states:
        hs.append(to_patch_major(h))
    B, T, N, D = h.shape
    seq = _ln(h, sd, "post_layernorm", cfg.layer_norm_eps)            # modeling:1330
    if cfg.attention_type == "space_only":
        # The reference's tail (modeling:1333-1347) reads the encoder output as patch-major (B, N, T, D) in every mode, but in this
        # mode the encoder ran on the frame-major (B*T, N, D) embeddings: row (t, n) of a clip's output is its frame-major row number
        # n*T + t.  Reproduced as is — both outputs of this (never instantiated) mode mix frames the same way the reference's do.
        seq = seq.reshape(B, N, T, D).permute(0, 2, 1, 3).contiguous()
        if output_hidden_states:
            raise NotImplementedError("space_only: hidden_states are not restated")
    pooled = pooling_head(sd, cfg, seq.reshape(B * T, N, D)).reshape(B, T, D)
    out = {"last_hidden_state": seq, "pooler_output": pooled}
    if output_hidden_states:
        out["hidden_states"] = hs
    return out


def new_cache(cfg) -> List[dict]:
    return [dict() for _ in range(cfg.num_hidden_layers)]


def merge_lora(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """W += B @ A for every LoRA pair; returns a dict without the lora keys (inference-time fold)."""
    out = {k: v for k, v in sd.items() if "_lora_" not in k}
    for k in sd:
        if k.endswith("_lora_a.weight"):
            base = k[: -len("_lora_a.weight")]
            out[base + ".weight"] = out[base + ".weight"] + sd[base + "_lora_b.weight"] @ sd[k]
    return out


# --------------------------------------------------------------------------------------------
# loss heads of BASELINE config #3 (modeling:221-237, 2238-2282, 2324-2351)
# --------------------------------------------------------------------------------------------
def siglip_loss(img: Tensor, txt: Tensor, logit_scale_exp: Tensor, logit_bias: Tensor,
                negative_only: bool = False) -> Tensor:
    """SigLipLoss._loss (modeling:221-237): -sum(logsigmoid(labels * logits)) / B, labels = 2I-1."""
    logits = logit_scale_exp * img @ txt.t() + logit_bias
    n = img.shape[0]
    labels = -torch.ones(n, txt.shape[0], dtype=img.dtype)
    if not negative_only:
        labels = labels + 2 * torch.eye(n, dtype=img.dtype)
    return -F.logsigmoid(labels * logits).sum() / n


def retrieval_loss(pooler: Tensor, text_features: Tensor, logit_scale: Tensor, logit_bias: Tensor,
                   other_rank_text: Optional[List[Tensor]] = None) -> Tensor:
    """TimesformerVideoRetrievalHead.forward (modeling:2324-2351), world_size 1 unless
    ``other_rank_text`` lists the other ranks' text features (negatives only, modeling:250-280)."""
    img = pooler[:, -1, :]
    img = img / img.norm(p=2, dim=-1, keepdim=True)
    txt = text_features / text_features.norm(p=2, dim=-1, keepdim=True)
    loss = siglip_loss(img, txt, logit_scale.exp(), logit_bias)
    for t in other_rank_text or []:
        t = t / t.norm(p=2, dim=-1, keepdim=True)
        loss = loss + siglip_loss(img, t, logit_scale.exp(), logit_bias, negative_only=True)
    return loss


def localization_loss(pooler: Tensor, label_emb: Tensor, labels: Tensor, logit_scale: Tensor,
                      logit_bias: Tensor) -> Tensor:
    """TimesformerUniversalLocalizationHead.forward, training branch (modeling:2238-2282), with one
    label-embedding table [L, D] shared by the batch (the synthetic config uses one dataset)."""
    B, T, D = pooler.shape
    img = pooler / pooler.norm(p=2, dim=-1, keepdim=True)
    total = pooler.new_zeros(())
    for i in range(B):
        logits = (img[i] @ label_emb.t()) * logit_scale.exp() + logit_bias      # [T, L]
        target = -torch.ones_like(logits)
        fg = labels[i] >= 0
        target[torch.arange(T)[fg], labels[i][fg]] = 1
        total = total + (-F.logsigmoid(target * logits).sum() / T)
    return total / B
