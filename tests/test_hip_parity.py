"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference's golden
outputs.  Tolerances (max-abs), stated once:

  * fp32-accurate mode (bf16x3 products): last_hidden_state <= 1e-3 (north_star bound), pooler <= 1e-3;
    measured values are ~1e-4 and the tests also assert a 5e-4 ceiling so regressions show early.
  * bf16 throughput mode: last_hidden_state <= 5e-2 (SURVEY §7's suggestion; measured 2.8-3.4e-2), pooler <= 3e-2 —
    SURVEY suggested 2e-2 for the pooler, but the measured value at SigLIP-base is 2.1-2.6e-2: the pooled vector sums
    196 value rows whose bf16 rounding errors add (the reference itself shows 0.08-0.11 / 0.012 in pure bf16,
    BASELINE.md §2, with weights whose scale differs from the seeded ones here); cosine >= 0.9995.
"""
import os

import numpy as np
import pytest
import torch

from oracle import streamformer_oracle as O
from streamformer_amd.configuration import siglip_base
from streamformer_amd.init_weights import make_state_dict
from tests.helpers import frames, load_npz, maxabs, small_cfg

pytestmark = pytest.mark.gpu

ACC_TOL = 1e-3
ACC_CEIL = 5e-4
# bf16 mode.  pooler_output: 3e-2 until round 3 (measured 2.7e-2).  Round 4 runs the pooling head's per-frame tail in bf16x3 in both modes: the
# head's own contribution fell from 1.5e-2 to 6.7e-3 and the SigLIP-base clip of bench.py measures 1.93e-2, of which 1.86e-2 is what the
# encoder's bf16 tokens carry through an EXACT head (tools/pool_err.py) — the operand floor, not the head.  SURVEY's 2e-2 holds for the
# SigLIP-base fixtures; the 128-wide two-layer fixture F7 (T = 32) measured 2.06e-2, hence 2.2e-2 in round 4.  Round 5: the head no longer
# projects the tokens to k / v at all (sf_pool_head.hip: scores and weighted sums on the fp32 tokens, bf16x3 / fp32 arithmetic in both
# modes), so what is left is exactly what the encoder's tokens carry: back to SURVEY's 2e-2 for every fixture.
BF16_LHS, BF16_POOL = 4e-2, 2e-2


@pytest.fixture(scope="module")
def sa():
    import streamformer_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return streamformer_amd


def build(sa, cfg, sd, mode, fuse=True):
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode, fuse_temporal_proj=fuse)
    m.load_state_dict(sd)
    return m.to("cuda").eval()


def cosine(a, b):
    a, b = a.double().flatten().cpu(), torch.as_tensor(np.asarray(b)).double().flatten()
    return float((a @ b) / (a.norm() * b.norm()))


# ------------------------------------------------------------------------------------------------
# single operators
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,D", [(1, 128), (37, 768), (1000, 768), (5, 1024), (3, 2048)])
def test_op_layernorm(sa, rows, D):
    nat = sa._native
    g = torch.Generator().manual_seed(rows * 7 + D)
    x = torch.randn(rows, D, generator=g) * 3 + 0.5
    w = torch.randn(D, generator=g) * 0.1 + 1
    b = torch.randn(D, generator=g) * 0.1
    want = torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-6)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    y = torch.empty_like(xd)
    nat.check(nat.lib.sf_op_layernorm(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), rows, D, 1e-6,
                                      nat.current_stream_handle(xd.device)))
    torch.cuda.synchronize()
    assert maxabs(y, want) <= 5e-6


def _linear(sa, x, w, b, resid, alpha, gelu, mode):
    nat = sa._native
    M, K = x.shape
    N = w.shape[0]
    xd, wd = x.cuda(), w.cuda()
    bd = b.cuda() if b is not None else None
    rd = resid.cuda() if resid is not None else None
    y = torch.empty(M, N, device="cuda")
    nb = nat.lib.sf_op_linear_workspace_bytes(M, N, K)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    nat.check(nat.lib.sf_op_linear(xd.data_ptr(), wd.data_ptr(), nat.ptr(bd), nat.ptr(rd), alpha, int(gelu), y.data_ptr(),
                                   M, N, K, mode, ws.data_ptr(), nb, nat.current_stream_handle(y.device)))
    torch.cuda.synchronize()
    return y.cpu()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 384, 128), (1, 768, 768), (333, 2304, 768), (129, 768, 3072)])
@pytest.mark.parametrize("mode", [0, 1])
def test_op_linear(sa, M, N, K, mode):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5       # asymmetric, non-square: catches transposes
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    tol = 2e-5 * K ** 0.5 if mode == 1 else 2e-2
    if mode == 0:   # compare against the same operand rounding so the bound is tight
        ref = x.bfloat16().double() @ w.bfloat16().double().t() + b.double()
        tol = 1e-4
    assert maxabs(_linear(sa, x, w, b, None, 1.0, False, mode), ref) <= tol
    assert maxabs(_linear(sa, x, w, b, r, 0.37, False, mode), r.double() + 0.37 * ref) <= tol
    got = _linear(sa, x, w, b, None, 1.0, True, mode)
    want = torch.nn.functional.gelu(ref)
    assert maxabs(got, want) <= (tol + (1e-5 if mode else 2e-2))   # bf16 output rounding in mode 0


@pytest.mark.parametrize("M,N,K,bm256", [(2300, 1024, 256, False), (2100, 1280, 128, True), (2077, 768, 256, 160), (2500, 768, 128, 192)])
@pytest.mark.parametrize("mode", [0, 1])
def test_op_linear_large_tiles(sa, M, N, K, bm256, mode, switches):
    """The 256x256 (and 224x256) persistent MFMA kernel incl. its bf16x3 variant (hi + lo planes, three products):
    ragged M, every epilogue the encoder uses on it."""
    if bm256 is True:
        switches("SF_G256_NO_BM224")
    elif bm256:
        switches("SF_G256_FORCE_BM", bm256)        # the short row tiles of the bf16x3 variant
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    if mode == 0:
        ref = x.bfloat16().double() @ w.bfloat16().double().t() + b.double()
        tol = 1e-4
    else:
        ref = x.double() @ w.double().t() + b.double()
        tol = 2e-5 * K ** 0.5
    assert maxabs(_linear(sa, x, w, b, None, 1.0, False, mode), ref) <= tol
    assert maxabs(_linear(sa, x, w, b, r, 0.37, False, mode), r.double() + 0.37 * ref) <= tol
    assert maxabs(_linear(sa, x, w, b, None, 1.0, True, mode), torch.nn.functional.gelu(ref)) <= (tol + (1e-5 if mode else 2e-2))


def test_op_linear_shape_sweep(sa):
    """Dispatch heuristics (skinny / panel / 128^2 / 256^2 with 160..256-row tiles, both modes) over ragged shapes:
    every (M, N, K) must give the same numbers whichever kernel it lands on."""
    rng = np.random.default_rng(7)
    shapes = [(25088, 768, 768), (25088, 2304, 768), (6272, 768, 3072), (3136, 3072, 768),
              (784, 2304, 768), (1568, 768, 3072), (600, 1000, 128), (2048, 3072, 768),      # several streams per call: the 64 x 64 tiles
              (4096, 1152, 1152), (2304, 3456, 1152), (3000, 1152, 4352)]      # N = 256 j + 128 (so400m widths): 256-column kernel + 128-column tail
    for _ in range(14):
        shapes.append((int(rng.integers(513, 9000)), int(rng.choice([768, 1024, 1536, 2304, 3072])), int(rng.choice([128, 256, 384, 768, 1152]))))
    for M, N, K in shapes:
        g = torch.Generator().manual_seed(M * 7 + N + K)
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) * K ** -0.5
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        xd, wd = x.cuda().double(), w.cuda().double()
        for mode in (0, 1):
            if mode == 0:
                ref = (x.bfloat16().cuda().double() @ w.bfloat16().cuda().double().t() + b.cuda().double()).cpu()
                tol = 1e-4
            else:
                ref = (xd @ wd.t() + b.cuda().double()).cpu()
                tol = 2e-5 * K ** 0.5
            err = maxabs(_linear(sa, x, w, b, r, 0.37, False, mode), r.double() + 0.37 * ref)
            assert err <= tol, (M, N, K, mode, err)
            if M <= 9000:
                err = maxabs(_linear(sa, x, w, b, None, 1.0, True, mode), torch.nn.functional.gelu(ref))
                assert err <= tol + (1e-5 if mode else 2e-2), (M, N, K, mode, err)


def _attention(sa, qkv, groups, L, heads, causal, temporal, ntok, mode, head_dim=64):
    nat = sa._native
    D = heads * head_dim
    q = qkv.cuda().contiguous()
    ctx = torch.empty(groups * L, D, device="cuda")
    nb = nat.lib.sf_op_attention_workspace_bytes(groups, L, heads, head_dim)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    nat.check(nat.lib.sf_op_attention(q.data_ptr(), ctx.data_ptr(), groups, L, heads, head_dim, int(causal), int(temporal), ntok,
                                      mode, ws.data_ptr(), nb, nat.current_stream_handle(q.device)))
    torch.cuda.synchronize()
    return ctx.cpu()


def _attn_ref(qkv, heads, mask=None):
    D = qkv.shape[-1] // 3
    q, k, v = qkv.double().split(D, dim=-1)
    return O._mha(q, k, v, heads, mask)[0]


@pytest.mark.parametrize("frames_,N,heads", [(3, 196, 12), (2, 9, 2), (1, 33, 1), (2, 224, 2), (4, 1, 2), (2, 16, 3),
                                              (2, 193, 2), (1, 208, 3), (2, 192, 1), (1, 209, 2),       # both edges of the 13-tile (compile-time count) instance
                                              (2, 225, 2), (1, 576, 3), (1, 1024, 1), (2, 400, 2)])   # > 224: streaming-key kernel
@pytest.mark.parametrize("mode", [0, 1])
def test_op_spatial_attention(sa, frames_, N, heads, mode, monkeypatch):
    _spatial_attention_case(sa, frames_, N, heads, mode)


@pytest.mark.parametrize("frames_,N,heads", [(2, 196, 2), (1, 37, 12), (3, 224, 1)])
def test_op_spatial_attention_accurate_fp32_inputs(sa, frames_, N, heads, switches):
    """The accurate mode's register-staged kernel (fp32 q / k / v: what streaming and output_attentions use) next to
    the DMA kernel on hi + lo planes that the plain call above takes."""
    switches("SF_DISABLE_SPATIAL_DMA_ACC")
    _spatial_attention_case(sa, frames_, N, heads, 1)


def _ntc_digest_child():
    """Child of test_compile_time_tile_count_is_bit_identical: digests of the spatial attention forward (op entry, bf16 mode) and
    backward at 196 and 200 tokens, printed for the parent (SF_DISABLE_SPATIAL_NTC is read once per process)."""
    import hashlib
    import streamformer_amd as sa_mod
    nat = sa_mod._native
    out = []
    for N, heads, frames_ in ((196, 12, 3), (200, 2, 2)):
        g = torch.Generator().manual_seed(N + heads)
        qkv = torch.randn(frames_, N, 3 * heads * 64, generator=g) * 1.5
        ctx = _attention(sa_mod, qkv.reshape(frames_ * N, -1), frames_, N, heads, False, False, 0, 0)
        out.append(hashlib.sha256(ctx.numpy().tobytes()).hexdigest()[:16])
        D = heads * 64
        dev = torch.device("cuda", 0)
        q16 = qkv.reshape(frames_ * N, -1).bfloat16().to(dev)
        o16 = ctx.bfloat16().to(dev)
        do16 = torch.randn(frames_ * N, D, generator=g).bfloat16().to(dev)
        dq = torch.zeros_like(q16)
        nat.check(nat.lib.sf_op_attention_bwd(q16.data_ptr(), o16.data_ptr(), do16.data_ptr(), dq.data_ptr(), 0, frames_, N, 1, heads, 0,
                                              nat.current_stream_handle(dev)))
        torch.cuda.synchronize()
        out.append(hashlib.sha256(dq.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16])
    print("NTC_DIGEST " + " ".join(out), flush=True)


def test_compile_time_tile_count_is_bit_identical(sa):
    """193 .. 208 tokens per frame take kernel instances whose tile / block count is a template constant (loops unrolled into one
    schedulable region); they must return the bits of the run-time-count kernels, forward and backward."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for off in (False, True):
        env = dict(os.environ)
        env.pop("SF_DISABLE_SPATIAL_NTC", None)
        if off:
            env["SF_DISABLE_SPATIAL_NTC"] = "1"
        r = subprocess.run([sys.executable, "-c", "import tests.test_hip_parity as t; t._ntc_digest_child()"], env=env, cwd=root,
                           capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("NTC_DIGEST")]
        assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
        digests.append(lines[-1])
    assert digests[0] == digests[1], digests


def _spatial_attention_case(sa, frames_, N, heads, mode, head_dim=64):
    g = torch.Generator().manual_seed(N * 13 + heads)
    qkv = torch.randn(frames_, N, 3 * heads * head_dim, generator=g) * 1.5
    qkv[..., 5] += 6.0                                   # a spiky column: exercises the max-subtraction
    src = qkv if mode == 1 else qkv.bfloat16().float()
    want = _attn_ref(src, heads).reshape(frames_ * N, -1)
    got = _attention(sa, qkv.reshape(frames_ * N, -1), frames_, N, heads, False, False, 0, mode, head_dim)
    assert maxabs(got, want) <= (2e-4 if mode == 1 else 3e-2)


@pytest.mark.parametrize("head_dim", [72, 32, 8, 96, 128])
@pytest.mark.parametrize("frames_,N,heads", [(2, 256, 2), (3, 9, 3), (1, 37, 16), (1, 729, 1)])
@pytest.mark.parametrize("mode", [0, 1])
def test_op_spatial_attention_any_head_dim(sa, frames_, N, heads, mode, head_dim):
    """Head widths other than 64 (configuration_streamformer.py:90-135 takes any hidden_size / heads; SigLIP-so400m = 72) run on
    sf_attention_generic.hip: fp32 products on the f32 matrix pipe in both modes, against the fp64 reference on the operands the mode
    sees (bf16-rounded q / k / v in bf16 mode; the output is then rounded to bf16 once more)."""
    _spatial_attention_case(sa, frames_, N, heads, mode, head_dim)


@pytest.mark.parametrize("head_dim", [72, 32, 128])
@pytest.mark.parametrize("B,L,Nt,heads,causal", [(2, 16, 9, 2, 1), (1, 5, 4, 3, 1), (1, 1, 3, 2, 1), (1, 64, 2, 2, 1), (2, 16, 5, 3, 0), (1, 33, 2, 1, 1)])
@pytest.mark.parametrize("mode", [0, 1])
def test_op_temporal_attention_any_head_dim(sa, B, L, Nt, heads, causal, mode, head_dim):
    _temporal_attention_case(sa, B, L, Nt, heads, causal, mode, head_dim)


@pytest.mark.parametrize("B,L,Nt,heads,causal", [(2, 16, 9, 2, 1), (1, 5, 4, 12, 1), (1, 1, 3, 2, 1), (1, 64, 2, 2, 1),
                                                  (2, 16, 5, 3, 0), (1, 33, 2, 1, 1), (1, 100, 1, 2, 1)])
@pytest.mark.parametrize("mode", [0, 1])
def test_op_temporal_attention(sa, B, L, Nt, heads, causal, mode):
    _temporal_attention_case(sa, B, L, Nt, heads, causal, mode)


@pytest.mark.parametrize("B,L,Nt,heads,causal", [(2, 16, 9, 2, 1), (1, 5, 4, 12, 1), (2, 16, 5, 3, 0)])
def test_op_temporal_attention_accurate_fp32_inputs(sa, B, L, Nt, heads, causal, switches):
    """Short sequences in the accurate mode on the register-staged kernel with fp32 q / k / v (what streaming keeps using);
    the plain call above takes the DMA kernel on hi + lo planes for L <= 16."""
    switches("SF_DISABLE_TEMPORAL_DMA_ACC")
    _temporal_attention_case(sa, B, L, Nt, heads, causal, 1)


def _temporal_attention_case(sa, B, L, Nt, heads, causal, mode, head_dim=64):
    g = torch.Generator().manual_seed(L * 17 + Nt)
    D = heads * head_dim
    qkv = torch.randn(B, L, Nt, 3 * D, generator=g) * 1.5          # the encoder's [B,T,N,3D] layout
    src = qkv if mode == 1 else qkv.bfloat16().float()
    seq = src.permute(0, 2, 1, 3).reshape(B * Nt, L, 3 * D)
    mask = torch.tril(torch.ones(L, L, dtype=torch.bool)) if causal else None
    want = _attn_ref(seq, heads, mask).reshape(B, Nt, L, D).permute(0, 2, 1, 3).reshape(B * L * Nt, D)
    got = _attention(sa, qkv.reshape(B * L * Nt, 3 * D), B * Nt, L, heads, causal, True, Nt, mode, head_dim)
    assert maxabs(got, want) <= (2e-4 if mode == 1 else 3e-2)


# ------------------------------------------------------------------------------------------------
# whole forward, small config (golden F1/F5/F7 + oracle)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("fuse", [True, False])
def test_forward_small_vs_golden(sa, golden_dir, mode, fuse):
    g = load_npz(os.path.join(golden_dir, "f1_small.npz"))
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=1)
    m = build(sa, cfg, sd, mode, fuse)
    for T in (1, 5, 16):
        x = frames(100 + T, (2, T, 3, 48, 48))
        out = m(x.cuda(), output_hidden_states=True)
        torch.cuda.synchronize()
        lt, pt = (ACC_CEIL, ACC_CEIL) if mode == "fp32" else (BF16_LHS, BF16_POOL)
        assert out.last_hidden_state.shape == (2, T, 9, 128) and out.pooler_output.shape == (2, T, 128)
        assert maxabs(out.last_hidden_state, g[f"T{T}_last_hidden_state"]) <= lt
        assert maxabs(out.pooler_output, g[f"T{T}_pooler_output"]) <= pt
        hs = torch.stack([h.cpu() for h in out.hidden_states])            # patch-major like the reference
        assert hs.shape == (3, 2, 9 * T, 128)
        assert maxabs(hs, g[f"T{T}_hidden_states"]) <= (ACC_CEIL if mode == "fp32" else 6e-2)      # pre-LayerNorm residual rows up to |7.4|: measured 3.9e-2 (tools/hs_err.py)
        tup = m(x.cuda(), return_dict=False)
        assert isinstance(tup, tuple) and torch.equal(tup[0], out.last_hidden_state)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_forward_shapes_f7(sa, golden_dir, mode):
    g = load_npz(os.path.join(golden_dir, "f7_shapes.npz"))
    cfg = small_cfg()
    m = build(sa, cfg, make_state_dict(cfg, seed=1), mode)
    lt, pt = (ACC_CEIL, ACC_CEIL) if mode == "fp32" else (BF16_LHS, BF16_POOL)
    for T in (8, 32):
        out = m(frames(200 + T, (1, T, 3, 48, 48)).cuda())
        assert maxabs(out.last_hidden_state, g[f"T{T}_last_hidden_state"]) <= lt
        assert maxabs(out.pooler_output, g[f"T{T}_pooler_output"]) <= pt
    out = m(frames(299, (1, 4, 3, 32, 64)).cuda())                        # resized position table
    assert maxabs(out.last_hidden_state, g["rect_last_hidden_state"]) <= lt
    assert maxabs(out.pooler_output, g["rect_pooler_output"]) <= pt
    cfg_bi = small_cfg(enable_causal_temporal=False)
    m2 = build(sa, cfg_bi, make_state_dict(cfg_bi, seed=2), mode)
    out = m2(frames(300, (2, 16, 3, 48, 48)).cuda())
    assert maxabs(out.last_hidden_state, g["bi_last_hidden_state"]) <= lt
    assert maxabs(out.pooler_output, g["bi_pooler_output"]) <= pt


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_forward_lora_f5(sa, golden_dir, mode):
    g = load_npz(os.path.join(golden_dir, "f5_lora.npz"))
    cfg = small_cfg(add_lora_spatial=True)
    m = build(sa, cfg, make_state_dict(cfg, seed=3), mode)
    out = m(frames(400, (2, 16, 3, 48, 48)).cuda())
    lt, pt = (ACC_CEIL, ACC_CEIL) if mode == "fp32" else (BF16_LHS, BF16_POOL)
    assert maxabs(out.last_hidden_state, g["last_hidden_state"]) <= lt
    assert maxabs(out.pooler_output, g["pooler_output"]) <= pt


def test_causal_property_hip(sa):
    cfg = small_cfg()
    m = build(sa, cfg, make_state_dict(cfg, seed=1), "bf16")
    x = frames(7, (1, 16, 3, 48, 48))
    x2 = x.clone()
    x2[:, 9:] += 1.0
    a = m(x.cuda()).last_hidden_state
    b = m(x2.cuda()).last_hidden_state
    assert torch.equal(a[:, :9], b[:, :9]), "frames < 9 must not see frames >= 9 (bit-exact)"
    assert not torch.equal(a[:, 9:], b[:, 9:])


@pytest.mark.gpu
def test_pool_head_lds_request_is_result_neutral(sa, switches):
    """The pooling head's probe / combine kernels ask for a CU's whole LDS so that no other workgroup shares their CU (DESIGN.md 4,
    "Device sharing"); SF_POOL_SHARE_CU=1 gives them their exact sizes back.  Same kernels, same arithmetic: bit-identical outputs
    (the oracle comparison of this configuration is test_forward_small_vs_golden's)."""
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=4)
    m = build(sa, cfg, sd, "bf16")
    xc = frames(91, (2, 8, 3, 48, 48))
    whole = m(xc.cuda())
    switches("SF_POOL_SHARE_CU")
    exact = m(xc.cuda())
    assert torch.equal(whole.pooler_output, exact.pooler_output) and torch.equal(whole.last_hidden_state, exact.last_hidden_state)
    assert bool(torch.isfinite(whole.pooler_output).all()) and float(whole.pooler_output.abs().max()) > 0


# ------------------------------------------------------------------------------------------------
# streaming with the KV-cache (golden F4)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("tag,nf", [("nf16", 16), ("nf64", 64)])
def test_streaming_f4(sa, golden_dir, mode, tag, nf):
    g = load_npz(os.path.join(golden_dir, "f4_streaming.npz"))
    cfg = small_cfg(num_frames=nf, add_lora_spatial=True)
    m = build(sa, cfg, make_state_dict(cfg, seed=4), mode)
    x = frames(500 + nf, (1, nf, 3, 48, 48)).cuda()
    want = g[f"{tag}_last_hidden_state"]
    lt = ACC_CEIL if mode == "fp32" else BF16_LHS
    full = m(x).last_hidden_state
    assert maxabs(full, want) <= lt
    for chunks in ([nf], [nf // 2, nf // 2], [1] * nf, [3, 1, nf - 4]):
        cache, outs, pos = None, [], 0
        for c in chunks:
            o = m(x[:, pos:pos + c], use_cache=True, past_key_values=cache)
            cache = o.past_key_values
            assert cache.get_seq_length() == pos + c
            outs.append(o.last_hidden_state)
            pos += c
        got = torch.cat(outs, 1)
        assert maxabs(got, want) <= lt
        # streamed == full clip on the same device path, to fp32 re-ordering noise
        # streamed vs full clip on the device: fp32 re-ordering noise; in bf16 mode the small-M calls fold the LayerNorms into
        # the GEMMs (statistics of the bf16-rounded rows) while a 64-frame clip runs the separate LayerNorm: two bf16 roundings apart
        assert maxabs(got, full) <= (1e-4 if mode == "fp32" else 4e-2)
    with pytest.raises(Exception):                        # past the time-embedding rows / cache capacity
        m(x[:, :1], use_cache=True, past_key_values=cache)
    cache.reset()
    o = m(x[:, :2], use_cache=True, past_key_values=cache)
    assert maxabs(o.last_hidden_state, want[:, :2]) <= lt


class _TowerCfg:
    """stand-in for LLaVA's model config object (vqa_enc:1494-1500 reads these with getattr)"""
    streaming_mode = True
    context_length = 6


@pytest.mark.parametrize("mode,tol", [("fp32", ACC_CEIL), ("bf16", BF16_LHS)])
def test_vision_tower_streams_against_the_oracle(sa, tmp_path, mode, tol):
    """TimesformerVisionTower (vqa_enc:1462-1598) built the reference's way — checkpoint directory + config object — fed one
    frame per call; every returned window is compared with the ORACLE's full-clip forward (causal => identical)."""
    cfg = small_cfg(num_frames=16)
    sd = make_state_dict(cfg, seed=1)
    build(sa, cfg, sd, mode).save_pretrained(str(tmp_path))
    tower = sa.TimesformerVisionTower(str(tmp_path), _TowerCfg(), delay_load=True, compute_dtype=mode)
    assert tower.is_loaded and tower.streaming_mode and tower.context_length == 6
    assert tower.device.type == "cuda" and tower.dtype == torch.float32
    assert (tower.hidden_size, tower.num_patches, tower.num_patches_per_side, tower.image_size) == (128, 9, 3, 48)
    assert tower.dummy_feature.shape == (1, 128) and tower.dummy_feature.device.type == "cuda"
    assert all(not p.requires_grad for p in tower.parameters())                    # vqa_enc:1524
    assert tower.image_processor.size == (48, 48)
    x = frames(11, (1, 10, 3, 48, 48))
    want = O.forward(sd, cfg, x)["last_hidden_state"]
    for t in range(10):
        feats = tower(x[:, t:t + 1].cuda())
        lo = max(0, t + 1 - 6)
        assert feats.shape == (1, t + 1 - lo, 9, 128)
        assert maxabs(feats, want[:, lo:t + 1]) <= tol
    # a new stream after clear_cache (vqa_enc:1528-1530): K/V buffers recycled, results as from scratch
    first = tower.past_key_values
    tower.clear_cache()
    assert tower.past_key_values is None and tower.hidden_states is None
    y = frames(12, (1, 3, 3, 48, 48))
    got = tower(y.cuda())
    assert tower.past_key_values is first
    assert maxabs(got, O.forward(sd, cfg, y)["last_hidden_state"]) <= tol
    # the non-streaming branches: a clip tensor -> last_hidden_state, a list of clips -> last encoder hidden state
    plain = sa.TimesformerVisionTower(tower.vision_tower, streaming_mode=False)
    ow = O.forward(sd, cfg, y, output_hidden_states=True)
    assert maxabs(plain(y.cuda()), ow["last_hidden_state"]) <= tol
    lst = plain([y[0].cuda()])
    # bf16: relative to the tensor's abs-max (un-normalised residual values reach ~30; an absolute 0.25 could hide a regression — VERDICT r5)
    hs_ref = ow["hidden_states"][-1]
    assert len(lst) == 1 and maxabs(lst[0], hs_ref) <= (tol if mode == "fp32" else 1e-2 * float(np.abs(np.asarray(hs_ref)).max()))


@pytest.mark.parametrize("mode,tol", [("fp32", ACC_CEIL), ("bf16", None)])
def test_streaming_with_output_hidden_states(sa, mode, tol):
    """The tower's literal call form (vqa_enc:1536): output_hidden_states=True with use_cache=True and cache_position=None.
    hidden_states of the new frames (patch-major) against the oracle's for the same frames of the full clip.
    bf16 tolerance: un-normalised residual stream values reach ~30 and the bf16 operand error scales with them, so the bound is
    RELATIVE: 1e-2 of the layer tensor's abs-max (an absolute 0.25 could hide a regression)."""
    cfg = small_cfg(num_frames=16)
    sd = make_state_dict(cfg, seed=4)
    m = build(sa, cfg, sd, mode)
    x = frames(5, (2, 8, 3, 48, 48))
    want = O.forward(sd, cfg, x, output_hidden_states=True)
    N, D = 9, 128
    cache, pos = None, 0
    for c in (3, 1, 3):
        o = m(x[:, pos:pos + c].cuda(), output_hidden_states=True, use_cache=True, past_key_values=cache, cache_position=None)
        cache = o.past_key_values
        assert len(o.hidden_states) == cfg.num_hidden_layers + 1
        for li, h in enumerate(o.hidden_states):
            assert h.shape == (2, N * c, D)
            w = want["hidden_states"][li].reshape(2, N, 8, D)[:, :, pos:pos + c].reshape(2, N * c, D)     # token = n*T + t
            lim = tol if tol is not None else 1e-2 * float(torch.as_tensor(w).abs().max())
            assert maxabs(h, w) <= lim, (li, pos, lim)
        assert maxabs(o.last_hidden_state, want["last_hidden_state"][:, pos:pos + c]) <= (ACC_CEIL if mode == "fp32" else BF16_LHS)
        pos += c
    tup = m(x[:, pos:pos + 1].cuda(), output_hidden_states=True, use_cache=True, past_key_values=cache, return_dict=False)
    assert len(tup) == 3 and tup[2] is cache and len(tup[1]) == cfg.num_hidden_layers + 1


def test_module_protocol_on_the_gpu(sa):
    """modeling:1362 holds the encoder as `self.timesformer = TimesformerMultiTaskingModelSigLIP(config)`; vqa_enc:1574-1581 read
    device / dtype off its parameters.  A parent module moves it, saves it and reloads it through the plain nn.Module calls."""
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=6)

    class Wrapper(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.timesformer = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="fp32")
            self.logit_scale = torch.nn.Parameter(torch.tensor(2.3))

    w = Wrapper()
    assert next(w.timesformer.parameters()).device.type == "cpu"
    w.load_state_dict({**{"timesformer." + k: v for k, v in sd.items()}, "logit_scale": torch.tensor(1.0)})
    w.to("cuda").eval()
    p = next(w.timesformer.parameters())
    assert p.device.type == "cuda" and p.dtype == torch.float32 and w.timesformer.device.type == "cuda"
    x = frames(3, (1, 4, 3, 48, 48))
    want = O.forward(sd, cfg, x)
    assert maxabs(w.timesformer(x.cuda()).last_hidden_state, want["last_hidden_state"]) <= ACC_CEIL
    saved = w.state_dict()
    assert set(saved) == {"logit_scale"} | {"timesformer." + k for k in sd}
    assert all(torch.equal(saved["timesformer." + k].cpu(), v.float()) for k, v in sd.items())
    # in-place weight change through the parameter (what an optimizer does) is picked up by the next forward
    sd2 = make_state_dict(cfg, seed=7)
    with torch.no_grad():
        for k, prm in w.timesformer.named_parameters():
            prm.copy_(sd2[k])
    assert maxabs(w.timesformer(x.cuda()).last_hidden_state, O.forward(sd2, cfg, x)["last_hidden_state"]) <= ACC_CEIL
    # a half-precision module (vqa_enc:1536 casts the images to tower.dtype) answers in its dtype
    w.to(torch.bfloat16)
    assert w.timesformer.dtype == torch.bfloat16
    o = w.timesformer(x.cuda().bfloat16())
    assert o.last_hidden_state.dtype == torch.bfloat16
    sdb = {k: (v.bfloat16().float() if v.is_floating_point() else v) for k, v in sd2.items()}
    wantb = O.forward(sdb, cfg, x.bfloat16().float())["last_hidden_state"]
    assert maxabs(o.last_hidden_state.float(), wantb) <= 4e-2          # bf16 output rounding of values up to ~6


@pytest.mark.parametrize("mode,tol", [("fp32", ACC_CEIL), ("bf16", BF16_LHS)])
@pytest.mark.parametrize("num_frames,cap,streams", [(8, 6, 1), (4, 6, 2), (16, 16, 1)])
def test_sliding_window_cache_outlives_num_frames(sa, mode, tol, num_frames, cap, streams):
    """VERDICT r2 missing #5 / SURVEY 8 f-2 "bounded-memory policy": policy="slide" — past `max_frames` every new frame
    replaces the oldest cached one, attends to the last `max_frames` frames and (past the time-embedding table) reuses the
    table's last row — against the oracle's restatement of the same policy (oracle `window=`), 2.5 windows deep, also from
    inside the position-free graph; the default policy still stops where the reference raises (vqa_enc:343-348)."""
    cfg = small_cfg(num_frames=num_frames)
    sd = make_state_dict(cfg, seed=4)
    m = build(sa, cfg, sd, mode)
    total = int(2.5 * cap) + 1
    x = frames(9, (streams, total, 3, 48, 48))
    ocache = O.new_cache(cfg)
    cache = m.new_cache(streams, cap, policy="slide")
    t = 0
    first = min(3, cap, num_frames)                     # a multi-frame call while the window still has room, then single frames
    chunks = [first] + [1] * (total - first)
    for c in chunks:
        want = O.forward(sd, cfg, x[:, t:t + c], cache=ocache, window=cap)
        got = m(x[:, t:t + c].cuda(), use_cache=True, past_key_values=cache)
        assert maxabs(got.last_hidden_state, want["last_hidden_state"]) <= tol, (t, c)
        assert maxabs(got.pooler_output, want["pooler_output"]) <= tol, (t, c)
        t += c
        assert cache.frames_seen == t and cache.get_seq_length() == min(t, cap)
    with pytest.raises(Exception):                       # a full window advances one frame per call
        m(x[:, :2].cuda(), use_cache=True, past_key_values=cache)
    stop = m.new_cache(streams, cap)                     # default policy: the reference's hard stop
    lim = min(cap, num_frames)
    m(x[:, :lim].cuda(), use_cache=True, past_key_values=stop)
    with pytest.raises(Exception):
        m(x[:, lim:lim + 1].cuda(), use_cache=True, past_key_values=stop)
    tower = sa.TimesformerVisionTower(m, streaming_mode=True, context_length=3, max_frames=cap, cache_policy="slide")
    for i in range(total):
        out = tower(x[:1, i:i + 1] if streams == 1 else x[:, i:i + 1])
    assert out.shape[1] == 3 and torch.isfinite(out).all()


@pytest.mark.parametrize("mode,tol", [("fp32", ACC_CEIL), ("bf16", BF16_LHS)])
@pytest.mark.parametrize("cap", [72, 132])
def test_long_cache_single_query_temporal_attention(sa, mode, tol, cap):
    """The single-query temporal kernels past 64 cached frames: whole-line loads with two key passes (65..128 keys, bf16
    rows), one key per lane with four passes (129..256 keys, and every fp32 row) — one frame per call against the oracle, the last
    frames of a `cap`-frame stream (every key pass of the wave holds live keys there), eagerly and from the position-free graph."""
    cfg = small_cfg(num_frames=cap, num_hidden_layers=1)
    sd = make_state_dict(cfg, seed=6)
    m = build(sa, cfg, sd, mode)
    x = frames(12, (1, cap, 3, 48, 48))
    ocache = O.new_cache(cfg)
    cache = m.new_cache(1, cap)
    for t in range(cap):
        want = O.forward(sd, cfg, x[:, t:t + 1], cache=ocache)
        got = m(x[:, t:t + 1].cuda(), use_cache=True, past_key_values=cache)
        if t < 3 or t % 16 == 15 or t >= cap - 3:
            assert maxabs(got.last_hidden_state, want["last_hidden_state"]) <= tol, t
            assert maxabs(got.pooler_output, want["pooler_output"]) <= tol, t
    assert cache.get_seq_length() == cap


def test_stale_cache_is_refused(sa):
    """a cache created before the weights were re-packed must not be written with another element size."""
    nat = sa._native
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=1)
    m = build(sa, cfg, sd, "bf16")
    x = frames(2, (1, 3, 3, 48, 48)).cuda()
    o = m(x[:, :1], use_cache=True)
    cache = o.past_key_values
    tower = sa.TimesformerVisionTower(m, streaming_mode=True, context_length=4)
    tower(x[:, :1])
    m.set_compute_dtype("fp32")
    full = m(x).last_hidden_state                          # re-packs: every live cache is invalidated
    assert not cache.valid
    with pytest.raises(RuntimeError, match="start a new cache"):
        m(x[:, 1:2], use_cache=True, past_key_values=cache)
    got = tower(x[:, :2])                                  # the tower notices and starts a fresh stream
    assert maxabs(got, full[:, :2]) <= 1e-4
    # the same guard inside the C ABI: re-finalising a handle under a live cache
    c2 = m.new_cache(1, 4)
    nat.check(nat.lib.sf_finalize_weights(m._handle, nat.SF_COMPUTE_BF16, 1, 1))
    n = nat.C.c_size_t()
    assert nat.lib.sf_stream_workspace_bytes(m._handle, c2._h, 1, nat.C.byref(n)) == nat.SF_ERR_STATE
    lhs = torch.empty(1, 1, 9, 128, device="cuda")
    ws = torch.empty(1 << 24, dtype=torch.uint8, device="cuda")
    rc = nat.lib.sf_forward_stream(m._handle, c2._h, x.data_ptr(), nat.SF_F32, 1, lhs.data_ptr(), None, None, None, ws.data_ptr(),
                                   ws.numel(), nat.current_stream_handle(lhs.device))
    assert rc == nat.SF_ERR_STATE and b"earlier weight packing" in nat.lib.sf_last_error()
    m.refresh_weights()


def test_forward_features_pooling_methods(sa):
    """modeling:1525-1536: "mean" is the default, "no_pooling" returns every frame, any other string the last frame."""
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=9)
    m = build(sa, cfg, sd, "fp32")
    x = frames(4, (2, 5, 3, 48, 48))
    pooled = O.forward(sd, cfg, x)["pooler_output"]
    assert maxabs(m.forward_features(x.cuda()), pooled.mean(1)) <= ACC_CEIL
    assert maxabs(m.forward_features(x.cuda(), pooling_method="no_pooling"), pooled) <= ACC_CEIL
    assert maxabs(m.forward_features(x.cuda(), pooling_method="last"), pooled[:, -1]) <= ACC_CEIL
    assert maxabs(m.forward_features(x.cuda(), "anything else"), pooled[:, -1]) <= ACC_CEIL


# ------------------------------------------------------------------------------------------------
# SigLIP-base (golden F2) and full BASELINE size properties
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def base_models(sa):
    cfg = siglip_base()
    sd = make_state_dict(cfg, seed=0)
    return {mode: build(sa, cfg, sd, mode) for mode in ("fp32", "bf16")}


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["randn", "clamped"])
def test_base_vs_golden_f2(sa, golden_dir, base_models, mode, tag):
    g = load_npz(os.path.join(golden_dir, "f2_base.npz"))
    torch.manual_seed(0)
    x = torch.randn(1, 16, 3, 224, 224)
    if tag == "clamped":
        x = x.clamp(-1, 1)
    out = base_models[mode](x.cuda())
    lhs = out.last_hidden_state
    lt, pt = (ACC_TOL, ACC_TOL) if mode == "fp32" else (BF16_LHS, BF16_POOL)
    d_pool = maxabs(out.pooler_output, g[f"{tag}_pooler_output"])
    d_lhs = maxabs(lhs[0][[0, 7, 15]][:, [0, 97, 195]], g[f"{tag}_lhs_slices"])
    print(f"[{mode}/{tag}] max-abs lhs-slices {d_lhs:.3e} pooler {d_pool:.3e}")
    assert d_pool <= pt and d_lhs <= lt
    norms = lhs[0].double().flatten(1).norm(dim=1).cpu()
    assert maxabs(norms, g[f"{tag}_lhs_frame_norms"]) <= (2e-2 if mode == "fp32" else 1.0)
    assert abs(float(lhs.double().sum()) - float(g[f"{tag}_lhs_checksum"])) <= (1.0 if mode == "fp32" else 400.0)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_base_full_tensor_vs_oracle(sa, base_models, mode):
    """Full last_hidden_state of one 16x224^2 clip against the CPU oracle (the number bench.py reports)."""
    cfg = siglip_base()
    sd = make_state_dict(cfg, seed=0)
    torch.manual_seed(0)
    x = torch.randn(1, 16, 3, 224, 224)
    want = O.forward(sd, cfg, x)
    out = base_models[mode](x.cuda())
    d_lhs, d_pool = maxabs(out.last_hidden_state, want["last_hidden_state"]), maxabs(out.pooler_output, want["pooler_output"])
    print(f"[{mode}] full-tensor max-abs lhs {d_lhs:.3e} pooler {d_pool:.3e} cosine {cosine(out.last_hidden_state, want['last_hidden_state']):.6f}")
    if mode == "fp32":
        assert d_lhs <= ACC_TOL and d_pool <= ACC_TOL
    else:
        assert d_lhs <= BF16_LHS and d_pool <= BF16_POOL
        assert cosine(out.last_hidden_state, want["last_hidden_state"]) >= 0.9995


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_baseline_batch8_properties(sa, base_models, mode):
    """BASELINE config #2 size (8 x 16 x 224^2).  At this size the bf16 mode runs the LayerNorm-folded
    schedule (panel + 256^2 GEMMs), so it is checked against the oracle directly, plus the properties
    the domain offers: a clip's output does not depend on the other clips of the batch (bit-exact),
    equals the clip run alone up to kernel-schedule rounding, and obeys causality at full size."""
    m = base_models[mode]
    cfg = siglip_base()
    sd = make_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(123)
    xc = torch.randn(8, 16, 3, 224, 224, generator=g)
    x = xc.cuda()
    out = m(x)
    assert out.last_hidden_state.shape == (8, 16, 196, 768) and out.pooler_output.shape == (8, 16, 768)
    assert torch.isfinite(out.last_hidden_state).all() and torch.isfinite(out.pooler_output).all()
    lt, pt = (ACC_TOL, ACC_TOL) if mode == "fp32" else (BF16_LHS, BF16_POOL)
    for i in (0, 5):
        want = O.forward(sd, cfg, xc[i:i + 1])
        d1 = maxabs(out.last_hidden_state[i], want["last_hidden_state"][0])
        d2 = maxabs(out.pooler_output[i], want["pooler_output"][0])
        print(f"[{mode}] B=8 clip {i}: max-abs lhs {d1:.3e} pooler {d2:.3e}")
        assert d1 <= lt and d2 <= pt
        solo = m(x[i:i + 1])
        assert maxabs(solo.last_hidden_state[0], out.last_hidden_state[i]) <= (1e-4 if mode == "fp32" else 4e-2)
    # other clips' content does not leak into clip 3 (same batch shape => same schedule => bit-exact)
    x3 = x.clone()
    x3[:3] = 0.5 * x3[:3] + 1.0
    x3[4:] = -x3[4:]
    out3 = m(x3)
    assert torch.equal(out3.last_hidden_state[3], out.last_hidden_state[3])
    assert torch.equal(out3.pooler_output[3], out.pooler_output[3])
    for _ in range(10):      # race screen for the count-placed LDS hand-offs: every repeat bit-identical
        again = m(x)
        assert torch.equal(again.last_hidden_state, out.last_hidden_state)
        assert torch.equal(again.pooler_output, out.pooler_output)
    x2 = x.clone()
    x2[:, 12:] = -x2[:, 12:]
    out2 = m(x2)
    assert torch.equal(out2.last_hidden_state[:, :12], out.last_hidden_state[:, :12])
    assert not torch.equal(out2.last_hidden_state[:, 12:], out.last_hidden_state[:, 12:])


# ------------------------------------------------------------------------------------------------
# loss heads (golden F6)
# ------------------------------------------------------------------------------------------------
def test_heads_f6(sa, golden_dir):
    g = load_npz(os.path.join(golden_dir, "f6_heads.npz"))
    pooler = torch.tensor(g["pooler"]).cuda()
    r = sa.heads.RetrievalHead()
    loss, gp, gs = r.loss(pooler, torch.tensor(g["text"]).cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["retrieval_loss"])) <= 2e-5
    assert maxabs(gp, g["retrieval_grad"]) <= 1e-6
    l = sa.heads.LocalizationHead(torch.tensor(g["label_emb"]))
    loss, gp, gs = l.loss(pooler, torch.tensor(g["labels"]).cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["localization_loss"])) <= 2e-5
    assert maxabs(gp, g["localization_grad"]) <= 1e-6
    assert abs(float(gs[0]) - float(g["localization_logit_scale_grad"])) <= 2e-5
    assert abs(float(gs[1]) - float(g["localization_logit_bias_grad"])) <= 2e-5


def test_retrieval_multirank_negatives(sa, golden_dir):
    """world_size 2: local block + the other rank's captions as negatives only (modeling:250-280)."""
    g = load_npz(os.path.join(golden_dir, "f6_heads.npz"))
    pooler, txt = torch.tensor(g["pooler"]), torch.tensor(g["text"])
    other = torch.flip(txt, dims=[0]) * 1.3 + 0.1
    ls, lb = torch.log(torch.tensor(10.0)), torch.tensor(-2.0)
    for rank in (0, 1):
        allt = torch.cat([txt, other] if rank == 0 else [other, txt], 0)
        p = pooler.clone().requires_grad_(True)
        want = O.retrieval_loss(p, txt, ls, lb, other_rank_text=[other])
        want.backward()
        loss, gp, _ = sa.heads.RetrievalHead().loss(pooler.cuda(), allt.cuda(), rank=rank)
        assert abs(float(loss) - float(want)) <= 2e-5 and maxabs(gp, p.grad) <= 1e-6


@pytest.mark.gpu
def test_uint8_frames_fused_normalisation(golden_dir):
    """uint8 frames through the byte path of the patch kernel (rescale + normalize fused) == the same frames
    normalised by the image processor on the host (fixture F9 pins that against the reference)."""
    import streamformer_amd as sa
    f9 = load_npz(os.path.join(golden_dir, "f9_processor.npz"))
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=3)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="fp32")
    m.load_state_dict(sd)
    m.to("cuda")
    clip = m.image_processor.preprocess(list(f9["frames_big"]) + [f9["frame_small"]])["pixel_values"]     # uint8 [4,3,48,48]
    assert clip.dtype == torch.uint8
    x_u8 = clip[None].cuda()
    x_f = torch.from_numpy(f9["pixel_values"])[None].cuda()
    a, b = m(x_u8), m(x_f)
    assert maxabs(a.last_hidden_state, b.last_hidden_state) <= 1e-4 and maxabs(a.pooler_output, b.pooler_output) <= 1e-4
    want = O.forward(sd, cfg, x_f.cpu())
    assert maxabs(a.last_hidden_state, want["last_hidden_state"]) <= ACC_TOL
    # a non-default normalisation is honoured by the kernel
    m.image_processor.image_mean, m.image_processor.image_std = (0.4, 0.5, 0.6), (0.2, 0.25, 0.3)   # picked up by the next call
    c = m(x_u8)
    d = m(m.image_processor.normalize(clip)[None].cuda())
    assert maxabs(c.last_hidden_state, d.last_hidden_state) <= 1e-4
    # streaming entry takes bytes too
    cache = m.new_cache(1, 16)
    s = m(x_u8[:, :2], use_cache=True, past_key_values=cache)
    assert maxabs(s.last_hidden_state, c.last_hidden_state[:, :2]) <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol", [("fp32", ACC_TOL), ("bf16", BF16_LHS)])
def test_forward_more_than_224_patches(mode, tol):
    """384 x 384 frames = 576 patches per frame: spatial attention streams the keys (online softmax);
    full clip and streamed frame by frame, against the oracle."""
    import streamformer_amd as sa
    cfg = small_cfg(image_size=384, num_frames=4)
    sd = make_state_dict(cfg, seed=21)
    x = frames(21, (1, 4, 3, 384, 384))
    want = O.forward(sd, cfg, x)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda")
    out = m(x.cuda())
    assert maxabs(out.last_hidden_state, want["last_hidden_state"]) <= tol
    assert maxabs(out.pooler_output, want["pooler_output"]) <= tol
    cache = m.new_cache(1, 4)
    outs = [m(x[:, t:t + 1].cuda(), use_cache=True, past_key_values=cache).last_hidden_state for t in range(4)]
    assert maxabs(torch.cat(outs, 1), want["last_hidden_state"]) <= tol
    # output_attentions above 224 patches (round 4): the streamed-keys kernel's second sweep, against the oracle's probabilities
    col = {}
    O.forward(sd, cfg, x[:, :2], collect=col)
    oa = m(x[:, :2].cuda(), output_attentions=True)
    assert len(oa.attentions) == cfg.num_hidden_layers and tuple(oa.attentions[0].shape) == (2, cfg.num_attention_heads, 576, 576)
    for li in range(cfg.num_hidden_layers):
        assert maxabs(oa.attentions[li], col["attentions"][li]) <= (2e-5 if mode == "fp32" else 2e-2)
        assert float((oa.attentions[li].sum(-1) - 1).abs().max()) < 1e-4
    assert maxabs(oa.last_hidden_state, want["last_hidden_state"][:, :2]) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_output_attentions(golden_dir, mode, tol):
    """output_attentions=True (modeling:703-716): per-layer spatial probabilities [B*T, heads, N, N] vs the
    reference's own (fixture F10); same hidden states as without the flag; tuple form; SigLIP-base shape."""
    import streamformer_amd as sa
    f = load_npz(os.path.join(golden_dir, "f10_attentions.npz"))
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=10)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda")
    x = frames(10, (2, 5, 3, 48, 48)).cuda()
    out = m(x, output_attentions=True)
    assert len(out.attentions) == cfg.num_hidden_layers and tuple(out.attentions[0].shape) == (10, 2, 9, 9)
    got = torch.stack(list(out.attentions)).cpu()
    assert maxabs(got, f["attentions"]) <= tol
    assert float((got.sum(-1) - 1).abs().max()) < 1e-5
    # the accurate mode materialises probabilities with the fp32-input kernel and otherwise runs the DMA kernel on hi + lo
    # planes: same numbers to rounding, not to the bit
    plain = m(x).last_hidden_state
    assert torch.equal(out.last_hidden_state, plain) if mode == "bf16" else maxabs(out.last_hidden_state, plain) <= 2e-5
    tup = m(x, output_attentions=True, output_hidden_states=True, return_dict=False)
    assert len(tup) == 3 and len(tup[2]) == cfg.num_hidden_layers
    # output_attentions while streaming (timesformer_encoder.py:494, 557, 720-754): the new frames' probabilities, call by call,
    # are the full clip's rows for those frames (causal temporal attention: a frame never sees later ones)
    cache, pos = None, 0
    for csz in (2, 1, 2):
        o = m(x[:, pos:pos + csz], output_attentions=True, use_cache=True, past_key_values=cache)
        cache = o.past_key_values
        assert len(o.attentions) == cfg.num_hidden_layers and tuple(o.attentions[0].shape) == (2 * csz, 2, 9, 9)
        want = got.reshape(cfg.num_hidden_layers, 2, 5, 2, 9, 9)[:, :, pos:pos + csz].reshape(cfg.num_hidden_layers, 2 * csz, 2, 9, 9)
        assert maxabs(torch.stack(list(o.attentions)).cpu(), want) <= (5e-5 if mode == "fp32" else tol)
        assert maxabs(o.last_hidden_state, out.last_hidden_state[:, pos:pos + csz]) <= (5e-5 if mode == "fp32" else BF16_LHS)
        pos += csz
    tup = m(x[:, :1], output_attentions=True, use_cache=True, return_dict=False)
    assert len(tup) == 3 and len(tup[1]) == cfg.num_hidden_layers          # (lhs, attentions, cache)
    if mode == "bf16":      # N = 196, 12 heads: rows sum to one, softmax of what the kernel's own context used
        big = siglip_base(num_hidden_layers=1)
        mb = sa.TimesformerMultiTaskingModelSigLIP(big, compute_dtype="bf16")
        mb.load_state_dict(make_state_dict(big, seed=2))
        mb.to("cuda")
        ob = mb(frames(3, (1, 2, 3, 224, 224)).cuda(), output_attentions=True)
        a = ob.attentions[0]
        assert tuple(a.shape) == (2, 12, 196, 196) and float((a.sum(-1) - 1).abs().max()) < 1e-5 and float(a.min()) >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol", [("fp32", 2e-4), ("bf16", 5e-2)])
def test_submodule_calls_compose_to_the_forward(mode, tol):
    """embeddings -> encoder.layer[i] one by one -> post_layernorm -> head, with the reference's patch-major
    (B, N*T, D) tensors between the calls (adapter / classifier usage), equals model.forward and the oracle."""
    import streamformer_amd as sa
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=12)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda")
    x = frames(12, (2, 6, 3, 48, 48))
    collect = {}
    want = O.forward(sd, cfg, x, output_hidden_states=True, collect=collect)
    T = 6
    h = m.embeddings(x.cuda())
    assert tuple(h.shape) == (2, 9 * T, 128)
    assert maxabs(h, want["hidden_states"][0]) <= tol
    for i, blk in enumerate(m.encoder.layer):
        h = blk(h, T, output_attentions=False)[0]                       # the adapter's call (adapter:424-425)
        assert maxabs(h, want["hidden_states"][i + 1]) <= tol
    enc = m.encoder(m.embeddings(x.cuda()), num_frames=T, output_hidden_states=True, output_attentions=True)
    assert maxabs(enc.last_hidden_state, want["hidden_states"][-1]) <= tol and len(enc.hidden_states) == 3
    assert maxabs(torch.stack(list(enc.attentions)), torch.stack(collect["attentions"])) <= (2e-5 if mode == "fp32" else 2e-2)
    seq = m.post_layernorm(h)                                            # classifier: AR/...:130-134
    tok = seq.reshape(2, 9, T, 128).permute(0, 2, 1, 3).reshape(2 * T, 9, 128)
    pooled = m.head(tok).reshape(2, T, 128)
    assert maxabs(tok.reshape(2, T, 9, 128), want["last_hidden_state"]) <= tol
    assert maxabs(pooled, want["pooler_output"]) <= tol
    full = m(x.cuda())
    assert maxabs(full.pooler_output, pooled) <= (1e-4 if mode == "fp32" else 3e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_streaming_is_bit_reproducible_at_base_size(mode):
    """Race screen for the small-M kernels (skinny / K-parallel GEMM, query-split spatial attention, temporal
    attention over the cache): the same 24-frame stream twice, SigLIP-base shape, bit-identical outputs; and the
    streamed frames agree with the full-clip forward (different kernels serve the two)."""
    import streamformer_amd as sa
    cfg = siglip_base(num_hidden_layers=4, num_frames=32)
    sd = make_state_dict(cfg, seed=17)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda")
    x = frames(17, (1, 24, 3, 224, 224)).cuda()
    runs = []
    for rep in range(2):
        cache = m.new_cache(1, 32)
        outs = [m(x[:, t:t + 1], use_cache=True, past_key_values=cache) for t in range(24)]
        runs.append((torch.cat([o.last_hidden_state for o in outs], 1), torch.cat([o.pooler_output for o in outs], 1)))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    full = m(x)
    tol = 2e-4 if mode == "fp32" else BF16_LHS
    assert maxabs(runs[0][0], full.last_hidden_state) <= tol and maxabs(runs[0][1], full.pooler_output) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol_l,tol_p", [("fp32", ACC_TOL, ACC_TOL), ("bf16", BF16_LHS, BF16_POOL)])
def test_streaming_config5_full_size_vs_oracle(mode, tol_l, tol_p):
    """BASELINE configs[4] at full size: SigLIP-base with num_frames = 64, one 224^2 frame per call through the KV-cache,
    64 calls — against the ORACLE's full-clip forward of the same 64 frames (causal temporal attention makes the two
    identical in exact arithmetic; vqa_enc:491-560, 1316-1392).  The streamed path runs the small-M kernels (register-direct
    GEMM with the LayerNorm fold, the cache-append epilogue, temporal attention over the growing cache) inside hipGraph
    replays; a second pass over the same stream after cache.reset() must reproduce the first bit for bit."""
    import streamformer_amd as sa
    cfg = siglip_base(num_frames=64)
    sd = make_state_dict(cfg, seed=0)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    x = frames(64, (1, 64, 3, 224, 224))
    want = O.forward(sd, cfg, x)
    xd = x.cuda()
    cache = m.new_cache(1, 64)
    passes = []
    for rep in range(2):
        cache.reset()
        outs = [m(xd[:, t:t + 1], use_cache=True, past_key_values=cache) for t in range(64)]
        passes.append((torch.cat([o.last_hidden_state for o in outs], 1), torch.cat([o.pooler_output for o in outs], 1)))
    lhs, pool = passes[0]
    per_frame = (lhs.cpu() - want["last_hidden_state"]).abs().amax(dim=(0, 2, 3))
    assert float(per_frame.max()) <= tol_l, per_frame.tolist()
    assert maxabs(pool, want["pooler_output"]) <= tol_p
    assert cosine(lhs, want["last_hidden_state"]) >= (0.9995 if mode == "bf16" else 0.999999)
    assert torch.equal(passes[0][0], passes[1][0]) and torch.equal(passes[0][1], passes[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol_l,tol_p", [("fp32", ACC_TOL, ACC_TOL), ("bf16", BF16_LHS, BF16_POOL)])
@pytest.mark.parametrize("streams,nframes", [(4, 3), (9, 2)])
def test_several_streams_per_call_vs_oracle(mode, tol_l, tol_p, streams, nframes):
    """The vision tower's serving shape at SigLIP-base size: one cache, `streams` independent clips, one new frame of each
    per call (M = 784 / 1764 rows: the 64 x 64 GEMM tiles with the in-kernel LayerNorm fold in bf16 mode, the large split
    tiles in the accurate mode), against the oracle's full-clip forward of every stream; and the short-clip forward of the
    same frames outside streaming (same kernels, no cache)."""
    import streamformer_amd as sa
    cfg = siglip_base(num_hidden_layers=3)
    sd = make_state_dict(cfg, seed=4)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    x = frames(21, (streams, nframes, 3, 224, 224))
    want = O.forward(sd, cfg, x)
    xd = x.cuda()
    cache = m.new_cache(streams, cfg.num_frames)
    outs = [m(xd[:, t:t + 1], use_cache=True, past_key_values=cache) for t in range(nframes)]
    lhs = torch.cat([o.last_hidden_state for o in outs], 1)
    pool = torch.cat([o.pooler_output for o in outs], 1)
    assert maxabs(lhs, want["last_hidden_state"]) <= tol_l and maxabs(pool, want["pooler_output"]) <= tol_p
    full = m(xd)
    assert maxabs(full.last_hidden_state, want["last_hidden_state"]) <= tol_l and maxabs(full.pooler_output, want["pooler_output"]) <= tol_p
    again = m(xd)                                   # no atomics, fixed reduction orders: bit-reproducible
    assert torch.equal(full.last_hidden_state, again.last_hidden_state) and torch.equal(full.pooler_output, again.pooler_output)
    cache.reset()
    outs2 = [m(xd[:, t:t + 1], use_cache=True, past_key_values=cache) for t in range(nframes)]
    assert torch.equal(torch.cat([o.last_hidden_state for o in outs2], 1), lhs)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol_l,tol_p", [("fp32", ACC_TOL, ACC_TOL), ("bf16", BF16_LHS, BF16_POOL)])
def test_forward_non_base_width_vs_oracle(mode, tol_l, tol_p):
    """StreamformerConfig accepts any hidden_size / heads (models/configuration_streamformer.py:90-135); a ViT-L-shaped
    encoder (D = 1024, 16 heads, I = 4096, head_dim 64) leaves every SigLIP-base-only fast path (N = 768 panel tiles, the plane-form
    residual stream) and must still equal the oracle: whole clips at two batch sizes (M = 392 on the small-M kernels, M = 6272 on
    the generic 128^2 / 256^2 tiles), the pooling head at 16 heads, and three streamed frames through the KV-cache (the head's
    row-vector tail at K = 4096)."""
    import streamformer_amd as sa
    from streamformer_amd.configuration import StreamformerConfig
    cfg = StreamformerConfig(image_size=224, patch_size=16, num_frames=16, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16,
                             intermediate_size=4096, enable_causal_temporal=True)
    sd = make_state_dict(cfg, seed=12)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    # M = 392 (small-M kernels), 6 272 (generic tiles, fp32 residual stream + standalone LayerNorms), 9 408 (round 5: plane-form residual
    # stream + LayerNorm fold on the 256^2 kernel as producer and consumer, both modes)
    for B, T, seed in ((1, 2, 5), (2, 16, 6), (3, 16, 8)):
        x = frames(seed, (B, T, 3, 224, 224))
        want = O.forward(sd, cfg, x)
        out = m(x.cuda())
        assert out.last_hidden_state.shape == (B, T, 196, 1024) and out.pooler_output.shape == (B, T, 1024)
        assert maxabs(out.last_hidden_state, want["last_hidden_state"]) <= tol_l, (B, T)
        assert maxabs(out.pooler_output, want["pooler_output"]) <= tol_p, (B, T)
        again = m(x.cuda())
        assert torch.equal(out.last_hidden_state, again.last_hidden_state) and torch.equal(out.pooler_output, again.pooler_output)
    x = frames(7, (1, 3, 3, 224, 224))
    want = O.forward(sd, cfg, x)
    cache = m.new_cache(1, 16)
    outs = [m(x.cuda()[:, t:t + 1], use_cache=True, past_key_values=cache) for t in range(3)]
    assert maxabs(torch.cat([o.last_hidden_state for o in outs], 1), want["last_hidden_state"]) <= tol_l
    assert maxabs(torch.cat([o.pooler_output for o in outs], 1), want["pooler_output"]) <= tol_p


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol_l,tol_p", [("fp32", ACC_TOL, ACC_TOL), ("bf16", BF16_LHS, BF16_POOL)])
def test_forward_three_heads_vs_oracle(mode, tol_l, tol_p):
    """D = 192 / 3 heads: the ragged k-step splits of the pooling-head kernels (six k-steps over four waves) in whole clips (token splits
    S = 1 and S > 1) and streamed frames, against the oracle."""
    import streamformer_amd as sa
    cfg = small_cfg(hidden_size=192, num_attention_heads=3, intermediate_size=384, image_size=96)      # 36 patches
    sd = make_state_dict(cfg, seed=31)
    m = build(sa, cfg, sd, mode)
    for B, T, seed in ((1, 3, 1), (20, 16, 2)):
        x = frames(seed, (B, T, 3, 96, 96))
        want = O.forward(sd, cfg, x)
        out = m(x.cuda())
        assert maxabs(out.last_hidden_state, want["last_hidden_state"]) <= tol_l and maxabs(out.pooler_output, want["pooler_output"]) <= tol_p, (B, T)
    x = frames(3, (2, 4, 3, 96, 96))
    want = O.forward(sd, cfg, x)
    cache = m.new_cache(2, 16, 96, 96)
    outs = [m(x.cuda()[:, t:t + 1], use_cache=True, past_key_values=cache) for t in range(4)]
    assert maxabs(torch.cat([o.last_hidden_state for o in outs], 1), want["last_hidden_state"]) <= tol_l
    assert maxabs(torch.cat([o.pooler_output for o in outs], 1), want["pooler_output"]) <= tol_p


@pytest.mark.gpu
def test_unsupported_widths_are_refused_with_a_message():
    """What the HIP path still cannot run is refused at construction with SF_ERR_INVALID and a message that names the limit — not a wrong
    answer, not a crash at the first forward: more than 16 heads, a head_dim that is not a multiple of 8 or above 128, a hidden_size that
    is not a multiple of 64.  (Round 6: head_dim 72 / intermediate 4304 / 14 x 14 patches — the SigLIP-so400m shape — are no longer here.)"""
    import streamformer_amd as sa
    import streamformer_amd._native as nat
    from streamformer_amd.configuration import StreamformerConfig
    for kw, word in ((dict(hidden_size=1280, num_attention_heads=20, intermediate_size=5120), "heads"),
                     (dict(hidden_size=320, num_attention_heads=2, intermediate_size=640), "head_dim"),          # 160 > 128
                     (dict(hidden_size=192, num_attention_heads=16, intermediate_size=384), "head_dim"),         # 12: not a multiple of 8
                     (dict(hidden_size=144, num_attention_heads=2, intermediate_size=304), "hidden_size")):      # F13's 72-wide heads at 144
        cfg = StreamformerConfig(image_size=224, patch_size=16, num_frames=16, num_hidden_layers=1, enable_causal_temporal=True, **kw)
        with pytest.raises(nat.NativeError) as ei:
            m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
            m.load_state_dict(make_state_dict(cfg, seed=1))
            m.to("cuda").eval()(frames(1, (1, 1, 3, 224, 224)).cuda())
        assert word in str(ei.value)


HD72W = dict(image_size=42, patch_size=14, num_frames=8, hidden_size=576, num_hidden_layers=2, num_attention_heads=8, intermediate_size=1072)
HD32 = dict(image_size=48, patch_size=16, num_frames=8, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol_l,tol_p", [("fp32", ACC_TOL, ACC_TOL), ("bf16", BF16_LHS, BF16_POOL)])
@pytest.mark.parametrize("fixture,tag,kw,wseed,xseed", [("f15_hd72_hip.npz", "hd72w", HD72W, 15, 150), ("f13_widths.npz", "hd32", HD32, 13, 131)])
def test_forward_head_widths_other_than_64_vs_reference_fixture(golden_dir, mode, tol_l, tol_p, fixture, tag, kw, wseed, xseed):
    """configuration_streamformer.py:90-135 takes any hidden_size / heads.  head_dim 72 with intermediate 1072 (not a multiple of 64) and
    14 x 14 patches (C P P = 588) — SigLIP-so400m's shape in small — and head_dim 32, against the REFERENCE's outputs (F15 / F13, made by
    oracle/make_golden_widths_hip.py / make_golden_variants.py): generic-width attention + pooling-head kernels, zero-padded MLP and
    patch-embedding weights, generic patch extraction.  Streamed frame by frame through the KV-cache the same frames must agree with
    the full clip, bit-reproducibly."""
    import streamformer_amd as sa
    from streamformer_amd.configuration import StreamformerConfig
    g = load_npz(os.path.join(golden_dir, fixture))
    cfg = StreamformerConfig(enable_causal_temporal=True, **kw)
    sd = make_state_dict(cfg, seed=wseed)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    x = frames(xseed, (2, cfg.num_frames, 3, cfg.image_size, cfg.image_size)).cuda()
    out = m(x)
    assert maxabs(out.last_hidden_state, g[f"{tag}_last_hidden_state"]) <= tol_l
    assert maxabs(out.pooler_output, g[f"{tag}_pooler_output"]) <= tol_p
    again = m(x)
    assert torch.equal(out.last_hidden_state, again.last_hidden_state) and torch.equal(out.pooler_output, again.pooler_output)
    cache = m.new_cache(2, cfg.num_frames)
    outs = [m(x[:, t:t + 1], use_cache=True, past_key_values=cache) for t in range(cfg.num_frames)]
    lhs = torch.cat([o.last_hidden_state for o in outs], 1)
    pool = torch.cat([o.pooler_output for o in outs], 1)
    assert maxabs(lhs, g[f"{tag}_last_hidden_state"]) <= tol_l and maxabs(pool, g[f"{tag}_pooler_output"]) <= tol_p
    # uint8 frames through the generic patch extraction (rescale + normalise fused), against the same frames as floats
    u8 = (x[:1].clamp(-1, 1) * 127.5 + 127.5).round().to(torch.uint8)
    a = m(u8)
    b = m((u8.float() / 127.5 - 1.0))
    assert maxabs(a.last_hidden_state, b.last_hidden_state) <= (2e-4 if mode == "fp32" else tol_l)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_output_attentions_at_head_dim_72(mode, tol):
    """output_attentions=True (modeling:703-716) at a head width other than 64: the generic attention kernel's second sweep writes the
    spatial probabilities; against the ORACLE's (collect=...), rows sum to one, hidden states unchanged by the flag."""
    import streamformer_amd as sa
    from streamformer_amd.configuration import StreamformerConfig
    cfg = StreamformerConfig(enable_causal_temporal=True, **HD72W)
    sd = make_state_dict(cfg, seed=15)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    x = frames(151, (2, 4, 3, cfg.image_size, cfg.image_size))
    col = {}
    O.forward(sd, cfg, x, collect=col)
    want = torch.stack([a.reshape(2 * 4, cfg.num_attention_heads, 9, 9) for a in col["attentions"]])
    out = m(x.cuda(), output_attentions=True)
    got = torch.stack(list(out.attentions)).cpu()
    assert got.shape == want.shape and maxabs(got, want) <= tol
    assert float((got.sum(-1) - 1).abs().max()) < 1e-5
    # accurate mode: with the flag q / k / v reach the kernel as fp32, without it as hi + lo planes — same numbers to rounding, not to the bit
    plain = m(x.cuda()).last_hidden_state
    assert torch.equal(out.last_hidden_state, plain) if mode == "bf16" else maxabs(out.last_hidden_state, plain) <= 2e-5


@pytest.mark.gpu
def test_so400m_shaped_layer_runs_and_matches_the_oracle():
    """The real SigLIP-so400m widths (hidden 1152, 16 heads of 72, intermediate 4304, patch 14 at 224 x 224 = 256 tokens) on ONE layer and a
    4-frame clip: the generic kernels at full width against the CPU oracle, both modes."""
    import streamformer_amd as sa
    from streamformer_amd.configuration import StreamformerConfig
    cfg = StreamformerConfig(image_size=224, patch_size=14, num_frames=4, hidden_size=1152, num_hidden_layers=1, num_attention_heads=16,
                             intermediate_size=4304, enable_causal_temporal=True)
    sd = make_state_dict(cfg, seed=16)
    x = frames(16, (1, 4, 3, 224, 224))
    want = O.forward(sd, cfg, x)
    for mode, tl, tp in (("fp32", ACC_TOL, ACC_TOL), ("bf16", BF16_LHS, BF16_POOL)):
        m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
        m.load_state_dict(sd)
        m.to("cuda").eval()
        out = m(x.cuda())
        assert maxabs(out.last_hidden_state, want["last_hidden_state"]) <= tl, mode
        assert maxabs(out.pooler_output, want["pooler_output"]) <= tp, mode


@pytest.mark.gpu
def test_feature_extraction_harness():
    """features.py against the reference's extraction loops restated on the oracle: sliding 6-frame windows with the
    clamped tail (extract_oad_feature.py:34-35, 122-136), last-frame pooled feature per window; long-video per-frame
    features in num_frames clips with tail padding (modeling:1551-1621)."""
    import streamformer_amd as sa
    from streamformer_amd.features import long_video_features, sliding_window_features, window_starts
    cfg = small_cfg()
    sd = make_state_dict(cfg, seed=19)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="fp32")
    m.load_state_dict(sd)
    m.to("cuda")
    video = frames(19, (27, 3, 48, 48))                                  # 27 frames -> 4 windows, the last start == 27 (clamped)
    got = sliding_window_features(m, video, batch_windows=3)
    want = []
    for s in window_starts(27):
        q = video[27 - 6:] if s + 6 > 27 else video[s:s + 6]
        want.append(O.forward(sd, cfg, q[None])["pooler_output"][:, -1])
    want = torch.cat(want)
    assert got.shape == (4, 128) and got.dtype == np.float32
    assert maxabs(torch.from_numpy(got), want) <= ACC_TOL
    long = frames(20, (1, 40, 3, 48, 48))                                # 40 frames: 16 + 16 + 8 (padded to 16)
    feats = long_video_features(m, long, window_size=32)
    ref = []
    for i in range(0, 40, 16):
        clip = long[:, i:i + 16]
        n = clip.shape[1]
        if n < 16:
            clip = torch.cat([clip, torch.zeros(1, 16 - n, 3, 48, 48)], 1)
        ref.append(O.forward(sd, cfg, clip)["pooler_output"][:, :n])
    assert tuple(feats.shape) == (1, 40, 128)
    assert maxabs(feats, torch.cat(ref, 1)) <= ACC_TOL


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_plane_form_residual_stream_equals_the_fp32_one(sa, mode):
    """Whole forwards at BASELINE-sized M carry the residual stream as bf16 planes (hi + lo, + lo2 in the accurate mode) instead of
    fp32; asking for hidden_states keeps the fp32 stream.  Both must agree to the planes' precision — 4 clips
    (98-row panel tiles in bf16 mode), the unfused temporal projection pair (two residual-free + one residual GEMM), and against
    the oracle."""
    cfg = siglip_base()
    sd = make_state_dict(cfg, seed=2)
    m = build(sa, cfg, sd, mode, fuse=False)
    xc = torch.randn(4, 16, 3, 224, 224, generator=torch.Generator().manual_seed(77))
    x = xc.cuda()
    planes = m(x)
    plain = m(x, output_hidden_states=True)
    # bf16 mode: a 2^-18 difference in a residual row flips bf16 roundings of the next GEMM's operand, so two schedules sit one
    # operand-rounding apart (the same bound as batch vs solo in test_baseline_batch8_properties); accurate mode: the bf16x3 operand precision
    tol = 4e-2 if mode == "bf16" else 1e-4          # accurate: measured 5.3e-5 (fold + planes against LayerNorm launches + fp32: different operand splits)
    d1 = maxabs(planes.last_hidden_state, plain.last_hidden_state)
    d2 = maxabs(planes.pooler_output, plain.pooler_output)
    print(f"[{mode}] planes vs fp32 stream: lhs {d1:.3e} pooler {d2:.3e}")
    assert d1 <= tol and d2 <= tol
    assert d1 > 0 or d2 > 0          # two different schedules ran
    want = O.forward(sd, cfg, xc[2:3])
    lt, pt = (ACC_TOL, ACC_TOL) if mode == "fp32" else (BF16_LHS, BF16_POOL)
    assert maxabs(planes.last_hidden_state[2], want["last_hidden_state"][0]) <= lt
    assert maxabs(planes.pooler_output[2], want["pooler_output"][0]) <= pt


def test_two_clips_run_statistics_producing_tiles(sa, switches):
    """Two clips per call (M = 6 272; README.md:55-71 at B = 2): the residual producers stay on the narrow tile kernel but emit the
    LayerNorm row statistics (128 x 192 tiles: four pairs per row), the folded consumers run on the 256^2 kernel.  Against the
    oracle, against the in-kernel-statistics schedule it replaces (SF_TILE_FOLD_MIN_M above M), bit-reproducible, clips
    independent, and the fp32 hidden_states route of the same schedule; a 24-frame clip (M = 4 704, ragged 128-row tiles) too."""
    cfg = siglip_base()
    sd = make_state_dict(cfg, seed=0)
    m = build(sa, cfg, sd, "bf16")
    xc = torch.randn(2, 16, 3, 224, 224, generator=torch.Generator().manual_seed(21))
    out = m(xc.cuda())
    want = O.forward(sd, cfg, xc[1:2])
    assert maxabs(out.last_hidden_state[1], want["last_hidden_state"][0]) <= BF16_LHS
    assert maxabs(out.pooler_output[1], want["pooler_output"][0]) <= BF16_POOL
    again = m(xc.cuda())
    assert torch.equal(again.last_hidden_state, out.last_hidden_state) and torch.equal(again.pooler_output, out.pooler_output)
    x2 = xc.clone(); x2[0] = -x2[0]
    assert torch.equal(m(x2.cuda()).last_hidden_state[1], out.last_hidden_state[1])
    hs = m(xc.cuda(), output_hidden_states=True)
    assert maxabs(hs.last_hidden_state[1], want["last_hidden_state"][0]) <= BF16_LHS
    x24 = torch.randn(1, 24, 3, 224, 224, generator=torch.Generator().manual_seed(22))
    o24 = m(x24.cuda())
    w24 = O.forward(sd, cfg, x24)
    assert maxabs(o24.last_hidden_state, w24["last_hidden_state"]) <= BF16_LHS
    assert maxabs(o24.pooler_output, w24["pooler_output"]) <= BF16_POOL
    switches("SF_TILE_FOLD_MIN_M", 1 << 30)              # the schedule it replaces: statistics inside the wide tile consumers
    old = m(xc.cuda())
    d = maxabs(old.last_hidden_state, out.last_hidden_state)
    assert 0 < d <= 4e-2, d                              # two schedules, one operand rounding apart
    assert maxabs(old.last_hidden_state[1], want["last_hidden_state"][0]) <= BF16_LHS


def test_five_clips_run_the_folded_schedule(sa):
    """Five and six clips fill 59 % / 71 % of the panel kernel's MFMA rows; from five clips on the folded schedule (panel producers,
    plane-form residual, 256^2 consumers) is taken anyway (sf_gemm_panel.hip: panel_plan).  Against the oracle, clip by clip
    independence included."""
    cfg = siglip_base()
    sd = make_state_dict(cfg, seed=0)
    m = build(sa, cfg, sd, "bf16")
    xc = torch.randn(5, 16, 3, 224, 224, generator=torch.Generator().manual_seed(9))
    out = m(xc.cuda())
    want = O.forward(sd, cfg, xc[4:5])
    assert maxabs(out.last_hidden_state[4], want["last_hidden_state"][0]) <= BF16_LHS
    assert maxabs(out.pooler_output[4], want["pooler_output"][0]) <= BF16_POOL
    x2 = xc.clone(); x2[:4] = -x2[:4]
    out2 = m(x2.cuda())
    assert torch.equal(out2.last_hidden_state[4], out.last_hidden_state[4])
