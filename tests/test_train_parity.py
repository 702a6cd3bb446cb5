"""GPU parity of the training step against oracle/train_oracle.py — the CPU restatement
that oracle/make_golden_train.py pins to the reference's own modules (fixture F8) — and against plain
torch autograd for the single backward operators.  Everything goes through the C ABI.

Tolerances (bf16 MFMA operands, fp32 accumulation — the `dtype` of BASELINE configs #3/#4):
  single operators : max-abs error <= 2e-2 of the tensor's max-abs (inputs pre-rounded to bf16, so the
                     only differences are the bf16 rounding of P/dS/outputs and summation order)
  whole-model grads: relative L2 error per tensor <= 2.6e-2 (measured 1.7e-2), cosine >= 0.9995 vs the fp32 oracle; 0-dim
                     parameters (temporal gates, logit scale / bias) are single sums over ~1e6 bf16-rounded
                     products with heavy cancellation: <= 10 % of the value (measured 6.7 %)
"""
import math
import os

import numpy as np
import pytest
import torch

from tests.helpers import load_npz, small_cfg

pytestmark = pytest.mark.gpu

OP_TOL = 2e-2
GRAD_REL_L2 = 2.6e-2        # measured worst 1.7e-2 (head.probe, small model), 1.1e-2 on SigLIP-base; + 50 %
GRAD_COS = 0.9995           # measured lowest 0.99987
SCALAR_REL = 0.10           # measured worst 6.7e-2 (a temporal gate: one sum over ~1e6 bf16-rounded products)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def rel_max(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return float((got - want).abs().max() / (want.abs().max() + 1e-30))


def rel_l2(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return float((got - want).norm() / (want.norm() + 1e-30))


def cosine(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


# ---------------------------------------------------------------------------------------------------
# single operators
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N1,N2", [(3136, 768, 768), (777, 136, 72), (40, 128, 256), (25088, 768, 3072), (130, 64, 64), (25088, 2304, 768), (2100, 256, 512)])
def test_wgrad_matches_torch(M, N1, N2):
    import streamformer_amd._native as nat
    dev = _dev()
    g = torch.Generator().manual_seed(M + N1)
    dy = torch.randn(M, N1, generator=g).to(dev).bfloat16()
    x = torch.randn(M, N2, generator=g).to(dev).bfloat16()
    base = torch.randn(N1, N2, generator=g).to(dev)
    out = base.clone()
    db = torch.full((N1,), 3.0, device=dev)
    nat.check(nat.lib.sf_op_wgrad(dy.data_ptr(), N1, x.data_ptr(), N2, M, N1, N2, 0.5, 1, out.data_ptr(), N2, db.data_ptr(),
                                  nat.current_stream_handle(dev)))
    want = base.double() + 0.5 * dy.double().t() @ x.double()
    assert rel_max(out, want) < 1e-4          # fp32 accumulation of exact bf16 products
    assert rel_max(db - 3.0, 0.5 * dy.double().sum(0)) < 1e-4      # bias gradient of the same Linear
    # column-sliced operands (leading dimension > width), no accumulate
    out2 = torch.full((N1, N2), 7.0, device=dev)
    wide = torch.randn(M, N1 + 64, generator=g).to(dev).bfloat16()
    nat.check(nat.lib.sf_op_wgrad(wide.data_ptr(), N1 + 64, x.data_ptr(), N2, M, N1, N2, 1.0, 0, out2.data_ptr(), N2, 0,
                                  nat.current_stream_handle(dev)))
    assert rel_max(out2, wide[:, :N1].double().t() @ x.double()) < 1e-4


def _attn_ref(qkv, d_o, nseq, L, heads, causal):
    """torch autograd reference on [nseq, L, 3D] fp64 tensors -> (o, d_qkv)."""
    D = heads * 64
    t = qkv.double().clone().requires_grad_(True)
    q, k, v = (t[..., i * D:(i + 1) * D].reshape(nseq, L, heads, 64).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), float("-inf"))
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(nseq, L, D)
    o.backward(d_o.double())
    return o.detach(), t.grad


@pytest.mark.parametrize("nseq,L,heads", [(3, 196, 2), (2, 9, 2), (1, 224, 1), (5, 33, 3), (2, 193, 2), (1, 208, 1), (1, 192, 2), (2, 209, 1)])
def test_spatial_attention_bwd_matches_autograd(nseq, L, heads):
    import streamformer_amd._native as nat
    dev = _dev()
    D = heads * 64
    g = torch.Generator().manual_seed(L)
    qkv = (torch.randn(nseq, L, 3 * D, generator=g) * 1.5).bfloat16()
    d_o = torch.randn(nseq, L, D, generator=g).bfloat16()
    o_ref, dqkv_ref = _attn_ref(qkv.float(), d_o.float(), nseq, L, heads, False)
    o = o_ref.bfloat16()
    dq = torch.full((nseq, L, 3 * D), float("nan")).bfloat16().to(dev)
    qd, od, dod = qkv.to(dev), o.to(dev), d_o.to(dev)
    nat.check(nat.lib.sf_op_attention_bwd(qd.data_ptr(), od.data_ptr(), dod.data_ptr(), dq.data_ptr(), 0, nseq, L, 1, heads, 0,
                                          nat.current_stream_handle(dev)))
    torch.cuda.synchronize()
    scale = float(dqkv_ref.abs().max())
    for i, name in enumerate("qkv"):
        e = float((dq[..., i * D:(i + 1) * D].double().cpu() - dqkv_ref[..., i * D:(i + 1) * D]).abs().max()) / scale
        assert e < OP_TOL, (name, e)


@pytest.mark.parametrize("B,N,L,heads,causal", [(2, 5, 16, 2, 1), (1, 3, 4, 2, 1), (2, 2, 24, 1, 1), (1, 4, 16, 2, 0), (1, 2, 32, 1, 1), (1, 2, 1, 1, 1)])
def test_temporal_attention_bwd_matches_autograd(B, N, L, heads, causal):
    import streamformer_amd._native as nat
    dev = _dev()
    D = heads * 64
    g = torch.Generator().manual_seed(100 + L)
    # token row of (b, t, n) = (b*L + t)*N + n
    qkv = (torch.randn(B, L, N, 3 * D, generator=g) * 1.5).bfloat16()
    d_o = torch.randn(B, L, N, D, generator=g).bfloat16()
    seq = qkv.float().permute(0, 2, 1, 3).reshape(B * N, L, 3 * D)
    o_ref, dqkv_ref = _attn_ref(seq, d_o.float().permute(0, 2, 1, 3).reshape(B * N, L, D), B * N, L, heads, bool(causal))
    o = o_ref.reshape(B, N, L, D).permute(0, 2, 1, 3).contiguous().bfloat16()
    want = dqkv_ref.reshape(B, N, L, 3 * D).permute(0, 2, 1, 3)
    dq = torch.full((B, L, N, 3 * D), float("nan")).bfloat16().to(dev)
    qd, od, dod = qkv.to(dev), o.to(dev), d_o.to(dev)
    nat.check(nat.lib.sf_op_attention_bwd(qd.data_ptr(), od.data_ptr(), dod.data_ptr(), dq.data_ptr(), 1, B * N, L, N, heads, causal,
                                          nat.current_stream_handle(dev)))
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    for i, name in enumerate("qkv"):
        e = float((dq[..., i * D:(i + 1) * D].double().cpu() - want[..., i * D:(i + 1) * D]).abs().max()) / scale
        assert e < OP_TOL, (name, e)


@pytest.mark.parametrize("rows,D", [(1000, 768), (37, 128), (5000, 64)])
def test_layernorm_bwd_matches_autograd(rows, D):
    import streamformer_amd._native as nat
    dev = _dev()
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g) * 2 + 0.3
    dy = torch.randn(rows, D, generator=g)
    gamma = torch.randn(D, generator=g)
    g_in = torch.randn(rows, D, generator=g)
    xr = x.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = torch.zeros(D, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6).backward(dy.double())
    xd, dyd, gd, gind = x.to(dev), dy.to(dev), gamma.to(dev), g_in.to(dev)
    dx = torch.empty_like(xd)
    dg = torch.ones(D, device=dev)
    db = torch.ones(D, device=dev)
    nat.check(nat.lib.sf_op_layernorm_bwd(xd.data_ptr(), dyd.data_ptr(), gd.data_ptr(), gind.data_ptr(), dx.data_ptr(),
                                          dg.data_ptr(), db.data_ptr(), rows, D, 1e-6, nat.current_stream_handle(dev)))
    assert rel_max(dx, xr.grad + g_in.double()) < 1e-5
    assert rel_max(dg - 1, gr.grad) < 1e-5 and rel_max(db - 1, br.grad) < 1e-5


# ---------------------------------------------------------------------------------------------------
# whole model: gradients of one micro-step vs the oracle
# ---------------------------------------------------------------------------------------------------
def _trainer_and_oracle(cfg, freeze, seed, lora, lr=1e-3, wd=0.05):
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    sd = make_state_dict(cfg, seed=seed, lora=lora)
    tr = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], freeze_spatial=freeze, device=_dev(), lr=lr, weight_decay=wd)
    orc = TO.OracleTrainer(sd, cfg, ["retrieval", "localization"], freeze_spatial=freeze, lr=lr, weight_decay=wd)
    return tr, orc


def _to_dev(ti, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in ti.items()}


def _compare_grads(tr, orc, floor=1e-3, scalar_rel=SCALAR_REL):
    og = orc.grads()
    names = set(tr.parameter_names(trainable_only=True))
    assert names == set(og), sorted(names ^ set(og))[:6]
    gmax = max(float(v.abs().max()) for v in og.values())
    report = {}
    for n, want in og.items():
        got = tr.grad(n).detach().cpu()
        if float(want.abs().max()) < floor * gmax * 1e-3:        # gradient that is (numerically) zero: absolute check
            assert float((got - want).abs().max()) < floor * gmax, n
            continue
        report[n] = (rel_l2(got, want), cosine(got, want))
    scalars = {n: r for n, r in report.items() if og[n].numel() == 1}
    report = {n: r for n, r in report.items() if og[n].numel() > 1}
    if os.environ.get("SF_TEST_VERBOSE"):
        for n in scalars:
            print(f"[0-dim] {n}: got {float(tr.grad(n)):+.5e} want {float(og[n]):+.5e} rel {scalars[n][0]:.3e}")
    # 0-dim parameters: one sum over ~1e6 bf16-rounded products with heavy cancellation.  A gate whose gradient happens to cancel to
    # 1e-4 when its eleven siblings sit at 5e-2 carries the same ABSOLUTE error as they do, so the bound is relative to the
    # larger of the value and the family's median magnitude (family = last name component: the twelve temporal gates, ...)
    fam = {}
    for n in scalars:
        fam.setdefault(n.rsplit(".", 1)[-1], []).append(abs(float(og[n])))
    for n, r in scalars.items():
        w = float(og[n])
        scale = max(abs(w), float(np.median(fam[n.rsplit(".", 1)[-1]])))
        assert abs(float(tr.grad(n)) - w) < scalar_rel * scale, (n, r, float(tr.grad(n)), w, scale)
    worst = max(report.items(), key=lambda kv: kv[1][0])
    wc = min(report.items(), key=lambda kv: kv[1][1])
    ws = max(scalars.items(), key=lambda kv: kv[1][0]) if scalars else ("-", (0.0, 1.0))
    print(f"[grad parity] worst rel-L2 {worst[1][0]:.3e} ({worst[0]}), lowest cosine {wc[1][1]:.6f} ({wc[0]}), worst 0-dim rel {ws[1][0]:.3e} ({ws[0]})")
    assert worst[1][0] < GRAD_REL_L2, worst
    assert wc[1][1] > GRAD_COS, wc
    return report


@pytest.mark.parametrize("task_idx", [0, 1])
@pytest.mark.parametrize("lora,freeze", [(True, True), (False, False), (True, False)])
def test_small_model_gradients_match_oracle(task_idx, lora, freeze):
    from oracle import train_oracle as TO
    cfg = small_cfg(add_lora_spatial=lora)
    tr, orc = _trainer_and_oracle(cfg, freeze, seed=8, lora=lora)
    task, x, ti, _ = TO.schedule(cfg)[task_idx]
    want_loss = orc.loss(task, x, ti)
    want_loss.backward()
    dev = tr.device
    _, pooler = tr.forward(x.to(dev))
    loss, gp, gs = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
    tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
    tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
    tr.backward(gp)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(want_loss)) < 2e-2 * abs(float(want_loss))
    _compare_grads(tr, orc)


def test_three_head_model_gradients_match_oracle():
    """D = 192 (three heads of 64): six k-steps of 32 do not divide over the four waves / the instantiated k-step counts of the pooling-head
    kernels (forward K-split, backward KS = 8 over nks = 6) and the fused temporal projections run at D % 128 != 0 — the ragged paths
    (clamped addresses, zeroed fragments) against the oracle's autograd, both tasks."""
    from oracle import train_oracle as TO
    cfg = small_cfg(add_lora_spatial=True, hidden_size=192, num_attention_heads=3, intermediate_size=384)
    for task_idx in (0, 1):
        tr, orc = _trainer_and_oracle(cfg, True, seed=9, lora=True)
        task, x, ti, _ = TO.schedule(cfg)[task_idx]
        want_loss = orc.loss(task, x, ti)
        want_loss.backward()
        dev = tr.device
        _, pooler = tr.forward(x.to(dev))
        loss, gp, gs = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
        tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
        tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
        tr.backward(gp)
        torch.cuda.synchronize()
        assert abs(float(loss) - float(want_loss)) < 2e-2 * abs(float(want_loss))
        _compare_grads(tr, orc)


def test_wide_model_gradients_match_oracle():
    """D = 1024 / 16 heads / I = 4096 (a ViT-L-shaped layer): the trainer's kernels at the widest supported shape — the pooling-head
    backward's LDS images need more token splits there, the fused temporal projections and the D x D algebra run at D = 1024."""
    from oracle import train_oracle as TO
    cfg = small_cfg(add_lora_spatial=True, hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=1, image_size=224,
                    num_frames=4)
    tr, orc = _trainer_and_oracle(cfg, True, seed=10, lora=True)
    task, x, ti, _ = TO.schedule(cfg, B=2)[1]
    want_loss = orc.loss(task, x, ti)
    want_loss.backward()
    dev = tr.device
    _, pooler = tr.forward(x.to(dev))
    loss, gp, gs = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
    tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
    tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
    tr.backward(gp)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(want_loss)) < 2e-2 * abs(float(want_loss))
    _compare_grads(tr, orc)


def test_gradients_are_deterministic_and_accumulate():
    from oracle import train_oracle as TO
    cfg = small_cfg(add_lora_spatial=True)
    tr, _ = _trainer_and_oracle(cfg, True, seed=8, lora=True)
    task, x, ti, _ = TO.schedule(cfg)[1]
    dev = tr.device

    def run():
        _, pooler = tr.forward(x.to(dev))
        _, gp, _ = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
        tr.backward(gp)
        torch.cuda.synchronize()
    run()
    g1 = tr.grads.clone()
    tr.zero_grad()
    run()
    assert torch.equal(g1, tr.grads)          # fixed-order reductions: bit-reproducible
    run()
    assert rel_max(tr.grads, 2 * g1) < 1e-6   # += semantics for update_freq > 1


def test_f8_three_optimizer_steps_follow_the_reference(golden_dir):
    """losses of the 4 micro-batches and the parameter updates after 3 AdamW steps vs fixture F8 (made
    by the reference's modules); Adam amplifies noise on near-zero gradients, so the update is compared
    on the elements whose first-step gradient is significant (same rule as make_golden_train.py)."""
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import make_state_dict, state_dict_sha256
    f8 = load_npz(os.path.join(golden_dir, "f8_train.npz"))
    cfg = small_cfg(add_lora_spatial=True)
    sd = make_state_dict(cfg, seed=8, lora=True)
    if state_dict_sha256(sd) != str(f8["sha256"]):
        pytest.skip("seeded weights differ from the fixture's (RNG drift on this box)")
    tr, _ = _trainer_and_oracle(cfg, True, seed=8, lora=True, lr=float(f8["lr"]), wd=float(f8["wd"]))
    dev = tr.device
    init = tr.state_dict()
    losses = []
    for i, (task, x, ti, uf) in enumerate(TO.schedule(cfg)):
        if i == 0:       # first-step gradients against the reference's
            _, pooler = tr.forward(x.to(dev))
            _, gp, gs = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
            tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
            tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
            tr.backward(gp)
            for k in f8:
                if k.startswith("grad0/"):
                    n = k[len("grad0/"):]
                    want = torch.from_numpy(f8[k])
                    got = tr.grad(n).cpu().reshape(want.shape)
                    if float(want.abs().max()) > 1e-6:
                        assert rel_l2(got, want) < GRAD_REL_L2 and cosine(got, want) > GRAD_COS, n
            tr.zero_grad()
        losses.append(float(tr.micro_step(task, x.to(dev), _to_dev(ti, dev), update_freq=uf)))
    want_losses = f8["losses"]
    assert np.allclose(losses, want_losses, rtol=3e-2), (losses, want_losses)
    assert tr.step_count == 3
    after = tr.state_dict()
    lr = float(f8["lr"])
    for k in f8:
        if not k.startswith("param/"):
            continue
        n = k[len("param/"):]
        want = torch.from_numpy(f8[k]).double()
        got = after[n].cpu().double().reshape(want.shape)
        assert float((got - want).abs().max()) <= 6 * lr, n   # two trajectories, each moving <= ~lr per step
        if "grad0/" + n in f8:
            g0 = torch.from_numpy(f8["grad0/" + n]).abs()
            sig = g0 > 0.05 * g0.max()
            if int(sig.sum()) > 8:
                upd_w = (want - init[n].cpu().double().reshape(want.shape))[sig]
                upd_g = (got - init[n].cpu().double().reshape(want.shape))[sig]
                assert float((upd_w - upd_g).norm() / (upd_w.norm() + 1e-30)) < 0.15, n
    for k in f8:
        if k.startswith("paramnorm/"):
            n = k[len("paramnorm/"):]
            assert abs(float(after[n].double().norm()) - float(f8[k])) <= 2e-3 * float(f8[k]) + 3 * lr * math.sqrt(after[n].numel()), n


def test_base_model_gradients_match_oracle():
    """SigLIP-base (LoRA recipe), one clip of 4 frames: every trainable tensor's gradient vs CPU autograd."""
    from oracle import train_oracle as TO
    from streamformer_amd.configuration import siglip_base
    cfg = siglip_base(add_lora_spatial=True)
    tr, orc = _trainer_and_oracle(cfg, True, seed=3, lora=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 3, 224, 224, generator=g)
    lab_emb = torch.randn(20, cfg.hidden_size, generator=g)
    lab_emb = lab_emb / lab_emb.norm(dim=-1, keepdim=True)
    ti = {"kind": "localization", "label_emb": lab_emb, "labels": torch.randint(-1, 20, (1, 4), generator=g)}
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // 2))
    want = orc.loss("localization", x, ti)
    want.backward()
    dev = tr.device
    _, pooler = tr.forward(x.to(dev))
    loss, gp, gs = tr.loss_and_grad("localization", pooler, _to_dev(ti, dev))
    tr.grad("task_heads.localization.logit_scale").add_(gs[0])
    tr.grad("task_heads.localization.logit_bias").add_(gs[1])
    tr.backward(gp)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(want)) < 2e-2 * abs(float(want))
    _compare_grads(tr, orc)


def _base_grad_case(B, T, task, seed, freeze=True):
    from oracle import train_oracle as TO
    from streamformer_amd.configuration import siglip_base
    cfg = siglip_base(add_lora_spatial=True)
    tr, orc = _trainer_and_oracle(cfg, freeze, seed=seed, lora=True)
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.randn(B, T, 3, 224, 224, generator=g)
    if task == "localization":
        lab_emb = torch.randn(20, cfg.hidden_size, generator=g)
        lab_emb = lab_emb / lab_emb.norm(dim=-1, keepdim=True)
        ti = {"kind": "localization", "label_emb": lab_emb, "labels": torch.randint(-1, 20, (B, T), generator=g)}
    else:
        ti = {"kind": "retrieval", "text": torch.randn(B, cfg.hidden_size, generator=g)}
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // 2))
    want = orc.loss(task, x, ti)
    want.backward()
    dev = tr.device
    _, pooler = tr.forward(x.to(dev))
    loss, gp, gs = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
    tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
    tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
    tr.backward(gp)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(want)) < 2e-2 * abs(float(want))
    return _compare_grads(tr, orc)


def test_base_model_gradients_match_oracle_at_sixteen_frames():
    """the T = 16 temporal backward at D = 768 / 12 heads — SigLIP-base (LoRA recipe), ONE clip of the
    16 frames the bench times, every trainable tensor's gradient vs CPU autograd (M = 3136 token rows)."""
    _base_grad_case(1, 16, "retrieval", seed=4)


def test_base_model_gradients_lora_with_unfrozen_base():
    """the shipped pretraining recipe passes --enable_lora_spatial WITHOUT --frozen_spatial
    (scripts/pretrain_streamformer.sh:32-33): the LoRA-adapted spatial Linears then need the rank-32 factor gradients AND the full
    [3D, D] / [D, D] base-weight gradients.  The side stream's scratch is sized for the rank-32 shapes only, so these Linears must stay
    on the caller's stream (lin_wgrad_side).  SigLIP-base, one clip of 16 frames (M = 3136: groupable weight-gradient shapes), every
    trainable tensor against CPU autograd."""
    _base_grad_case(1, 16, "localization", seed=7, freeze=False)


@pytest.mark.skipif(os.environ.get("SF_TEST_BIG_TILES_INNER") != "1", reason="run by test_base_model_gradients_on_the_big_tile_kernels in a child process")
def test_base_model_gradients_big_tiles_inner():
    _base_grad_case(2, 16, "localization", seed=6)


def test_base_model_gradients_on_the_big_tile_kernels():
    """Two clips x 16 frames (M = 6272) with the one-to-two-clip tile family switched off, so that forward and backward take
    the panel / 256^2 kernels and their 98- / 196-row tiles — the dispatch of the 8-clip step — against CPU autograd.  The
    switch is read once per process: the case runs in a child interpreter."""
    import subprocess
    import sys
    _dev()
    env = dict(os.environ, SF_TEST_BIG_TILES_INNER="1", SF_DISABLE_GEMM_TILE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-s", "-k",
                        "test_base_model_gradients_big_tiles_inner"], env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    print([l for l in r.stdout.splitlines() if "grad parity" in l])


# ---------------------------------------------------------------------------------------------------
# data parallel: 2 ranks (both on cuda:0, gloo) == one rank on the concatenated batch
# ---------------------------------------------------------------------------------------------------
def _gn_worker(rank, world, reduce_dtype):
    """One rank of the retrieval micro-step with cross-rank negatives, driven through StreamformerTrainer."""
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    torch.cuda.set_device(0)
    cfg = small_cfg(add_lora_spatial=True)
    sd = make_state_dict(cfg, seed=8, lora=True)
    tr = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], freeze_spatial=True, device="cuda:0", bucket_mb=0.05,
                             grad_reduce_dtype=reduce_dtype)
    task, x, ti, _ = TO.schedule(cfg, B=4)[0]
    assert task == "retrieval"
    lo, hi = rank * 2, rank * 2 + 2
    # default: gather_negatives on when world > 1
    _, pooler = tr.forward(x[lo:hi].cuda())
    loss, gp, gs = tr.loss_and_grad(task, pooler, {"kind": "retrieval", "text": ti["text"][lo:hi].cuda()})
    tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
    tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
    tr.backward(gp, reduce=True)
    torch.cuda.synchronize()
    grads = (tr.grads / world).cpu()
    # the same through micro_step (forward + loss + backward + all-reduce + clipped AdamW): the whole enqueue path;
    # lr = 0 keeps the parameters
    tr.zero_grad()
    l2 = tr.micro_step(task, x[lo:hi].cuda(), {"kind": "retrieval", "text": ti["text"][lo:hi].cuda()}, lr=0.0, weight_decay=0.0,
                       clip_grad=1.0)
    torch.cuda.synchronize()
    assert abs(float(l2) - float(loss)) < 1e-6
    assert float(tr.grads.abs().max()) == 0.0          # cleared by the optimizer kernel
    return float(loss), grads.numpy()          # by value: tensors would travel as shared-memory handles of a process that exits


@pytest.mark.parametrize("reduce_dtype", ["fp32", "bf16"])
def test_two_rank_retrieval_step_with_gathered_negatives(reduce_dtype):
    """The reference's distributed SigLipLoss (modeling:239-297): each rank's loss = its own block with positives +
    every other rank's captions as negatives; DDP averages the gradients.  Driven through the trainer on 2 ranks
    (gloo, both on cuda:0) and compared with the oracle's autograd on the same split."""
    from oracle import train_oracle as TO
    from tests.helpers import run_ranks
    _dev()
    res = run_ranks(_gn_worker, 2, (reduce_dtype,))
    ret = {"loss0": res[0][0], "loss1": res[1][0], "grads": torch.from_numpy(res[0][1])}
    cfg = small_cfg(add_lora_spatial=True)
    tr, orc = _trainer_and_oracle(cfg, True, seed=8, lora=True)
    task, x, ti, _ = TO.schedule(cfg, B=4)[0]
    total = 0.0
    for rank in range(2):
        lo, hi = rank * 2, rank * 2 + 2
        olo, ohi = (1 - rank) * 2, (1 - rank) * 2 + 2
        want = orc.loss(task, x[lo:hi], {"kind": "retrieval", "text": ti["text"][lo:hi], "other_rank_text": [ti["text"][olo:ohi]]})
        assert abs(ret[f"loss{rank}"] - float(want)) < 2e-2 * abs(float(want)), (rank, ret[f"loss{rank}"], float(want))
        (want / 2).backward()                                   # DDP: mean over ranks
    og = orc.grads()
    got_all = ret["grads"]
    worst = 0.0
    for n, want in og.items():
        if want.numel() == 1 or float(want.abs().max()) == 0.0:
            continue
        e = tr._entry(n)
        got = got_all[e["offset"]: e["offset"] + e["numel"]].view(e["shape"])
        worst = max(worst, rel_l2(got, want))
    assert worst < (GRAD_REL_L2 if reduce_dtype == "fp32" else 6e-2), worst


def test_retrieval_loss_has_no_batch_limit():
    """B * Bt > 4096 used to fail (32 clips x 8 ranks).  64 local clips against 1024 gathered captions."""
    import streamformer_amd as sa
    from oracle import streamformer_oracle as O
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, T, D, Bt, rank = 64, 3, 768, 1024 + 64, 5
    pooler = torch.randn(B, T, D, generator=g)
    text = torch.randn(Bt, D, generator=g)
    ls, lb = torch.tensor(math.log(10.0)), torch.tensor(-2.0)
    p = pooler.clone().requires_grad_(True)
    lsr, lbr = ls.clone().requires_grad_(True), lb.clone().requires_grad_(True)
    others = [text[r * B:(r + 1) * B] for r in range(Bt // B) if r != rank]
    want = O.retrieval_loss(p, text[rank * B:(rank + 1) * B], lsr, lbr, other_rank_text=others)
    want.backward()
    loss, gp, gs = sa.heads.RetrievalHead(ls.to(dev), lb.to(dev)).loss(pooler.to(dev), text.to(dev), rank=rank)
    assert abs(float(loss) - float(want)) <= 1e-4 * abs(float(want))
    assert float((gp.cpu() - p.grad).abs().max()) <= 1e-5 * float(p.grad.abs().max()) + 1e-8
    assert abs(float(gs[0]) - float(lsr.grad)) <= 1e-4 * abs(float(lsr.grad)) and abs(float(gs[1]) - float(lbr.grad)) <= 1e-4 * abs(float(lbr.grad))
    with pytest.raises(ValueError):
        sa.heads.RetrievalHead().loss(pooler.to(dev), text[:32].to(dev), rank=1)


def _dp_worker(rank, world):
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    torch.cuda.set_device(0)
    cfg = small_cfg(add_lora_spatial=True)
    sd = make_state_dict(cfg, seed=8, lora=True)
    tr = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], freeze_spatial=True, device="cuda:0", bucket_mb=0.05)
    assert len(tr.buckets) > 1
    task, x, ti, _ = TO.schedule(cfg, B=4)[1]
    lo, hi = rank * 2, rank * 2 + 2
    ti_r = {"kind": "localization", "label_emb": ti["label_emb"].cuda(), "labels": ti["labels"][lo:hi].cuda()}
    _, pooler = tr.forward(x[lo:hi].cuda())
    _, gp, gs = tr.loss_and_grad(task, pooler, ti_r)
    tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
    tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
    tr.backward(gp, reduce=True)
    torch.cuda.synchronize()
    return (tr.grads / world).cpu().numpy()


def test_two_rank_allreduced_gradients_equal_the_big_batch():
    from oracle import train_oracle as TO
    from tests.helpers import run_ranks
    _dev()
    ret = {"grads": torch.from_numpy(run_ranks(_dp_worker, 2)[0])}
    cfg = small_cfg(add_lora_spatial=True)
    tr, _ = _trainer_and_oracle(cfg, True, seed=8, lora=True)
    task, x, ti, _ = TO.schedule(cfg, B=4)[1]
    dev = tr.device
    _, pooler = tr.forward(x.to(dev))
    _, gp, gs = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
    tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
    tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
    tr.backward(gp)
    torch.cuda.synchronize()
    # mean over 4 clips == average of the two ranks' means over 2 clips; only summation order differs
    assert rel_l2(ret["grads"], tr.grads.cpu()) < 2e-3


def _nccl_worker(rank, world, reduce_dtype):
    """World-size-1 RCCL: every collective of the N > 1 step is issued on the `nccl` backend (a 1-rank all-reduce / all-gather
    is the identity), next to a trainer that issues none."""
    import torch.distributed as dist
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.parallel import all_gather_rows, all_reduce_mean_
    from streamformer_amd.training import StreamformerTrainer
    assert dist.get_backend() == "nccl" and world == 1
    cfg = small_cfg(add_lora_spatial=True)
    sd = make_state_dict(cfg, seed=8, lora=True)
    kw = dict(freeze_spatial=True, device="cuda:0", bucket_mb=0.05, lr=1e-3)
    tr = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], grad_reduce_dtype=reduce_dtype, collectives_at_world_1=True, **kw)
    ref = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], **kw)          # world 1, no flag: no collective is issued
    assert tr._collectives and not ref._collectives and len(tr.buckets) > 1
    out = {}
    for idx in (0, 1):                      # retrieval (all-gathered captions) and localization
        task, x, ti, _ = TO.schedule(cfg, B=4)[idx]
        tin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ti.items()}
        grads = []
        for t in (tr, ref):
            t.zero_grad()
            _, pooler = t.forward(x.cuda())
            loss, gp, gs = t.loss_and_grad(task, pooler, tin)
            t.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
            t.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
            t.backward(gp, reduce=True)             # tr: bucketed async all-reduce + wait on RCCL; ref: plain backward
            torch.cuda.synchronize()
            grads.append((float(loss), t.grads.clone()))
        want = grads[1][1] if reduce_dtype == "fp32" else grads[1][1].to(torch.bfloat16).float()      # bf16 on the wire
        out[f"{task}_loss_equal"] = grads[0][0] == grads[1][0]
        out[f"{task}_grads_equal"] = bool(torch.equal(grads[0][1], want))
        out[f"{task}_grad_absmax"] = float(want.abs().max())
    for idx in (0, 1):                      # the whole enqueue path: micro_step = the above + clipped AdamW (after the gradient
        task, x, ti, _ = TO.schedule(cfg, B=4)[idx]      # comparisons: with bf16 on the wire the two trainers' weights part here)
        tin = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ti.items()}
        for t in (tr, ref):
            t.zero_grad()
            t.micro_step(task, x.cuda(), tin, clip_grad=1.0)
        torch.cuda.synchronize()
        out[f"{task}_params_maxdiff"] = float((tr.params - ref.params).abs().max())
    t = torch.randn(5, 7, device="cuda")
    out["gather_identity"] = bool(torch.equal(all_gather_rows(t, at_world_1=True), t))
    b = [torch.randn(1000, device="cuda"), torch.randn(3, device="cuda")]
    b0 = [v.clone() for v in b]
    all_reduce_mean_(b, at_world_1=True)
    out["allreduce_identity"] = all(bool(torch.equal(u, v)) for u, v in zip(b, b0))
    out["iso_ms"] = tr.time_bucket_allreduce(iters=2)
    return out


@pytest.mark.parametrize("reduce_dtype", ["fp32", "bf16"])
def test_rccl_world_size_1_step_equals_the_plain_step(reduce_dtype):
    """VERDICT r2 #1b: the `nccl` (= RCCL) branches of the training step — bucketed async all-reduce + wait (fp32 and bf16 wire
    format), the caption all-gather, `all_reduce_mean_` — executed on one GPU at world_size 1 and compared bit for bit with
    the step that issues no collective."""
    from tests.helpers import run_ranks
    _dev()
    r = run_ranks(_nccl_worker, 1, (reduce_dtype,), backend="nccl")[0]
    for task in ("retrieval", "localization"):
        assert r[f"{task}_loss_equal"] and r[f"{task}_grads_equal"] and r[f"{task}_grad_absmax"] > 0, r
        if reduce_dtype == "fp32":
            assert r[f"{task}_params_maxdiff"] == 0.0, r
        else:
            assert r[f"{task}_params_maxdiff"] < 2.5e-3, r        # <= two AdamW steps of lr 1e-3 on bf16-rounded gradients
    assert r["gather_identity"] and r["allreduce_identity"] and r["iso_ms"] > 0, r


def test_twenty_step_trajectory_tracks_the_oracle():
    """A short TRAINING RUN, not a single step: 20 optimizer steps (alternating tasks, update_freq 1, clipped gradients,
    lr / weight decay written per step from the cosine tables like tools/finetune_tools.py:406-410) on the HIP trainer and on
    oracle/train_oracle.py (torch autograd + torch.optim.AdamW, pinned to the reference's modules by fixture F8) from the same
    weights and batches.  bf16 operands perturb every step, so the two runs are compared as trajectories: per-step losses
    within 3 %, and the parameter DISPLACEMENT after 20 steps (what training did) agreeing in direction and size."""
    from oracle import train_oracle as TO
    from streamformer_amd.training import cosine_scheduler
    cfg = small_cfg(add_lora_spatial=True)
    tr, orc = _trainer_and_oracle(cfg, True, seed=8, lora=True, lr=1e-3, wd=0.05)
    dev = tr.device
    sched = TO.schedule(cfg, B=4)
    lrs = cosine_scheduler(1e-3, 1e-5, epochs=1, niter_per_ep=20, warmup_epochs=0.25)     # 5 warm-up steps
    wds = cosine_scheduler(0.05, 0.05, epochs=1, niter_per_ep=20)
    start = {k: v.detach().clone() for k, v in orc.named.items()}
    got, want = [], []
    for it in range(20):
        task, x, ti, _ = sched[it % 4]
        want_loss = orc.loss(task, x, ti)
        want_loss.backward()
        torch.nn.utils.clip_grad_norm_(list(orc.named.values()), 1.0)
        for g, wd in zip(orc.opt.param_groups, (wds[it], 0.0)):
            g["lr"], g["weight_decay"] = lrs[it], wd
        orc.opt.step()
        orc.opt.zero_grad(set_to_none=True)
        want.append(float(want_loss.detach()))
        got.append(float(tr.micro_step(task, x.to(dev), _to_dev(ti, dev), lr=lrs[it], weight_decay=wds[it], clip_grad=1.0)))
    rel = [abs(a - b) / abs(b) for a, b in zip(got, want)]
    assert max(rel) < 3e-2, list(zip(got, want))
    assert want[16] < want[0] and got[16] < got[0]                      # the run actually trains (same batch as step 0)
    sd = tr.state_dict()
    num = den = dot = 0.0
    for k, p0 in start.items():
        if p0.numel() < 64:
            continue
        dw_o = (orc.named[k].detach() - p0).double().flatten()
        dw_h = (sd[k].cpu() - p0).double().flatten()
        dot += float(dw_o @ dw_h); num += float(dw_h @ dw_h); den += float(dw_o @ dw_o)
    cos = dot / (num ** 0.5 * den ** 0.5)
    assert cos > 0.97 and 0.9 < (num / den) ** 0.5 < 1.1, (cos, (num / den) ** 0.5)


def test_checkpoint_resume_and_reference_optimizer_layout(golden_dir, tmp_path):
    """checkpoint-*.pth in the reference's layout (utils.py:608-636): wrapper-keyed weights + torch.optim.AdamW state.
    (1) the optimizer entry loads into a torch.optim.AdamW built the reference's way (optim_factory.py:59-104) over
    parameters enumerated in the REFERENCE's named_parameters() order (fixture F12), and every moment lands on the right
    tensor; (2) a fresh trainer resumed from the file continues bit-identically."""
    import json
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    cfg = small_cfg(add_lora_spatial=True)
    tr, _ = _trainer_and_oracle(cfg, True, seed=8, lora=True)
    dev = tr.device
    sched = TO.schedule(cfg, B=4)
    for it in range(3):
        task, x, ti, _ = sched[it % 4]
        tr.micro_step(task, x.to(dev), _to_dev(ti, dev))
    path = str(tmp_path / "checkpoint-2.pth")
    tr.save_checkpoint(path, epoch=2, args={"lr": 1e-3})
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert ck["epoch"] == 2 and set(ck) >= {"model", "optimizer", "epoch", "scaler", "args"}
    assert all(k.startswith(("timesformer.", "task_heads.")) or k in ("logit_scale", "logit_bias") for k in ck["model"])
    # (1) reference-side optimizer over the parameters of the reference WRAPPER in its named_parameters() order (F12
    # "wrapper_lora": its own logit_scale / logit_bias first, then timesformer.*, then the heads' deep copies)
    with open(os.path.join(golden_dir, "f12_param_order.json")) as f:
        rows = [r for r in json.load(f)["wrapper_lora"] if r[2]]
    ours = {"TaskRetrieval": "retrieval", "TaskLocalization": "localization"}

    def our_name(n):
        return ".".join(ours.get(part, part) for part in n.split("."))
    params = {}
    for n, shape, _ in rows:
        params[n] = torch.nn.Parameter(ck["model"][our_name(n)].clone() if our_name(n) in ck["model"] else torch.zeros(shape))
    groups = {}
    for n, p in params.items():
        g = "no_decay" if (p.dim() == 1 or n.endswith(".bias")) else "decay"
        groups.setdefault(g, {"weight_decay": 0.05 if g == "decay" else 0.0, "params": []})["params"].append(p)
    opt = torch.optim.AdamW(list(groups.values()), lr=1e-3)
    opt.load_state_dict(ck["optimizer"])
    assert [len(g["params"]) for g in opt.param_groups] == [len(g["params"]) for g in ck["optimizer"]["param_groups"]]
    assert opt.param_groups[0]["weight_decay"] == 0.05 and opt.param_groups[1]["weight_decay"] == 0.0
    want_steps = {"retrieval": 1.0, "localization": 2.0}            # schedule: retrieval, localization, localization
    for n, p in params.items():
        if n in ("logit_scale", "logit_bias"):
            assert p not in opt.state or not opt.state[p]           # the wrapper's unused scalars never get state
            continue
        key = our_name(n)
        key = key[len("timesformer."):] if key.startswith("timesformer.") else key
        e = tr._entry(key)
        sl = slice(e["offset"], e["offset"] + e["numel"])
        st = opt.state[p]
        assert float(st["step"]) == (want_steps[key.split(".")[1]] if key.startswith("task_heads.") else 3.0), n
        assert torch.equal(st["exp_avg"].flatten(), tr.exp_avg[sl].cpu()), n
        assert torch.equal(st["exp_avg_sq"].flatten(), tr.exp_avg_sq[sl].cpu()), n
    # (1b) the other direction: a state dict written by that torch optimizer after one more step in which only the
    # retrieval head and the encoder had gradients (torch skips grad-None parameters) loads into a trainer
    for n, p in params.items():
        if n.startswith("timesformer.") or "TaskRetrieval" in n:
            p.grad = torch.full_like(p, 1e-3)
    opt.step()
    tr3 = StreamformerTrainer(cfg, make_state_dict(cfg, seed=8, lora=True), ["retrieval", "localization"], freeze_spatial=True, device=dev)
    tr3.load_optimizer_state_dict(opt.state_dict())
    assert tr3.step_count == 4 and tr3.head_steps == {"retrieval": 2, "localization": 2}
    pq = params["timesformer.encoder.layer.1.intermediate.dense.weight"]
    e = tr3._entry("encoder.layer.1.intermediate.dense.weight")
    assert torch.equal(opt.state[pq]["exp_avg"].flatten(), tr3.exp_avg[e["offset"]: e["offset"] + e["numel"]].cpu())
    # (2) exact resume
    tr2 = StreamformerTrainer(cfg, make_state_dict(cfg, seed=99, lora=True), ["retrieval", "localization"], freeze_spatial=True,
                              device=dev, lr=5e-4, weight_decay=0.01)
    assert tr2.load_checkpoint(path) == 2 and tr2.step_count == 3 and tr2.lr == 1e-3 and tr2.weight_decay == 0.05
    assert tr2.head_steps == tr.head_steps == {"retrieval": 1, "localization": 2}
    task, x, ti, _ = sched[3]
    la = tr.micro_step(task, x.to(dev), _to_dev(ti, dev))
    lb = tr2.micro_step(task, x.to(dev), _to_dev(ti, dev))
    assert float(la) == float(lb)
    assert torch.equal(tr.params, tr2.params) and torch.equal(tr.exp_avg, tr2.exp_avg) and torch.equal(tr.exp_avg_sq, tr2.exp_avg_sq)


@pytest.mark.parametrize("task_idx", [0, 1])
def test_drop_path_gradients_match_the_oracle_on_the_same_draw(task_idx):
    """VERDICT r2 missing #4: stochastic depth in the training step (modeling:460-486, 846-856: layer i drops each residual
    branch per sample with rate linspace(0, drop_path_rate, L)[i]; sample = dim-0 entry of the tensor the branch returns).
    The trainer draws the keep / drop factors per forward from a seeded generator; the oracle's autograd replays the same
    draw.  Also: the draw changes per forward, an evaluation forward (drop_path off) ignores it, dropout still raises."""
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    cfg = small_cfg(add_lora_spatial=True, drop_path_rate=0.4, num_hidden_layers=3)
    sd = make_state_dict(cfg, seed=8, lora=True)
    dev = _dev()
    tr = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], freeze_spatial=True, device=dev, drop_path_seed=5)
    orc = TO.OracleTrainer(sd, cfg, ["retrieval", "localization"], freeze_spatial=True)
    task, x, ti, _ = TO.schedule(cfg, B=4)[task_idx]
    _, pooler = tr.forward(x.to(dev))
    dp = tr.last_drop_path
    assert dp is not None and tuple(dp.shape) == (3, 4 * 9 + 4 * 4 + 4)
    assert float(dp[0].min()) == 1.0 and float(dp[0].max()) == 1.0                 # layer 0: rate 0
    assert {round(v, 4) for v in torch.unique(dp[2]).tolist()} <= {0.0, round(1.0 / 0.6, 4)} and float(dp[2].min()) == 0.0      # layer 2: rate 0.4
    loss, gp, gs = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
    tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
    tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
    tr.backward(gp)
    torch.cuda.synchronize()
    want_loss = orc.loss(task, x, ti, drop_path=dp)
    want_loss.backward()
    assert abs(float(loss) - float(want_loss)) < 2e-2 * abs(float(want_loss))
    _compare_grads(tr, orc, scalar_rel=0.2)       # gate gradients: the same cancelling sum over 40 % fewer live rows (measured 12 %)
    plain = float(orc.loss(task, x, ti).detach())
    assert abs(plain - float(want_loss)) > 1e-4 * abs(plain)                      # the draw really changes the forward
    _, p2 = tr.forward(x.to(dev))
    assert not torch.equal(tr.last_drop_path, dp)                                  # a new draw per forward
    tr.drop_path = False
    _, p3 = tr.forward(x.to(dev))
    want = orc.loss(task, x, ti)
    l3, _, _ = tr.loss_and_grad(task, p3, _to_dev(ti, dev))
    assert tr.last_drop_path is None and abs(float(l3) - float(want.detach())) < 2e-2 * abs(float(want.detach()))


@pytest.mark.parametrize("task_idx,hid,att,dpr", [(0, 0.2, 0.0, 0.0), (1, 0.15, 0.0, 0.3), (0, 0.0, 0.2, 0.0), (1, 0.1, 0.1, 0.0)])
def test_dropout_gradients_match_the_oracle_on_the_same_masks(task_idx, hid, att, dpr):
    """VERDICT r3 missing #2: config.hidden_dropout_prob / attention_probs_dropout_prob in the training step, at the reference's
    sites (modeling:374, 378 embeddings; 752, 761 SelfOutput; 822, 835 MLP; 556, 603, 669, 705 attention probabilities).  Masks are
    counter-based — a function of (seed, site, element index) — so the oracle's autograd replays exactly the draw of the forward;
    also together with drop_path.  A new seed per forward; `dropout = False` gives the plain forward."""
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    cfg = small_cfg(add_lora_spatial=True, hidden_dropout_prob=hid, attention_probs_dropout_prob=att, drop_path_rate=dpr, num_hidden_layers=3)
    sd = make_state_dict(cfg, seed=8, lora=True)
    dev = _dev()
    tr = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], freeze_spatial=True, device=dev, drop_path_seed=7)
    orc = TO.OracleTrainer(sd, cfg, ["retrieval", "localization"], freeze_spatial=True)
    task, x, ti, _ = TO.schedule(cfg, B=4)[task_idx]
    _, pooler = tr.forward(x.to(dev))
    drop = tr.last_dropout
    assert drop is not None and drop[1:] == (hid, att)
    loss, gp, gs = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
    tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
    tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
    tr.backward(gp)
    torch.cuda.synchronize()
    want_loss = orc.loss(task, x, ti, drop_path=tr.last_drop_path, dropout=drop)
    want_loss.backward()
    assert abs(float(loss) - float(want_loss)) < 2e-2 * abs(float(want_loss))
    _compare_grads(tr, orc, scalar_rel=0.2)
    plain = float(orc.loss(task, x, ti).detach())
    assert abs(plain - float(want_loss)) > 1e-4 * abs(plain)                      # the masks really change the forward
    tr.forward(x.to(dev))
    assert tr.last_dropout[0] != drop[0]                                           # a new seed per forward
    tr.dropout = False
    tr.drop_path = False
    _, p3 = tr.forward(x.to(dev))
    l3, _, _ = tr.loss_and_grad(task, p3, _to_dev(ti, dev))
    assert tr.last_dropout is None and abs(float(l3) - plain) < 2e-2 * abs(plain)


def test_localization_batch_with_per_clip_dataset_tables():
    """VERDICT r4 missing #4: the reference head walks the batch clip by clip with each clip's OWN dataset table — tables of different
    sizes in one batch (modeling:2250-2276).  The fast trainer groups the clips by dataset (one loss launch per table, weighted by
    the group's share) and scatters the gradients back: loss, d loss / d pooler_output and d loss / d (logit_scale, logit_bias)
    against the oracle's per-clip restatement, and the whole micro-step's parameter gradients against its autograd."""
    from oracle import streamformer_oracle as O
    cfg = small_cfg(add_lora_spatial=True)
    tr, orc = _trainer_and_oracle(cfg, True, seed=8, lora=True)
    dev = tr.device
    g = torch.Generator().manual_seed(77)
    B, T, D = 5, 16, cfg.hidden_size
    names = ["thumos", "tvseries", "thumos", "ek100", "tvseries"]
    sizes = {"thumos": 21, "tvseries": 31, "ek100": 7}
    tables = {}
    for k, L in sizes.items():
        e = torch.randn(L, D, generator=g)
        tables[k] = e / e.norm(dim=-1, keepdim=True)
    labels = torch.stack([torch.randint(-1, sizes[n], (T,), generator=g) for n in names])
    x = torch.randn(B, T, 3, 48, 48, generator=g)
    # ---- the head alone on a fixed pooler_output -------------------------------------------------------------------
    pooler = torch.randn(B, T, D, generator=g)
    ls, lb = torch.tensor(math.log(10.0), requires_grad=True), torch.tensor(-2.0, requires_grad=True)
    pw = pooler.clone().requires_grad_(True)
    want = sum(O.localization_loss(pw[i:i + 1], tables[n], labels[i:i + 1], ls, lb) for i, n in enumerate(names)) / B
    want.backward()
    ti = {"kind": "localization", "datasets": names, "label_embs": {k: v.to(dev) for k, v in tables.items()}, "labels": labels.to(dev)}
    loss, gp, gs = tr.loss_and_grad("localization", pooler.to(dev), ti)
    assert abs(float(loss) - float(want)) <= 2e-5 * abs(float(want))
    assert float((gp.cpu() - pw.grad).abs().max()) <= 1e-6
    assert abs(float(gs[0]) - float(ls.grad)) <= 2e-5 * abs(float(ls.grad)) + 1e-6 and abs(float(gs[1]) - float(lb.grad)) <= 2e-5 * abs(float(lb.grad)) + 1e-6
    # ---- one whole micro-step: every parameter gradient against the oracle's autograd through the same per-clip loss ------------
    out = O.forward_graph(orc.sd, cfg, x)
    h = orc.heads["localization"]
    want2 = sum(O.localization_loss(out["pooler_output"][i:i + 1], tables[n], labels[i:i + 1], h["logit_scale"], h["logit_bias"])
                for i, n in enumerate(names)) / B
    want2.backward()
    tr.zero_grad()
    l2 = tr.micro_step("localization", x.to(dev), ti, update_freq=2)       # update_freq 2: gradients stay in the buffer (scaled by 1/2)
    assert abs(float(l2) - float(want2)) < 2e-2 * abs(float(want2))
    tr.grads.mul_(2.0)
    _compare_grads(tr, orc)


def test_trainer_rejects_what_it_cannot_do():
    import streamformer_amd._native as nat
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    dev = _dev()
    cfg = small_cfg(add_lora_spatial=True)
    sd = make_state_dict(cfg, seed=8, lora=True)
    tr = StreamformerTrainer(cfg, sd, ["retrieval"], device=dev)
    with pytest.raises(ValueError):                       # wrong resolution
        tr.forward(torch.zeros(1, 4, 3, 64, 64, device=dev))
    with pytest.raises(nat.NativeError):                  # more frames than time-embedding rows
        tr.forward(torch.zeros(1, 17, 3, 48, 48, device=dev))
    with pytest.raises(RuntimeError):                     # backward without a forward
        tr.backward(torch.zeros(1, 4, cfg.hidden_size, device=dev))
    with pytest.raises(KeyError):                         # frozen parameters have no gradient slot
        tr.grad("encoder.layer.0.attention.attention.qkv.weight")
    sd2 = dict(sd)
    sd2.pop("head.probe")
    with pytest.raises(KeyError):
        StreamformerTrainer(cfg, sd2, ["retrieval"], device=dev)
    # attention-probability dropout exists for <= 224 patches per frame and <= 16 frames: refused at construction
    big = small_cfg(add_lora_spatial=True, image_size=256, attention_probs_dropout_prob=0.1)       # 16 x 16 = 256 patches
    with pytest.raises(NotImplementedError):
        StreamformerTrainer(big, make_state_dict(big, seed=8, lora=True), ["retrieval"], device=dev)
    # state_dict round trip keeps the reference key names and the head scalars
    out = tr.state_dict()
    assert {k for k in sd if not k.endswith(".mask")} <= set(out) and "task_heads.retrieval.logit_scale" in out   # masks: unused buffers
    assert abs(float(out["task_heads.retrieval.logit_scale"]) - math.log(10.0)) < 1e-6
    for k in ("embeddings.position_embeddings", "encoder.layer.1.output.dense.weight"):
        assert torch.equal(out[k].cpu(), sd[k].float())


def test_nonfinite_step_is_skipped_on_the_device_and_reported():
    """tools/finetune_tools.py:533-541 stops the run on a non-finite loss, utils.py:515-551 skips the step on inf gradients.
    The HIP step is sync-free, so the AdamW kernel makes the check itself: a poisoned micro-step leaves parameters and
    moments bit-identical, clears the gradients, raises the sticky flag; check_finite() / checkpoint() then raise."""
    from oracle import train_oracle as TO
    cfg = small_cfg(add_lora_spatial=True)
    tr, _ = _trainer_and_oracle(cfg, True, seed=8, lora=True, lr=1e-3, wd=0.05)
    dev = tr.device
    task, x, ti, _ = TO.schedule(cfg, B=2)[0]
    other, xo, tio, _ = TO.schedule(cfg, B=2)[1]
    assert other != task
    tr.micro_step(other, xo.to(dev), _to_dev(tio, dev))        # the other head takes one clean step of its own
    tr.micro_step(task, x.to(dev), _to_dev(ti, dev))
    tr.check_finite()                                          # a finite step raises nothing
    assert tr.nonfinite_steps() == 0
    p0, m0, v0 = tr.params.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone()
    bad = x.clone()
    bad[0, 0, 0, 0, 0] = float("nan")
    loss = tr.micro_step(task, bad.to(dev), _to_dev(ti, dev))
    assert not math.isfinite(float(loss))
    assert torch.equal(tr.params, p0) and torch.equal(tr.exp_avg, m0) and torch.equal(tr.exp_avg_sq, v0)
    assert float(tr.grads.abs().max()) == 0.0                  # cleared although the update was skipped
    assert tr.nonfinite_steps() == 1
    with pytest.raises(FloatingPointError):
        tr.check_finite()
    with pytest.raises(FloatingPointError):
        tr.checkpoint()
    # acknowledging the skipped step takes it back out of the host-side counts, and an optimizer_step() with no
    # micro_step() in between does not re-check the stale NaN loss
    before, hb = tr.step_count, dict(tr.head_steps)
    assert tr.reset_nonfinite() == 1 and tr.step_count == before - 1 and tr.nonfinite_steps() == 0
    # only the head the skipped step involved is rewound; the other head's AdamW step count (bias correction) stays
    assert tr.head_steps[task] == hb[task] - 1 and tr.head_steps[other] == hb[other] == 1
    tr.check_finite()
    tr.optimizer_step()
    assert tr.nonfinite_steps() == 0 and tr._last_loss is None
    # a poisoned FIRST micro-step of an accumulation window is still seen by the step that closes the window
    tr.zero_grad()
    tr.micro_step(task, bad.to(dev), _to_dev(ti, dev), update_freq=2)
    tr.grads.zero_()                                           # finite gradients, non-finite loss sum
    tr.micro_step(task, x.to(dev), _to_dev(ti, dev), update_freq=2)
    assert tr.nonfinite_steps() == 1
    tr.reset_nonfinite()
    # inf gradients with a finite loss (the GradScaler case): poison the gradient buffer directly
    tr2, _ = _trainer_and_oracle(cfg, True, seed=8, lora=True)
    tr2.forward(x.to(dev))
    tr2.grads[5] = float("inf")
    q0 = tr2.params.clone()
    tr2.optimizer_step()
    assert torch.equal(tr2.params, q0) and tr2.nonfinite_steps() == 1
    # the guard can be switched off (then the NaN goes into the weights, as plain torch would let it)
    tr3 = TO_trainer_without_guard(cfg)
    tr3.forward(x.to(dev))
    tr3.grads[5] = float("inf")
    tr3.optimizer_step()
    assert tr3.nonfinite_steps() == 0 and not bool(torch.isfinite(tr3.params[:tr3.n_train]).all())


def TO_trainer_without_guard(cfg):
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    return StreamformerTrainer(cfg, make_state_dict(cfg, seed=8, lora=True), ["retrieval", "localization"], freeze_spatial=True,
                               device=_dev(), nonfinite_guard=False)


def _task_mismatch_worker(rank, world):
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    torch.cuda.set_device(0)
    cfg = small_cfg(add_lora_spatial=True)
    tr = StreamformerTrainer(cfg, make_state_dict(cfg, seed=8, lora=True), ["retrieval", "localization"], freeze_spatial=True, device="cuda:0")
    sched = TO.schedule(cfg, B=2)
    task, x, ti, _ = sched[rank % 2]                     # rank 0: retrieval, rank 1: localization
    try:
        tr.micro_step(task, x.cuda(), {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ti.items()})
    except RuntimeError as e:
        return "refused: " + str(e)[:60]
    return "ran"


def test_ranks_that_schedule_different_tasks_are_refused_not_hung():
    """sampler.py:218-337 gives every rank the same task per micro-step; a retrieval rank issues the caption all-gather, a
    localization rank does not, so a mismatch would hang in a collective.  The trainer checks the first micro-step."""
    from tests.helpers import run_ranks
    _dev()
    res = run_ranks(_task_mismatch_worker, 2)
    assert all(r.startswith("refused") for r in res), res


def test_training_reduces_the_loss_on_a_fixed_batch():
    """End-to-end sanity of forward + backward + AdamW + weight refresh: 40 optimizer steps on one fixed batch
    drive both task losses down (the oracle does the same in tests of its own trajectory, fixture F8)."""
    from oracle import train_oracle as TO
    cfg = small_cfg(add_lora_spatial=True)
    tr, _ = _trainer_and_oracle(cfg, True, seed=8, lora=True, lr=2e-3, wd=0.0)
    dev = tr.device
    sched = TO.schedule(cfg, B=4)
    batches = [(sched[0][0], sched[0][1].to(dev), _to_dev(sched[0][2], dev)), (sched[1][0], sched[1][1].to(dev), _to_dev(sched[1][2], dev))]
    first, last = {}, {}
    for it in range(40):
        task, x, ti = batches[it % 2]
        loss = float(tr.micro_step(task, x, ti))
        assert math.isfinite(loss)
        first.setdefault(task, loss)
        last[task] = loss
    assert last["retrieval"] < 0.5 * first["retrieval"], (first, last)
    assert last["localization"] < 0.7 * first["localization"], (first, last)
    assert float(tr.grad_norm()) == 0.0          # gradients cleared after the optimizer step


def test_grad_norm_and_clipping_follow_torch():
    """grad_norm() == torch's global L2 norm over the trainable gradients; clip_grad scales the AdamW input exactly
    like torch.nn.utils.clip_grad_norm_ (coefficient max_norm / (norm + 1e-6), only when norm > max_norm)."""
    from oracle import train_oracle as TO
    cfg = small_cfg(add_lora_spatial=True)
    task, x, ti, _ = TO.schedule(cfg)[0]
    results = []
    for clip in (None, 0.05):
        tr, _ = _trainer_and_oracle(cfg, True, seed=8, lora=True, lr=1e-3, wd=0.0)
        dev = tr.device
        _, pooler = tr.forward(x.to(dev))
        _, gp, gs = tr.loss_and_grad(task, pooler, _to_dev(ti, dev))
        tr.grad(f"task_heads.{task}.logit_scale").add_(gs[0])
        tr.grad(f"task_heads.{task}.logit_bias").add_(gs[1])
        tr.backward(gp)
        g = tr.grads.clone()
        norm = float(tr.grad_norm())
        assert abs(norm - float(g.double().norm())) < 1e-4 * norm and norm > 0.05
        p0, m0 = tr.params[: tr.n_train].clone(), None
        tr.optimizer_step(clip_grad=clip)
        # first Adam step from zero moments: exp_avg = (1 - beta1) * coef * g
        coef = 1.0 if clip is None else min(1.0, clip / (norm + 1e-6))
        assert rel_max(tr.exp_avg, 0.1 * coef * g) < 1e-5
        results.append((p0 - tr.params[: tr.n_train]).abs().max().item())
    assert results[0] > 0 and results[1] > 0


def test_full_size_training_step_properties():
    """BASELINE configs[2] size (SigLIP-base, LoRA recipe, 8 clips x 16 x 224^2): too big for the CPU oracle, so
    size-independent properties: finite, bit-reproducible, and batch additivity of the mean loss — the gradient of
    8 clips equals the average of the gradients of the two halves (different tile plans, same mathematics)."""
    from streamformer_amd.configuration import siglip_base
    from streamformer_amd.init_weights import make_state_dict
    from streamformer_amd.training import StreamformerTrainer
    dev = _dev()
    cfg = siglip_base(add_lora_spatial=True)
    sd = make_state_dict(cfg, seed=0, lora=True)
    tr = StreamformerTrainer(cfg, sd, ["localization"], freeze_spatial=True, device=dev)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(8, 16, 3, 224, 224, generator=g).to(dev)
    lab = torch.randn(20, 768, generator=g)
    lab = (lab / lab.norm(dim=-1, keepdim=True)).to(dev)
    labels = torch.randint(-1, 20, (8, 16), generator=g).to(dev)

    def grads_of(sl):
        tr.zero_grad()
        _, pooler = tr.forward(x[sl])
        loss, gp, gs = tr.loss_and_grad("localization", pooler, {"kind": "localization", "label_emb": lab, "labels": labels[sl]})
        tr.grad("task_heads.localization.logit_scale").add_(gs[0])
        tr.grad("task_heads.localization.logit_bias").add_(gs[1])
        tr.backward(gp)
        torch.cuda.synchronize()
        return float(loss), tr.grads.clone()

    l8, g8 = grads_of(slice(0, 8))
    l8b, g8b = grads_of(slice(0, 8))
    assert math.isfinite(l8) and bool(torch.isfinite(g8).all())
    assert l8 == l8b and torch.equal(g8, g8b)
    la, ga = grads_of(slice(0, 4))
    lb, gb = grads_of(slice(4, 8))
    assert abs(l8 - 0.5 * (la + lb)) < 1e-3 * abs(l8)
    avg = 0.5 * (ga + gb)
    assert rel_l2(g8, avg) < 2e-2 and cosine(g8, avg) > 0.9995
