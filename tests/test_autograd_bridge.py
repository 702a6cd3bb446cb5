"""The drop-in module under torch autograd (SURVEY.md §8b "callers that must keep working":
the wrapper's training forward, reference modeling:1486-1523, under loss.backward() + torch.optim,
tools/finetune_tools.py:560-570).  The encoder's forward / backward run in the HIP library behind one autograd node;
gradients are checked against the CPU oracle's autograd and the three-step trajectory against fixture F8 (made by the
reference's own modules + torch.optim.AdamW)."""
import math
import os

import numpy as np
import pytest
import torch

from tests.helpers import load_npz, small_cfg


def rel_l2(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    return float((got - want).norm() / (want.norm() + 1e-30))


def cosine(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def test_module_is_born_in_eval_mode_and_refuses_cpu_training():
    import streamformer_amd as sa
    m = sa.TimesformerMultiTaskingModelSigLIP(small_cfg())
    assert not m.training                       # like a from_pretrained() model; .train() opts into the autograd path
    m.train()
    with pytest.raises(RuntimeError):           # no CPU / eager fallback on the autograd path either
        m(torch.zeros(1, 4, 3, 48, 48))
    w = sa.StreamformerForMultiTaskingSigLIP(small_cfg(add_lora_spatial=True), {"TaskRetrieval": {}, "TaskLocalization": {}})
    assert w.training and w.timesformer.training
    w.prepare_for_multi_tasks()
    w.frozen_spatial()
    names = [n for n, p in w.named_parameters() if p.requires_grad]
    assert names[:2] == ["logit_scale", "logit_bias"]          # wrapper scalars first (modeling:1363-1364)
    assert not any(".attention.attention.qkv.weight" in n and "temporal" not in n for n in names)
    assert any("qkv_lora_a" in n for n in names) and "task_heads.TaskRetrieval.logit_scale" in names
    with pytest.raises(NotImplementedError):
        sa.StreamformerForMultiTaskingSigLIP(small_cfg(), {"SSV2": {}})


@pytest.mark.gpu
@pytest.mark.parametrize("lora", [False, True])
def test_backward_fills_parameter_grads_like_the_oracle(lora):
    """model.train(); out = model(x); loss(out.pooler_output, out.last_hidden_state).backward() -> .grad of every
    trainable nn.Parameter vs the oracle's autograd on the same loss."""
    import streamformer_amd as sa
    from oracle import streamformer_oracle as O
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg = small_cfg(add_lora_spatial=lora)
    sd = sa.make_state_dict(cfg, seed=8, lora=lora)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg)
    m.load_state_dict(sd)
    if lora:
        m.add_lora_spatial()
    m.cuda().train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 3, 48, 48, generator=g)
    wp = torch.randn(2, 4, cfg.hidden_size, generator=g)
    wl = torch.randn(2, 4, 9, cfg.hidden_size, generator=g) * 0.1

    def loss_of(out, wp, wl):
        return (out["pooler_output"] * wp).sum() + (out["last_hidden_state"] * wl).sum()
    out = m(x.cuda())
    assert out.pooler_output.grad_fn is not None and out.last_hidden_state.grad_fn is not None
    loss = loss_of(out, wp.cuda(), wl.cuda())
    loss.backward()
    torch.cuda.synchronize()
    osd = {k: v.clone().requires_grad_(m._named[k].requires_grad) for k, v in sd.items() if not k.endswith(".mask")}
    want = loss_of(O.forward_graph(osd, cfg, x), wp, wl)
    want.backward()
    assert abs(float(loss) - float(want)) < 2e-2 * abs(float(want)) + 1e-2
    worst = 0.0
    for k, p in m._named.items():
        if not p.requires_grad:
            assert p.grad is None and osd[k].grad is None, k
            continue
        assert p.grad is not None, k
        wg = osd[k].grad
        if wg.numel() == 1 or float(wg.abs().max()) < 1e-6:
            continue
        r = rel_l2(p.grad, wg)
        worst = max(worst, r)
        assert r < 5e-2 and cosine(p.grad, wg) > 0.998, (k, r)
    # a second backward accumulates (update_freq > 1), a pooler-only loss leaves d last_hidden_state undefined
    g1 = {k: p.grad.clone() for k, p in m._named.items() if p.grad is not None}
    out = m(x.cuda())
    (out.pooler_output * wp.cuda()).sum().backward()
    k0 = "encoder.layer.0.intermediate.dense.weight"
    assert float((m._named[k0].grad - g1[k0]).abs().max()) > 0
    # a stale graph is refused instead of reading overwritten activations
    o1 = m(x.cuda())
    m(x.cuda())
    with pytest.raises(RuntimeError):
        o1.pooler_output.sum().backward()
    # eval mode / no_grad keep the inference path: no grad_fn
    with torch.no_grad():
        assert m(x.cuda()).pooler_output.grad_fn is None
    m.eval()
    assert m(x.cuda()).pooler_output.grad_fn is None


@pytest.mark.gpu
def test_wrapper_with_torch_adamw_follows_fixture_f8(golden_dir):
    """The reference wrapper pattern (modeling:1486-1523) + torch.optim.AdamW with the optim_factory grouping: the 4
    micro-batches / 3 optimizer steps of fixture F8, losses and parameter updates vs the reference's trajectory."""
    import streamformer_amd as sa
    from oracle import train_oracle as TO
    from streamformer_amd.init_weights import state_dict_sha256
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    f8 = load_npz(os.path.join(golden_dir, "f8_train.npz"))
    cfg = small_cfg(add_lora_spatial=True)
    sd = sa.make_state_dict(cfg, seed=8, lora=True)
    if state_dict_sha256(sd) != str(f8["sha256"]):
        pytest.skip("seeded weights differ from the fixture's (RNG drift on this box)")
    model = sa.StreamformerForMultiTaskingSigLIP(cfg, {"TaskRetrieval": {}, "TaskLocalization": {"label2id": {"synthetic": {}}}})
    model.timesformer.load_state_dict(sd)
    model.prepare_for_multi_tasks()
    model.frozen_spatial()
    model.cuda().train()
    lr, wd = float(f8["lr"]), float(f8["wd"])
    decay, no_decay = [], []
    for n, p in model.named_parameters():          # optim_factory.py:59-104
        if not p.requires_grad:
            continue
        (no_decay if (p.dim() == 1 or n.endswith(".bias")) else decay).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": wd}, {"params": no_decay, "weight_decay": 0.0}], lr=lr)
    tname = {"retrieval": "TaskRetrieval", "localization": "TaskLocalization"}
    init = {k: v.detach().clone() for k, v in model.timesformer.state_dict().items()}
    losses, micro = [], 0
    for task, x, ti, uf in TO.schedule(cfg):
        if ti["kind"] == "retrieval":
            tin = {"text_features": ti["text"].cuda()}
        else:
            model.task_heads["TaskLocalization"].set_label_embeddings("synthetic", ti["label_emb"].cuda())
            tin = {"dataset": ["synthetic"] * x.shape[0], "label": ti["labels"].cuda()}
        ls, _ = model(x.cuda(), multi_task_input={"task_name": tname[task], "task_input": tin})
        loss = ls[tname[task]]
        losses.append(float(loss))
        (loss / uf).backward()                      # finetune_tools.py:560-570
        micro += 1
        if micro % uf == 0:
            opt.step()
            opt.zero_grad(set_to_none=True)
    assert np.allclose(losses, f8["losses"], rtol=3e-2), (losses, f8["losses"])
    after = {k: v.detach() for k, v in model.timesformer.state_dict().items()}
    for t, n in tname.items():
        after[f"task_heads.{t}.logit_scale"] = model.task_heads[n].logit_scale.detach()
        after[f"task_heads.{t}.logit_bias"] = model.task_heads[n].logit_bias.detach()
        init[f"task_heads.{t}.logit_scale"] = torch.tensor(math.log(10.0))
        init[f"task_heads.{t}.logit_bias"] = torch.tensor(-2.0)
    checked = 0
    for k in f8:
        if not k.startswith("param/"):
            continue
        n = k[len("param/"):]
        want = torch.from_numpy(f8[k]).double()
        got = after[n].cpu().double().reshape(want.shape)
        assert float((got - want).abs().max()) <= 6 * lr, n
        if "grad0/" + n in f8:
            g0 = torch.from_numpy(f8["grad0/" + n]).abs()
            sig = g0 > 0.05 * g0.max()
            if int(sig.sum()) > 8:
                i0 = init[n].cpu().double().reshape(want.shape)
                upd_w, upd_g = (want - i0)[sig], (got - i0)[sig]
                assert float((upd_w - upd_g).norm() / (upd_w.norm() + 1e-30)) < 0.15, n
                checked += 1
    assert checked > 5
    # the wrapper's own scalars never receive a gradient (only the heads' deep copies are used, modeling:2203-2204)
    assert model.logit_scale.grad is None and float(model.logit_scale) == pytest.approx(math.log(10.0))
