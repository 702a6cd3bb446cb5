"""The tile GEMM family of one to a few clips per call (sf_gemm_tile.hip; 2560 < M <= SF_TILE_MAX_M): every candidate tile
shape forced in turn (lab switch SF_TILE_SHAPE, read once per process -> each case runs in its own interpreter) on ragged
row counts, every epilogue sf_op_linear reaches (plain fp32, residual read-modify-write, erf-GELU bf16), and the whole
SigLIP-base forward of ONE clip (README.md:55-71: LayerNorm-folded consumers, embedding epilogue, KV-cache row remap of the
temporal qkv) against the reference's outputs (fixture F2)."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

PROBE = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["SF_ROOT"])
import streamformer_amd as sa
from tests.test_hip_parity import _linear
from tests.helpers import load_npz, maxabs
worst = 0.0
for (M, N, K) in [(3136, 768, 768), (2999, 768, 3072), (4100, 2304, 768), (2600, 3072, 256), (3136, 1536, 768)]:
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
    ref = x.bfloat16().double() @ w.bfloat16().double().t() + b.double()
    e1 = maxabs(_linear(sa, x, w, b, None, 1.0, False, 0), ref)
    e2 = maxabs(_linear(sa, x, w, b, r, 0.37, False, 0), r.double() + 0.37 * ref)
    e3 = maxabs(_linear(sa, x, w, b, None, 1.0, True, 0), torch.nn.functional.gelu(ref))
    assert e1 <= 1e-4 and e2 <= 1e-4 and e3 <= 2e-2 + 1e-4, (M, N, K, e1, e2, e3)
    worst = max(worst, e1, e2)
f2 = load_npz(os.path.join(os.environ["SF_ROOT"], "tests", "golden", "f2_base.npz"))
cfg = sa.siglip_base()
sd = sa.make_state_dict(cfg, seed=0)
if sa.state_dict_sha256(sd) == str(f2["sha256"]):
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16"); m.load_state_dict(sd); m.cuda().eval()
    torch.manual_seed(0)
    x = torch.randn(1, 16, 3, 224, 224)
    out = m(x.cuda())
    d = maxabs(out.pooler_output, f2["randn_pooler_output"])
    a = m(x.cuda()).pooler_output
    assert torch.equal(a, out.pooler_output)          # bit-reproducible
    assert d <= 3e-2, d
    print("F2 pooler max-abs", d)
print("OK worst", worst)
'''


@pytest.mark.parametrize("shape", ["auto", "0", "1", "2", "3", "4", "5"])
def test_tile_gemm_shapes(shape):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["SF_ROOT"] = ROOT
    env["SF_TILE_MAX_M"] = "6272"
    if shape != "auto":
        env["SF_TILE_SHAPE"] = shape
    r = subprocess.run([sys.executable, "-c", PROBE], env=env, cwd=ROOT, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0 and "OK worst" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


TRAIN_PROBE = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["SF_ROOT"])
import streamformer_amd as sa
from streamformer_amd.training import StreamformerTrainer
lora = os.environ.get("SF_PROBE_LORA", "1") == "1"
cfg = sa.siglip_base(add_lora_spatial=lora)
sd = sa.make_state_dict(cfg, seed=3, lora=lora)
tr = StreamformerTrainer(cfg, sd, ["localization"], freeze_spatial=lora, device="cuda:0")
g = torch.Generator().manual_seed(11)
x = torch.randn(1, 16, 3, 224, 224, generator=g).cuda()
lab = torch.randn(7, cfg.hidden_size, generator=g); lab = (lab / lab.norm(dim=-1, keepdim=True)).cuda()
labels = torch.randint(-1, 7, (1, 16), generator=g).cuda()
_, pooler = tr.forward(x)
loss, gp, gs = tr.loss_and_grad("localization", pooler, {"kind": "localization", "label_emb": lab, "labels": labels})
tr.backward(gp)
torch.cuda.synchronize()
names = ["encoder.layer.0.temporal_attention.attention.qkv.weight", "encoder.layer.5.intermediate.dense.weight",
         "encoder.layer.11.output.dense.weight", "encoder.layer.11.output.dense.bias", "encoder.layer.7.temporal_dense.weight",
         "encoder.layer.7.temporal_dense.bias", "encoder.layer.7.temporal_attention_gating", "encoder.layer.2.temporal_attention.output.dense.weight",
         "encoder.layer.2.temporal_attention.attention.qkv.bias", "embeddings.position_embeddings"]
names += (["encoder.layer.3.attention.attention.qkv_lora_b.weight"] if lora else
          ["encoder.layer.3.attention.attention.qkv.weight", "encoder.layer.3.attention.output.dense.weight",
           "encoder.layer.3.attention.output.dense.bias"])
out = {"loss": float(loss), "pooler": pooler.double().norm().item()}
torch.save({n: tr.grad(n).cpu() for n in names}, os.environ["SF_OUT"])
print("RESULT " + json.dumps(out))
'''


def test_training_step_of_one_clip_is_the_same_on_the_tile_kernels(tmp_path):
    """One SigLIP-base clip (M = 3136) through the training step: its forward / input-gradient GEMMs land on the tile kernels;
    the same step with the family switched off (panel / 256^2 kernels) must give the same loss and gradients up to summation
    order."""
    import json
    import torch
    res = {}
    for tag, off in (("tile", False), ("plain", True)):
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        env["SF_ROOT"] = ROOT
        env["SF_OUT"] = str(tmp_path / f"{tag}.pt")
        if off:
            env["SF_DISABLE_GEMM_TILE"] = "1"
        r = subprocess.run([sys.executable, "-c", TRAIN_PROBE], env=env, cwd=ROOT, capture_output=True, text=True, timeout=560)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0]
        res[tag] = (json.loads(line[7:]), torch.load(env["SF_OUT"]))
    a, b = res["tile"], res["plain"]
    assert abs(a[0]["loss"] - b[0]["loss"]) <= 2e-3 * abs(b[0]["loss"])
    for n in a[1]:
        rel = float((a[1][n].double() - b[1][n].double()).norm() / (b[1][n].double().norm() + 1e-30))
        assert rel < 2e-2, (n, rel)


@pytest.mark.parametrize("lora", [True, False])
def test_grouped_weight_gradients_equal_the_per_projection_launches(tmp_path, lora):
    """backward_layer batches the weight gradients of a layer's Linears into one launch (sf_launch_wgrad_group: shared tile
    numbering, partial and bias offsets per job).  SF_WGRAD_UNGROUPED=1 launches every projection by itself; both must give
    the same gradients up to the order of the token-split sums — with LoRA on the frozen spatial block (5 jobs per layer)
    and with every Linear trained (7 jobs)."""
    import json
    import torch
    res = {}
    for tag, off in (("grouped", False), ("single", True)):
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        env["SF_ROOT"] = ROOT
        env["SF_OUT"] = str(tmp_path / f"{tag}.pt")
        env["SF_PROBE_LORA"] = "1" if lora else "0"
        if off:
            env["SF_WGRAD_UNGROUPED"] = "1"
        r = subprocess.run([sys.executable, "-c", TRAIN_PROBE], env=env, cwd=ROOT, capture_output=True, text=True, timeout=560)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0]
        res[tag] = (json.loads(line[7:]), torch.load(env["SF_OUT"]))
    a, b = res["grouped"], res["single"]
    assert a[0]["loss"] == b[0]["loss"]            # the forward is the same code
    for n in a[1]:
        rel = float((a[1][n].double() - b[1][n].double()).norm() / (b[1][n].double().norm() + 1e-30))
        # the fused temporal projections (round 5) derive dW_dense / dW_out / dgate from G1 = g^T ctx through D x D GEMMs with bf16
        # operands: a last-bit difference of G1's token-split sums can flip the bf16 rounding of a few of its elements
        fused = any(k in n for k in ("temporal_dense", "temporal_attention.output.dense", "temporal_attention_gating"))
        # (the gate is a 0-dim parameter whose gradient is a cancelling sum over D x D products: 1e-3)
        assert rel < ((1e-3 if "gating" in n else 2e-4) if fused else 1e-5), (n, rel)
