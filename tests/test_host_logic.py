"""CPU: host-side mirror of the reference module — config, checkpoint formats, key handling, output
objects, and the loud failure without a GPU."""
import json
import os

import pytest
import torch

import streamformer_amd as sa
from streamformer_amd.configuration import StreamformerConfig
from streamformer_amd.modeling import ModelOutput, expected_keys, normalize_checkpoint_keys
from streamformer_amd.parallel import shard_range
from tests.helpers import small_cfg


def test_config_defaults_and_roundtrip(tmp_path):
    c = StreamformerConfig()
    assert (c.image_size, c.patch_size, c.num_frames, c.hidden_size, c.num_hidden_layers, c.num_attention_heads,
            c.intermediate_size, c.hidden_act, c.layer_norm_eps, c.qkv_bias, c.attention_type) == \
        (224, 16, 16, 768, 12, 12, 3072, "gelu", 1e-6, True, "divided_space_time")
    assert c.enable_causal_temporal is False and c.add_lora_spatial is False and c.model_type == "timesformer"
    c2 = StreamformerConfig(enable_causal_temporal=True, architectures=["X"], torch_dtype="float32")
    c2.save_pretrained(tmp_path)
    d = json.load(open(tmp_path / "config.json"))
    assert d["model_type"] == "timesformer" and d["architectures"] == ["X"]
    c3 = StreamformerConfig.from_pretrained(str(tmp_path))
    assert c3.to_dict() == c2.to_dict()
    with pytest.raises(ValueError):
        StreamformerConfig(attention_type="bogus")


def test_state_dict_keys_match_reference_layout():
    cfg = sa.siglip_base()
    sd = sa.make_state_dict(cfg, seed=0)
    exp = expected_keys(cfg)
    extra = set(sd) - set(exp)
    assert all(k.endswith("temporal_attention.attention.mask") for k in extra) and len(extra) == 12
    assert set(exp) <= set(sd)
    for k, shape in exp.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg)
    assert m.num_parameters() == 128350476              # reference count
    cfg_l = sa.siglip_base(add_lora_spatial=True)
    n_l = sa.TimesformerMultiTaskingModelSigLIP(cfg_l).num_parameters()
    assert n_l - 128350476 == 12 * 32 * (768 + 2304 + 768 + 768)


@pytest.mark.parametrize("safe", [True, False])
def test_save_and_from_pretrained_roundtrip(tmp_path, safe):
    cfg = small_cfg(add_lora_spatial=True)
    sd = sa.make_state_dict(cfg, seed=9)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg)
    m.load_state_dict(sd)
    m.save_pretrained(str(tmp_path), safe_serialization=safe)
    m2 = sa.TimesformerMultiTaskingModelSigLIP.from_pretrained(str(tmp_path), device="cpu")
    assert m2.config.add_lora_spatial and m2.config.hidden_size == 128
    a, b = m.state_dict(), m2.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_wrapper_checkpoint_keys_are_normalised():
    cfg = small_cfg()
    sd = sa.make_state_dict(cfg, seed=2)
    wrapped = {"timesformer." + k: v for k, v in sd.items()}
    wrapped["task_heads.retrieval.logit_scale"] = torch.tensor(1.0)
    wrapped["logit_bias"] = torch.tensor(-2.0)
    clean = normalize_checkpoint_keys(wrapped)
    assert set(clean) == set(sd)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg)
    m.load_state_dict(wrapped)                           # strict: nothing missing / unexpected
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in sd.items() if "probe" not in k})
    bad = dict(sd)
    bad["head.probe"] = torch.zeros(1, 1, 7)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)


def test_lora_surface():
    m = sa.TimesformerMultiTaskingModelSigLIP(small_cfg())
    n0 = len(m.state_dict())
    m.add_lora_spatial()
    sd = m.state_dict()
    assert len(sd) == n0 + 4 * 2
    assert all(float(v.abs().max()) == 0 for k, v in sd.items() if "_lora_b" in k)    # B = 0 (modeling:534)
    assert "encoder.layer.0.attention.attention.qkv.weight" not in m.trainable_parameter_names()
    assert "encoder.layer.0.attention.attention.qkv_lora_a.weight" in m.trainable_parameter_names()


def test_module_protocol_without_a_gpu():
    """The encoder is a torch.nn.Module with the reference's parameter names: a wrapper holds it (modeling:1362), its
    state_dict carries the prefix, dtype moves with .to(), requires_grad follows frozen_spatial / add_lora_spatial."""
    cfg = small_cfg()
    sd = sa.make_state_dict(cfg, seed=3)

    class Wrapper(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.timesformer = sa.TimesformerMultiTaskingModelSigLIP(cfg)

    w = Wrapper()
    assert isinstance(w.timesformer, torch.nn.Module)
    assert set(w.state_dict()) == {"timesformer." + k for k in sd}             # masks included, like the reference
    w.load_state_dict({"timesformer." + k: v for k, v in sd.items()})
    m = w.timesformer
    assert torch.equal(m.encoder.layer[1].temporal_attention_gating.data, sd["encoder.layer.1.temporal_attention_gating"])
    assert torch.equal(m.head.attention.out_proj.weight.data, sd["head.attention.out_proj.weight"])
    assert len(m.encoder.layer) == cfg.num_hidden_layers and m.encoder.layer[-1] is m.encoder.layer[cfg.num_hidden_layers - 1]
    assert next(m.parameters()).device.type == "cpu" and m.dtype == torch.float32
    w.to(torch.bfloat16)
    assert m.dtype == torch.bfloat16 and next(w.parameters()).dtype == torch.bfloat16
    m.frozen_spatial()
    assert not m.encoder.layer[0].attention.attention.qkv.weight.requires_grad
    assert m.encoder.layer[0].attention.output.dense.weight.requires_grad
    m.add_lora_spatial()
    assert m.encoder.layer[0].attention.attention.qkv_lora_a.weight.dtype == torch.bfloat16        # new factors follow the module
    assert not m.encoder.layer[0].attention.output.dense.weight.requires_grad
    m.requires_grad_(False)
    assert m.trainable_parameter_names() == []
    tower = sa.TimesformerVisionTower(m, streaming_mode=True, context_length=4)
    assert (tower.hidden_size, tower.num_patches, tower.num_patches_per_side, tower.image_size) == (128, 9, 3, 48)
    assert tower.device.type == "cpu" and tower.dtype == torch.bfloat16 and tower.is_loaded
    with pytest.raises(OSError, match="could not be fetched from the hub"):      # no network here: the hub id is tried (huggingface_hub) and the failure names the fix
        sa.TimesformerVisionTower("Go2Heart/StreamFormer-timesformer-siglip")
    # the module copies and pickles like any nn.Module (native state is rebuilt from the parameters on first use)
    import copy, io
    m2 = copy.deepcopy(m)
    assert m2 is not m and m2.encoder.layer[0]._root is m2 and m2.head._root is m2
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3.encoder.layer[1]._root is m3 and set(m3.state_dict()) == set(m.state_dict())
    res = m3.load_state_dict(m.state_dict())
    assert res.missing_keys == [] and res.unexpected_keys == [] and tuple(res) == ([], [])


def test_model_output_protocol():
    o = ModelOutput(last_hidden_state=1, pooler_output=2, hidden_states=None, attentions=None)
    assert o.last_hidden_state == 1 and o["pooler_output"] == 2 and o[0] == 1 and o[1] == 2
    assert o.to_tuple() == (1, 2)
    with pytest.raises(AttributeError):
        o.nope


def test_forward_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = sa.TimesformerMultiTaskingModelSigLIP(small_cfg())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 3, 48, 48))
    with pytest.raises(NotImplementedError):
        sa.TimesformerMultiTaskingModelSigLIP(small_cfg(attention_type="joint_space_time"))


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_siglip_weight_surgery_key_mapping():
    from streamformer_amd.convert import siglip_vision_to_streamformer
    cfg = small_cfg()
    D, I, N, P = cfg.hidden_size, cfg.intermediate_size, cfg.num_patches, cfg.patch_size
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    src = {"vision_model.embeddings.patch_embedding.weight": r(D, 3, P, P), "vision_model.embeddings.patch_embedding.bias": r(D),
           "vision_model.embeddings.position_embedding.weight": r(N, D), "vision_model.post_layernorm.weight": r(D),
           "vision_model.post_layernorm.bias": r(D), "vision_model.head.probe": r(1, 1, D),
           "vision_model.head.attention.in_proj_weight": r(3 * D, D), "vision_model.head.attention.in_proj_bias": r(3 * D),
           "vision_model.head.attention.out_proj.weight": r(D, D), "vision_model.head.attention.out_proj.bias": r(D),
           "vision_model.head.layernorm.weight": r(D), "vision_model.head.layernorm.bias": r(D),
           "vision_model.head.mlp.fc1.weight": r(I, D), "vision_model.head.mlp.fc1.bias": r(I),
           "vision_model.head.mlp.fc2.weight": r(D, I), "vision_model.head.mlp.fc2.bias": r(D),
           "text_model.whatever": r(3), "logit_scale": r(1)}
    for i in range(cfg.num_hidden_layers):
        p = f"vision_model.encoder.layers.{i}."
        for n_, shp in (("self_attn.q_proj", (D, D)), ("self_attn.k_proj", (D, D)), ("self_attn.v_proj", (D, D)),
                        ("self_attn.out_proj", (D, D)), ("mlp.fc1", (I, D)), ("mlp.fc2", (D, I))):
            src[p + n_ + ".weight"] = r(*shp)
            src[p + n_ + ".bias"] = r(shp[0])
        for n_ in ("layer_norm1", "layer_norm2"):
            src[p + n_ + ".weight"] = r(D)
            src[p + n_ + ".bias"] = r(D)
    sd = siglip_vision_to_streamformer(src, cfg)
    assert set(sd) == set(expected_keys(cfg))
    q = src["vision_model.encoder.layers.1.self_attn.q_proj.weight"]
    assert torch.equal(sd["encoder.layer.1.attention.attention.qkv.weight"][:D], q)
    assert torch.equal(sd["encoder.layer.1.attention.attention.qkv.bias"][2 * D:], src["vision_model.encoder.layers.1.self_attn.v_proj.bias"])
    assert torch.equal(sd["encoder.layer.0.layernorm_after.weight"], src["vision_model.encoder.layers.0.layer_norm2.weight"])
    assert torch.equal(sd["encoder.layer.0.output.dense.weight"], src["vision_model.encoder.layers.0.mlp.fc2.weight"])
    assert sd["embeddings.position_embeddings"].shape == (1, N, D) and float(sd["encoder.layer.0.temporal_attention_gating"]) == 0
    # gate 0 + zero time embeddings: the converted video model is per-frame SigLIP (oracle check, CPU)
    from oracle import streamformer_oracle as O
    x = torch.randn(1, 3, 3, 48, 48, generator=g).double()
    sd64 = O.cast_state_dict(sd, torch.float64)          # fp64: the unit-variance test weights blow fp32 noise up
    a = O.forward(sd64, cfg, x)["last_hidden_state"]
    b = torch.cat([O.forward(sd64, cfg, x[:, t:t + 1])["last_hidden_state"] for t in range(3)], 1)
    assert float((a - b).abs().max()) < 1e-9
    sa.TimesformerMultiTaskingModelSigLIP(cfg).load_state_dict(sd)


def test_window_starts_match_reference_formula():
    from streamformer_amd.features import window_starts
    import numpy as np
    for n in (6, 13, 100, 481):
        w = window_starts(n)
        assert len(w) == n // 6 and w[0] == 0 and (np.diff(w) >= 0).all()
        assert (w[-1] == n) == (n // 6 > 1)        # with >1 windows the last start is n itself -> clamped to the final 6 frames


def test_resize_center_crop_follows_the_reference_transform():
    """Resize(224, 'bilinear') on numpy frames is cv2.INTER_LINEAR — no antialiasing on downscale, long side
    int(size * W / H) (functional.py:26-75) — and CenterCrop offsets are int(round(./2)) (video_transforms.py:1158-1159)."""
    import numpy as np
    from streamformer_amd.features import resize_center_crop, resize_sizes
    assert resize_sizes(480, 854, 224) == (224, 398)               # int(), not round(): 398.53 -> 398
    assert resize_sizes(854, 480, 224) == (398, 224)
    assert resize_sizes(224, 224, 224) == (224, 224)
    rng = np.random.default_rng(3)
    # downscale of a 2-pixel checkerboard: plain bilinear sampling keeps contrast, an antialiasing filter would flatten it
    H, W = 96, 160
    yy, xx = np.mgrid[0:H, 0:W]
    board = (((yy // 2 + xx // 2) % 2) * 255).astype(np.uint8)
    f = np.repeat(board[None, :, :, None], 3, axis=3)
    out = resize_center_crop(f, 48)
    assert out.shape == (1, 3, 48, 48) and out.dtype == torch.uint8
    assert float(out.float().std()) > 60.0
    # the same numbers as a direct evaluation of the bilinear formula at half-pixel centres
    g = rng.integers(0, 256, (2, 60, 90, 3), dtype=np.uint8)
    nh, nw = resize_sizes(60, 90, 48)
    got = resize_center_crop(g, 48)
    sy = (np.arange(nh) + 0.5) * (60 / nh) - 0.5
    sx = (np.arange(nw) + 0.5) * (90 / nw) - 0.5
    y0 = np.clip(np.floor(sy), 0, 59).astype(int); x0 = np.clip(np.floor(sx), 0, 89).astype(int)
    y1 = np.minimum(y0 + 1, 59); x1 = np.minimum(x0 + 1, 89)
    fy = np.clip(sy - np.floor(sy), 0, 1) * (sy >= 0); fx = np.clip(sx - np.floor(sx), 0, 1) * (sx >= 0)
    a = g.astype(np.float64)
    ref = ((a[:, y0][:, :, x0] * (1 - fx)[None, None, :, None] + a[:, y0][:, :, x1] * fx[None, None, :, None]) * (1 - fy)[None, :, None, None]
           + (a[:, y1][:, :, x0] * (1 - fx)[None, None, :, None] + a[:, y1][:, :, x1] * fx[None, None, :, None]) * fy[None, :, None, None])
    top, left = int(round((nh - 48) / 2.0)), int(round((nw - 48) / 2.0))
    ref = np.clip(np.rint(ref), 0, 255)[:, top:top + 48, left:left + 48]
    assert np.abs(got.permute(0, 2, 3, 1).numpy().astype(np.int32) - ref.astype(np.int32)).max() <= 1
    # already at the minimal size: untouched apart from the crop (functional.py:31-33)
    h = rng.integers(0, 256, (1, 48, 70, 3), dtype=np.uint8)
    assert np.array_equal(resize_center_crop(h, 48)[0].permute(1, 2, 0).numpy(), h[0][:, 11:59])


def test_feature_cli_reads_reference_style_checkpoints(tmp_path):
    """the reference's save_model pickles its argparse.Namespace under "args" (utils.py:608-636); the
    weights-only load of --ckpt_path must accept exactly that."""
    import argparse
    path = str(tmp_path / "checkpoint-3.pth")
    torch.save({"model": {"timesformer.x": torch.ones(2)}, "epoch": 3, "args": argparse.Namespace(lr=1e-4, tasks=["a"])}, path)
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=True)
    with torch.serialization.safe_globals([argparse.Namespace]):
        ck = torch.load(path, map_location="cpu", weights_only=True)
    assert ck["args"].lr == 1e-4 and "timesformer.x" in ck["model"]
    import inspect
    from streamformer_amd import features
    assert "safe_globals([argparse.Namespace])" in inspect.getsource(features.main)


def test_drop_path_factors_follow_the_reference_rule():
    """modeling:846-856 (layer i: linspace(0, rate, L)[i]) and :460-486 (floor(keep + rand) / keep per dim-0 entry)."""
    from streamformer_amd.training import drop_path_factors
    g = torch.Generator().manual_seed(3)
    f = drop_path_factors(0.5, 5, B=64, T=4, N=9, generator=g)
    assert tuple(f.shape) == (5, 64 * 9 + 64 * 4 + 64)
    assert torch.all(f[0] == 1.0)                                         # first layer never drops
    for i, r in enumerate(torch.linspace(0, 0.5, 5).tolist()[1:], start=1):
        keep = 1.0 - r
        vals = set(round(v, 5) for v in torch.unique(f[i]).tolist())
        assert vals <= {0.0, round(1.0 / keep, 5)}
        assert abs(float((f[i] > 0).float().mean()) - keep) < 0.08        # 896 draws per layer
        assert abs(float(f[i].mean()) - 1.0) < 0.15                       # unbiased
    g2 = torch.Generator().manual_seed(3)
    assert torch.equal(f, drop_path_factors(0.5, 5, 64, 4, 9, g2))         # replayable from the seed
    assert torch.all(drop_path_factors(0.0, 3, 2, 2, 2) == 1.0)


# ---- training host logic (streamformer_amd/training.py; reference utils.py:574-605, run_finetuning_multi_task.py:386) ----
def test_cosine_scheduler_table():
    import math
    from streamformer_amd.training import cosine_scheduler, scaled_lr
    s = cosine_scheduler(1.0, 0.1, epochs=4, niter_per_ep=10, warmup_epochs=1, start_warmup_value=0.0)
    assert len(s) == 40
    assert s[0] == 0.0 and abs(s[9] - 1.0) < 1e-12                 # linspace includes both ends
    assert all(b > a for a, b in zip(s[:9], s[1:10]))
    assert abs(s[10] - 1.0) < 1e-12                                  # cosine part starts at the base value
    assert abs(s[25] - (0.1 + 0.45 * (1 + math.cos(math.pi * 15 / 30)))) < 1e-12
    assert s[-1] > 0.1 and all(b <= a + 1e-15 for a, b in zip(s[10:], s[11:]))
    flat = cosine_scheduler(0.05, 0.05, 2, 5)                        # weight-decay table of the recipe: constant
    assert flat == [0.05] * 10
    assert abs(scaled_lr(2e-5, 8, 1, 8) - 2e-5 * 64 / 256) < 1e-18


def test_allreduce_bucket_plan():
    from streamformer_amd.training import bucket_ranges
    # stage slices as the library lays them out: head last in the buffer, finished first
    stages = [(900, 100)] + [(100 + 200 * i, 200) for i in (3, 2, 1, 0)] + [(0, 100)]
    b = bucket_ranges(stages, 250)
    assert [x[0] for x in b] == [1, 3, 5]
    covered = sorted((off, off + n) for _, off, n in b)
    assert covered[0][0] == 0 and covered[-1][1] == 1000
    assert all(a[1] == c[0] for a, c in zip(covered, covered[1:]))      # contiguous, disjoint, complete
    one = bucket_ranges(stages, 10**9)
    assert one == [(5, 0, 1000)]


def test_trainer_refuses_to_run_without_a_gpu():
    """No CPU fallback on the training path either: constructing the trainer without an AMD GPU raises."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    import pytest
    from streamformer_amd.configuration import StreamformerConfig
    from streamformer_amd.training import StreamformerTrainer
    cfg = StreamformerConfig(image_size=48, patch_size=16, num_frames=16, hidden_size=128, num_hidden_layers=2,
                             num_attention_heads=2, intermediate_size=256)
    with pytest.raises(RuntimeError, match="GPU"):
        StreamformerTrainer(cfg, {}, ["retrieval"])


def test_parameter_and_state_dict_order_match_the_reference(golden_dir):
    """F12: optimizer state ids, DDP buckets and checkpoint-*.pth optimizer entries follow named_parameters() order."""
    import json
    import streamformer_amd as sa
    with open(os.path.join(golden_dir, "f12_param_order.json")) as f:
        ref = json.load(f)
    for key, lora in (("plain", False), ("lora", True)):
        m = sa.TimesformerMultiTaskingModelSigLIP(small_cfg(add_lora_spatial=lora))
        assert [(n, list(p.shape)) for n, p in m.named_parameters()] == [(r[0], r[1]) for r in ref[key]]
        assert list(m.state_dict().keys()) == ref[key + "_state_dict_keys"]
