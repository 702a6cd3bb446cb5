"""Fixture F11: the SigLIP -> StreamFormer weight surgery (reference tools/initialize_SigLIP_weights.py:25-264) pinned on
what it promises — a converted encoder (temporal gate 0, time embeddings 0) reproduces HF's ``SiglipVisionModel`` frame by
frame.  The reference's own tool needs hub weights inside ``main()`` and cannot run offline, so the pin is the HF model
itself (transformers, a third-party dependency of the reference) on seeded random weights of the same architecture.
Build container only:

    python oracle/make_golden_siglip.py     # writes tests/golden/f11_siglip_surgery.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import streamformer_oracle as O  # noqa: E402
from oracle.siglip_fixture import make_siglip_state_dict, siglip_cfg_kwargs  # noqa: E402
from streamformer_amd.configuration import StreamformerConfig  # noqa: E402
from streamformer_amd.convert import siglip_vision_to_streamformer  # noqa: E402


def fixture_cfg():
    # SigLIP's own activation (tanh-GELU): also the only fixture that drives hidden_act code 1 of the library
    return StreamformerConfig(image_size=48, patch_size=16, num_frames=4, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                              intermediate_size=256, hidden_act="gelu_pytorch_tanh", enable_causal_temporal=True)


def main():
    from transformers import SiglipVisionConfig, SiglipVisionModel
    cfg = fixture_cfg()
    ssd = make_siglip_state_dict(cfg, seed=11)
    hf = SiglipVisionModel(SiglipVisionConfig(**siglip_cfg_kwargs(cfg))).eval()
    hf_keys = set(hf.state_dict())
    # transformers 4.52 (the reference's pin) names the tower "vision_model.*" inside SiglipModel; 5.x flattens SiglipVisionModel
    res = hf.load_state_dict({(k if k in hf_keys else k[len("vision_model."):]): v for k, v in ssd.items()}, strict=True)
    print("HF load:", res)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 3, 48, 48, generator=g)              # [B, T, C, H, W]
    with torch.no_grad():
        out = hf(pixel_values=x.reshape(6, 3, 48, 48))
    lhs = out.last_hidden_state.reshape(2, 3, 9, 128)
    pool = out.pooler_output.reshape(2, 3, 128)
    sd = siglip_vision_to_streamformer(ssd, cfg, seed=0)
    mine = O.forward(sd, cfg, x)
    d1 = float((mine["last_hidden_state"] - lhs).abs().max())
    d2 = float((mine["pooler_output"] - pool).abs().max())
    print(f"oracle(converted) vs HF SiglipVisionModel: last_hidden_state {d1:.2e}, pooler_output {d2:.2e}")
    assert d1 < 2e-5 and d2 < 2e-5
    path = os.path.join(ROOT, "tests", "golden", "f11_siglip_surgery.npz")
    np.savez_compressed(path, last_hidden_state=lhs.numpy(), pooler_output=pool.numpy(), pixel_values=x.numpy())
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
