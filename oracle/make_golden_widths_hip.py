"""F15: a head_dim-72 configuration at a width the HIP path takes (hidden_size a multiple of 64), from the REFERENCE run here (needs
/root/reference; never travels to the GPU box).  Test infrastructure only: nothing in the product path imports this.

SigLIP-so400m is 1152 / 16 heads = head_dim 72, intermediate 4304 (not a multiple of 64), 14 x 14 patches (C P P = 588, not a multiple
of 8) — the three things round 6 made the HIP path accept (generic-width attention + pooling head, zero-padded MLP and patch weights,
generic patch extraction).  The fixture keeps all three at a size that stays small in git:
  hd72w   hidden 576, 8 heads (head_dim 72), intermediate 1072 (= 16 x 67), patch 14 on 42 x 42 pixels (N = 9), 2 layers, 8 frames
Outputs of the reference class on seeded frames + the oracle checked against them; streaming (the vqa_enc variant) is covered on the HIP
side by streamed == full clip, which F4 pins for the oracle.

    python oracle/make_golden_widths_hip.py        # writes tests/golden/f15_hd72_hip.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import streamformer_oracle as O  # noqa: E402
from oracle.make_golden import OUT, build_ref, check, frames, import_reference  # noqa: E402
from streamformer_amd.configuration import StreamformerConfig  # noqa: E402
from streamformer_amd.init_weights import make_state_dict, state_dict_sha256  # noqa: E402

CASES = {
    "hd72w": dict(image_size=42, patch_size=14, num_frames=8, hidden_size=576, num_hidden_layers=2, num_attention_heads=8,
                  intermediate_size=1072, enable_causal_temporal=True),
}


def main():
    ref_models = import_reference()
    torch.manual_seed(0)
    out = {}
    for case, (tag, kw) in enumerate(CASES.items()):
        cfg = StreamformerConfig(**kw)
        sd = make_state_dict(cfg, seed=15)
        m = build_ref(ref_models, cfg, sd)
        x = frames(150 + case, (2, cfg.num_frames, 3, cfg.image_size, cfg.image_size))
        with torch.no_grad():
            r = m(x)
        o = O.forward(sd, cfg, x)
        print(f"F15 {tag}: hidden {cfg.hidden_size}, heads {cfg.num_attention_heads}, intermediate {cfg.intermediate_size}, patch {cfg.patch_size}")
        check(f"{tag} last_hidden_state", o["last_hidden_state"], r.last_hidden_state)
        check(f"{tag} pooler_output", o["pooler_output"], r.pooler_output)
        out[f"{tag}_last_hidden_state"] = r.last_hidden_state.numpy()
        out[f"{tag}_pooler_output"] = r.pooler_output.numpy()
        out[f"{tag}_sha"] = np.frombuffer(state_dict_sha256(sd).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "f15_hd72_hip.npz"), **out)
    print("wrote", os.path.join(OUT, "f15_hd72_hip.npz"))


if __name__ == "__main__":
    main()
