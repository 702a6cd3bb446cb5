"""F13 / F14: configurations the HIP path refuses, from the REFERENCE run here (needs /root/reference; never travels to the GPU box).

    python oracle/make_golden_variants.py      # writes tests/golden/f13_widths.npz and f14_attention_types.npz, asserts oracle == reference

The HIP path refuses head_dim != 64 at construction; the reference's config schema
(models/configuration_streamformer.py:90-135) takes any hidden_size / num_attention_heads, e.g. the SigLIP-so400m shape
1152 / 16 = 72.  These fixtures pin the ORACLE on two such widths (test infrastructure for the kernels a later round writes):
  so400m-shaped tiny  hidden 144, 2 heads (head_dim 72), intermediate 304, patch 14 on 42 x 42 pixels (N = 9), 2 layers
  narrow heads        hidden 128, 4 heads (head_dim 32), intermediate 256, patch 16 on 48 x 48
F14: the two attention_type values StreamFormer never instantiates (modeling:914-933): space_only (frame-major embeddings without time
embedding, attention inside a frame — and the reference's tail then reads the result as patch-major, reproduced as is) and
joint_space_time (attention over all tokens of a clip), small config, T = 4 and T = 3 of num_frames = 4.
Inputs are regenerated from seeds; the stored tensors are the reference's own outputs."""
from __future__ import annotations

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
    "hd72": dict(image_size=42, patch_size=14, num_frames=8, hidden_size=144, num_hidden_layers=2, num_attention_heads=2,
                 intermediate_size=304, enable_causal_temporal=True),
    "hd32": dict(image_size=48, patch_size=16, num_frames=8, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                 intermediate_size=256, enable_causal_temporal=True),
}


def main():
    ref_models = import_reference()
    torch.manual_seed(0)
    out = {}
    for case, (tag, kw) in enumerate(CASES.items()):
        cfg = StreamformerConfig(**kw)
        sd = make_state_dict(cfg, seed=13)
        m = build_ref(ref_models, cfg, sd)
        x = frames(130 + case, (2, cfg.num_frames, 3, cfg.image_size, cfg.image_size))
        with torch.no_grad():
            r = m(x)
        lhs, pool = r.last_hidden_state, r.pooler_output
        o = O.forward(sd, cfg, x)
        print(f"F13 {tag}: hidden {cfg.hidden_size}, heads {cfg.num_attention_heads}")
        check(f"{tag} last_hidden_state", o["last_hidden_state"], lhs)          # both [B, T, N, D]
        check(f"{tag} pooler_output", o["pooler_output"], pool)                # both [B, T, D]
        out[f"{tag}_last_hidden_state"] = lhs.numpy()
        out[f"{tag}_pooler_output"] = pool.numpy()
        out[f"{tag}_sha"] = np.frombuffer(state_dict_sha256(sd).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "f13_widths.npz"), **out)
    print("wrote", os.path.join(OUT, "f13_widths.npz"))

    from oracle.make_golden import small_cfg
    f14 = {}
    for at in ("space_only", "joint_space_time"):
        cfg = small_cfg(attention_type=at, num_frames=4)
        sd = make_state_dict(cfg, seed=14)
        m = build_ref(ref_models, cfg, sd)
        f14[f"{at}_sha"] = np.frombuffer(state_dict_sha256(sd).encode(), dtype=np.uint8)
        for T in (4, 3):
            x = frames(140 + T, (2, T, 3, 48, 48))
            with torch.no_grad():
                r = m(x)
            o = O.forward(sd, cfg, x)
            print(f"F14 {at} T={T}")
            check(f"{at} T={T} last_hidden_state", o["last_hidden_state"], r.last_hidden_state)
            check(f"{at} T={T} pooler_output", o["pooler_output"], r.pooler_output)
            f14[f"{at}_T{T}_last_hidden_state"] = r.last_hidden_state.numpy()
            f14[f"{at}_T{T}_pooler_output"] = r.pooler_output.numpy()
    np.savez_compressed(os.path.join(OUT, "f14_attention_types.npz"), **f14)
    print("wrote", os.path.join(OUT, "f14_attention_types.npz"))


if __name__ == "__main__":
    main()
