"""F13: head widths other than 64, from the REFERENCE run here (needs /root/reference; never travels to the GPU box).

    python oracle/make_golden_widths.py        # writes tests/golden/f13_widths.npz, asserts oracle == reference

The HIP path refuses head_dim != 64 at construction; the reference's config schema
(models/configuration_streamformer.py:90-135) takes any hidden_size / num_attention_heads, e.g. the SigLIP-so400m shape
1152 / 16 = 72.  These fixtures pin the ORACLE on two such widths (test infrastructure for the kernels a later round writes):
  so400m-shaped tiny  hidden 144, 2 heads (head_dim 72), intermediate 304, patch 14 on 42 x 42 pixels (N = 9), 2 layers
  narrow heads        hidden 128, 4 heads (head_dim 32), intermediate 256, patch 16 on 48 x 48
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


if __name__ == "__main__":
    main()
