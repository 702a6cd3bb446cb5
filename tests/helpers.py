"""Shared test helpers: seeded inputs identical to oracle/make_golden.py, small configs."""
import numpy as np
import torch

from streamformer_amd.configuration import StreamformerConfig


def small_cfg(**kw):
    base = dict(image_size=48, patch_size=16, num_frames=16, hidden_size=128, num_hidden_layers=2,
                num_attention_heads=2, intermediate_size=256, enable_causal_temporal=True)
    base.update(kw)
    return StreamformerConfig(**base)


def frames(seed, shape, clamp=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    return x.clamp_(-1, 1) if clamp else x


def maxabs(a, b):
    a = torch.as_tensor(np.asarray(a)) if not torch.is_tensor(a) else a
    b = torch.as_tensor(np.asarray(b)) if not torch.is_tensor(b) else b
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def load_npz(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}
