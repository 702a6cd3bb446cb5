"""Feature-extraction harness over the encoder.

* ``sliding_window_features``: the online-action-detection dump of ``extract_oad_feature.py:34-35,
  122-136`` — 6-frame windows starting at ``np.linspace(0, n, n // 6).astype(int)``, the last window
  clamped to the final 6 frames, one pooled 768-d vector per window via
  ``forward_features(pooling_method="last")`` (``modeling:1525-1536``); output float32
  ``[num_windows, D]`` (what ``np.save`` writes for ``downstream/OAD``).  Windows are batched: they are
  independent clips, so one forward carries many of them.
* ``long_video_features``: per-frame pooled features of an arbitrarily long video, cut into
  ``config.num_frames`` clips inside 384-frame windows like ``extract_feature`` (``modeling:1551-1621``),
  zero-padded at the tail and trimmed back.
"""
from __future__ import annotations

import numpy as np
import torch


def window_starts(num_frames: int, window: int = 6) -> np.ndarray:
    return np.linspace(0, num_frames, num_frames // window).astype(int)     # extract_oad_feature.py:34-35


@torch.no_grad()
def sliding_window_features(model, frames: torch.Tensor, window: int = 6, batch_windows: int = 64) -> np.ndarray:
    """frames: [n, 3, H, W] (already resampled / normalised) -> float32 [num_windows, D]."""
    n = frames.shape[0]
    clips = []
    for s in window_starts(n, window):
        s = int(s)
        clips.append(frames[n - window:] if s + window > n else frames[s:s + window])
    feats = []
    for i in range(0, len(clips), batch_windows):
        x = torch.stack(clips[i:i + batch_windows]).to(model.device)
        feats.append(model.forward_features(x, pooling_method="last").float().cpu())
    return torch.cat(feats).numpy() if feats else np.zeros((0, model.config.hidden_size), np.float32)


@torch.no_grad()
def long_video_features(model, pixel_values: torch.Tensor, window_size: int = 384) -> torch.Tensor:
    """pixel_values [B, total_frames, 3, H, W] -> per-frame pooled features [B, total_frames, D]."""
    B, total = pixel_values.shape[:2]
    nf = model.config.num_frames
    outs = []
    for i in range(0, total, window_size):
        w = pixel_values[:, i:i + window_size]
        pad = (-w.shape[1]) % nf
        if pad:
            w = torch.cat([w, torch.zeros(B, pad, *w.shape[2:], dtype=w.dtype, device=w.device)], dim=1)
        clips = w.reshape(-1, nf, *w.shape[2:])
        pooled = model(clips.to(model.device)).pooler_output            # [B * clips, nf, D]
        outs.append(pooled.reshape(B, -1, pooled.shape[-1]))
    return torch.cat(outs, dim=1)[:, :total]
