"""Fixture F9: the reference's TimesformerImageProcessor on seeded uint8 frames (build container only).

    python oracle/make_golden_processor.py      # writes tests/golden/f9_processor.npz

Pins streamformer_amd.processing.TimesformerImageProcessor (and, on the GPU, the uint8 path of the patch
kernel that fuses rescale + normalize) against vqa_enc:1395-1459.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden as G  # noqa: E402
from streamformer_amd.processing import TimesformerImageProcessor  # noqa: E402


def main():
    enc = G.import_vqa_enc()
    rng = np.random.default_rng(9)
    frames = [rng.integers(0, 256, size=(70, 100, 3), dtype=np.uint8) for _ in range(3)]       # HWC, needs a resize
    frames += [rng.integers(0, 256, size=(48, 48, 3), dtype=np.uint8)]                          # already at size
    ref = enc.TimesformerImageProcessor(size=(48, 48))
    want = np.stack(ref.preprocess(frames, return_tensors="np")["pixel_values"]).astype(np.float32)
    mine = TimesformerImageProcessor(size=(48, 48))
    got = mine.preprocess(frames, fused=False)["pixel_values"].numpy()
    d = float(np.abs(got - want).max())
    print("processor max-abs vs reference:", d, want.shape)
    assert d <= 1e-6, d
    u8 = mine.preprocess(frames, fused=True)["pixel_values"]
    assert u8.dtype == torch.uint8 and float(np.abs(mine.normalize(u8).numpy() - want).max()) <= 1e-6
    np.savez_compressed(os.path.join(G.OUT, "f9_processor.npz"), frames_big=np.stack(frames[:3]), frame_small=frames[3],
                        pixel_values=want)
    print("wrote f9_processor.npz", os.path.getsize(os.path.join(G.OUT, "f9_processor.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
