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
* ``python -m streamformer_amd.features --pretrained_model DIR --video_list list.txt --save_path OUT``: the dump
  itself (``extract_oad_feature.py:37-140``): per video resample to 24 fps, short side to 224 (bilinear) + centre crop,
  sliding windows, ``np.save`` of float32 ``[num_windows, D]`` as ``<video stem>.npy``, existing outputs skipped, the
  video list sharded by ``--start_idx / --end_idx`` fractions (``scripts/downstream_extract_oad_feature.sh:29-50``: eight
  processes, one per GPU, no collective) or by RANK / WORLD_SIZE when launched with ``torch.distributed.run``.
  Video DECODING (decord in the reference) is outside the hot path: a "video" here is an ``.npy`` of decoded frames,
  uint8 ``[n, H, W, 3]``.  Frames travel to the GPU as bytes; rescale + normalize run inside the patch kernel.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from typing import List, Optional, Tuple

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


# ---------------------------------------------------------------------------------------------------------------
# the extraction driver
# ---------------------------------------------------------------------------------------------------------------
def resample_indices(num_frames: int, original_fps: float, target_fps: float = 24.0) -> np.ndarray:
    """extract_oad_feature.py:117-120: np.linspace(0, n - 1, int(n / fps * 24)).astype(int)"""
    return np.linspace(0, num_frames - 1, int(num_frames / original_fps * target_fps)).astype(int)


def resize_sizes(H: int, W: int, size: int) -> Tuple[int, int]:
    """(new_h, new_w) of the reference's ``Resize(size)`` on numpy frames: short side to ``size``, long side truncated with
    ``int()`` (functional.py:67-75; 854 x 480 -> 398 x 224, not 399)."""
    if W < H:
        return int(size * H / W), size
    return size, int(size * W / H)


def resize_center_crop(frames_u8: np.ndarray, size: int = 224) -> torch.Tensor:
    """uint8 [n, H, W, 3] -> uint8 [n, 3, size, size]: the reference's ``Resize(224, 'bilinear') + CenterCrop(224)``
    (extract_oad_feature.py:42-46) on numpy frames, i.e. ``cv2.resize(..., INTER_LINEAR)`` — plain bilinear sampling at
    half-pixel centres, NO antialiasing on downscale (PIL's BILINEAR low-pass filters there) — and the crop offset
    ``int(round((dim - size) / 2.0))`` (video_transforms.py:1158-1159).  Computed in float32 and rounded: cv2's 8-bit path uses
    11-bit fixed-point weights, so single pixels can differ by one grey level."""
    n, H, W, _ = frames_u8.shape
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).permute(0, 3, 1, 2)
    if not ((W <= H and W == size) or (H <= W and H == size)):          # functional.py:31-33: already at the minimal size
        nh, nw = resize_sizes(H, W, size)
        out = torch.empty(n, 3, nh, nw, dtype=torch.uint8)
        for i in range(0, n, 64):                                         # chunks: a long video as fp32 is GBs
            y = torch.nn.functional.interpolate(x[i:i + 64].float(), size=(nh, nw), mode="bilinear", align_corners=False, antialias=False)
            out[i:i + 64] = y.round_().clamp_(0, 255).to(torch.uint8)
        x = out
    nh, nw = x.shape[-2:]
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    return x[:, :, top:top + size, left:left + size].contiguous()


def shard_of_list(items: List[str], start_idx: Optional[float], end_idx: Optional[float]) -> List[str]:
    """Fractions of the list (extract_oad_feature.py:66-68) when given, else a balanced contiguous shard for
    RANK of WORLD_SIZE (parallel.shard_range), else everything."""
    if start_idx is not None or end_idx is not None:
        st, ed = int(len(items) * (start_idx or 0.0)), int(len(items) * (1.0 if end_idx is None else end_idx))
        return items[st:ed]
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        from .parallel import shard_range
        lo, hi = shard_range(len(items), rank, world)
        return items[lo:hi]
    return items


def output_path(save_path: str, vid_name: str) -> str:
    return os.path.join(save_path, vid_name.split("/")[-1].split(".")[0] + ".npy")     # extract_oad_feature.py:92


def extract_video(model, frames_u8: np.ndarray, fps: float, window: int = 6, batch_windows: int = 64) -> np.ndarray:
    idx = resample_indices(len(frames_u8), fps)
    clip = resize_center_crop(frames_u8[idx], model.config.image_size)
    return sliding_window_features(model, clip, window=window, batch_windows=batch_windows)


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description="Online-action-detection feature dump on the HIP StreamFormer encoder")
    ap.add_argument("--pretrained_model", required=True, help="checkpoint directory (config.json + weights)")
    ap.add_argument("--ckpt_path", default=None, help="optional training checkpoint ({'model': state_dict}) loaded on top; task_heads dropped")
    ap.add_argument("--enable_lora_spatial", action="store_true")
    ap.add_argument("--video_list", required=True, help="text file: one decoded-frames .npy per line, optionally followed by its fps")
    ap.add_argument("--data_path", default="", help="prefix of the entries of --video_list")
    ap.add_argument("--save_path", required=True)
    ap.add_argument("--fps", type=float, default=24.0, help="fps of entries that do not state their own")
    ap.add_argument("--start_idx", type=float, default=None)
    ap.add_argument("--end_idx", type=float, default=None)
    ap.add_argument("--compute_dtype", default="fp32", choices=["bf16", "fp32"])
    ap.add_argument("--batch_windows", type=int, default=64)
    args = ap.parse_args(argv)
    from .modeling import TimesformerMultiTaskingModelSigLIP
    if not torch.cuda.is_available():
        raise SystemExit("feature extraction runs on an AMD GPU only (no CPU fallback)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.makedirs(args.save_path, exist_ok=True)
    entries: List[Tuple[str, float]] = []
    with open(args.video_list) as f:
        for line in f:
            parts = line.split()
            if parts:
                entries.append((parts[0], float(parts[1]) if len(parts) > 1 else args.fps))
    mine = set(shard_of_list([e[0] for e in entries], args.start_idx, args.end_idx))
    entries = [e for e in entries if e[0] in mine]
    print(f"[{args.start_idx} / {args.end_idx}]: {len(entries)} videos to extract", flush=True)
    model = TimesformerMultiTaskingModelSigLIP.from_pretrained(args.pretrained_model, compute_dtype=args.compute_dtype,
                                                               device=f"cuda:{local}")
    if args.enable_lora_spatial:
        model.add_lora_spatial()
    if args.ckpt_path:
        # the reference's save_model stores its argparse.Namespace under "args" (utils.py:608-636): allow exactly that class
        with torch.serialization.safe_globals([argparse.Namespace]):
            ckpt = torch.load(args.ckpt_path, map_location="cpu", weights_only=True)
        ckpt = ckpt.get("model", ckpt)
        print("Loading checkpoint:", model.load_state_dict({k: v for k, v in ckpt.items() if "task_heads" not in k}, strict=False))
    model.eval()
    for i, (name, fps) in enumerate(entries):
        url = output_path(args.save_path, name)
        if os.path.exists(url):
            continue
        frames = np.load(os.path.join(args.data_path, name))
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[-1] != 3:
            raise SystemExit(f"{name}: expected decoded frames uint8 [n, H, W, 3], got {frames.dtype} {frames.shape}")
        t0 = time.time()
        feats = extract_video(model, frames, fps, batch_windows=args.batch_windows)
        tmp = url + ".tmp.npy"
        np.save(tmp, feats.astype(np.float32))
        os.replace(tmp, url)            # a killed run never leaves a half-written file that a restart would skip
        print(f"[{i} / {len(entries)}]: save feature on {url} with shape:{feats.shape}, used time {time.time() - t0:.2f}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
