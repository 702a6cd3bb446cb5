"""torch.autograd bridge of the drop-in encoder: ``model.train(); out = model(x); loss(out).backward()``.

The reference's multitask wrapper calls the encoder under autograd and steps it with ``torch.optim``
(``models/modeling_timesformer_siglip.py:1486-1523`` forward, ``tools/finetune_tools.py:560-570`` backward + step).
The HIP library has its own staged forward / backward (``sf_trainer_forward`` / ``sf_trainer_backward``, the kernels
``StreamformerTrainer`` drives); this module puts them behind ONE ``torch.autograd.Function`` so that the module's
``nn.Parameter`` tensors receive ``.grad`` like any other torch module:

* forward: the module's parameters are copied into the library's flat fp32 buffer when their version counters moved
  (one multi-tensor copy), the bf16 operand copies are refreshed, ``sf_trainer_forward`` keeps the activations in a
  workspace;
* backward: ``sf_trainer_backward`` on ``d pooler_output`` (and ``d last_hidden_state`` when it is used) into a zeroed
  flat gradient buffer; each parameter that requires grad gets a copy of its slice (autograd accumulates it into
  ``.grad``, so ``update_freq`` > 1 and several backward calls per step behave as in torch).

The training arithmetic is the bf16 mode (bf16 MFMA operands, fp32 accumulation and residual stream), whatever
``compute_dtype`` the module uses for inference.  No gradient is produced for ``pixel_values``.  There is no torch
fallback: without the library / a GPU this raises.
"""
from __future__ import annotations

from typing import List, Optional

import torch


def _is_spatial_base(name: str) -> bool:
    return (".attention.attention.qkv." in name or ".attention.output.dense." in name) and "temporal_attention" not in name


class TrainEngine:
    """Flat-buffer training state of one module on one device (a ``StreamformerTrainer`` without heads / optimizer)."""

    def __init__(self, module):
        from .training import StreamformerTrainer
        self.module_token = None
        named = module._named
        lora = module._lora
        # frozen spatial base weights (frozen_spatial(), modeling:1471-1484, with or without LoRA): the library then skips those
        # weight-gradient GEMMs instead of computing gradients that backward() would throw away
        freeze = self._frozen_spatial(module)
        sd = {k: p.detach() for k, p in named.items()}
        # stochastic depth / dropout of the training forwards draw from a generator seeded by `module.stochastic_seed` (default 0): set it
        # (and the engine's `tr._dp_gen` state on resume) to continue a sequence of masks instead of replaying it
        self.tr = StreamformerTrainer(module.config, sd, [], freeze_spatial=freeze, device=module.device, with_optimizer=False,
                                      drop_path_seed=int(getattr(module, "stochastic_seed", 0)), task_sync_check="never")
        self.freeze = freeze
        self.names: List[str] = list(named.keys())
        self._views = [self.tr._view(k) for k in self.names]
        self.forward_id = 0
        self.signature = self._signature(module)

    @staticmethod
    def _frozen_spatial(module) -> bool:
        base = [p for k, p in module._named.items() if _is_spatial_base(k)]
        return bool(base) and not any(p.requires_grad for p in base)

    @staticmethod
    def _signature(module):
        return (module.device, module._lora, TrainEngine._frozen_spatial(module), len(module._named))

    def upload(self, module) -> None:
        """Module parameters -> flat buffer + bf16 operand refresh, only when something changed."""
        tok = (sum(p._version for p in module._plist), tuple(p.data_ptr() for p in module._plist[:4]))
        if tok == self.module_token:
            return
        src = [p.detach() for p in module._plist]
        if all(s.dtype == torch.float32 for s in src):
            torch._foreach_copy_(self._views, src)
        else:
            for v, s_ in zip(self._views, src):
                v.copy_(s_)
        self.tr.sync_weights()
        self.module_token = tok


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, pixel_values, *params):
        eng: TrainEngine = module._train_engine()
        eng.upload(module)
        if pixel_values.dtype == torch.uint8:
            # the training forward normalises raw uint8 frames with the default mean = std = 0.5, rescale 1 / 255; a module whose
            # image processor differs would see different inputs in train() and eval(): apply the module's own
            # normalisation on the way in instead
            ip = getattr(module, "image_processor", None)
            mean = tuple(float(m) for m in getattr(ip, "image_mean", (0.5, 0.5, 0.5))) if ip is not None else (0.5, 0.5, 0.5)
            std = tuple(float(m) for m in getattr(ip, "image_std", (0.5, 0.5, 0.5))) if ip is not None else (0.5, 0.5, 0.5)
            rescale = float(getattr(ip, "rescale_factor", 1.0 / 255.0)) if ip is not None else 1.0 / 255.0
            if any(abs(m - 0.5) > 1e-12 for m in mean) or any(abs(m - 0.5) > 1e-12 for m in std) or abs(rescale - 1.0 / 255.0) > 1e-12:
                c = pixel_values.shape[2]
                mt = torch.tensor(mean[:c], dtype=torch.float32, device=pixel_values.device).view(1, 1, c, 1, 1)
                st = torch.tensor(std[:c], dtype=torch.float32, device=pixel_values.device).view(1, 1, c, 1, 1)
                pixel_values = (pixel_values.to(torch.float32) * rescale - mt) / st
        lhs, pool = eng.tr.forward(pixel_values)
        eng.forward_id += 1
        ctx.eng, ctx.fid = eng, eng.forward_id
        ctx.names = [k for k, p in module._named.items() if p.requires_grad]
        ctx.set_materialize_grads(False)
        return lhs, pool

    @staticmethod
    def backward(ctx, d_lhs: Optional[torch.Tensor], d_pool: Optional[torch.Tensor]):
        eng: TrainEngine = ctx.eng
        if ctx.fid != eng.forward_id:
            raise RuntimeError("backward through a StreamFormer forward whose saved activations were overwritten by a later "
                               "training forward of the same module (one workspace per module: call backward before the next forward)")
        tr = eng.tr
        if d_pool is None:
            d_pool = torch.zeros_like(tr._pooler)
        tr.grads.zero_()
        tr.backward(d_pool, d_lhs)
        out = []
        for k in ctx.names:
            e = tr._entry(k)
            out.append(tr.grad(k).clone() if e["trainable"] else None)
        return (None, None) + tuple(out)


def encoder_forward_with_grad(module, pixel_values: torch.Tensor):
    """(last_hidden_state, pooler_output) connected to the module's parameters that require grad."""
    params = [p for p in module._plist if p.requires_grad]
    return _EncoderFn.apply(module, pixel_values, *params)
