"""The multitask wrapper of the pre-training step, as a torch module over the HIP encoder.

Mirror of ``StreamformerForMultiTaskingSigLIP`` (reference ``models/modeling_timesformer_siglip.py:1356-1536``) for the
two task families BASELINE config #3 trains — video-text retrieval (``TimesformerVideoRetrievalHead``, ``:2285-2351``)
and per-frame localization (``TimesformerUniversalLocalizationHead``, ``:2186-2282``):

    model = StreamformerForMultiTaskingSigLIP(config, {"TaskRetrieval": {}, "TaskLocalization": {"label2id": ...}})
    model.prepare_for_multi_tasks(); model.frozen_spatial(); model.cuda().train()
    losses, outputs = model(pixel_values, multi_task_input={"task_name": ..., "task_input": ...})
    losses[task].backward(); optimizer.step()                    # torch.optim over model.parameters()

Encoder forward / backward run in ``libstreamformer_hip.so`` behind one autograd node (``autograd.py``); the loss heads
are the HIP loss kernels (``sf_loss.hip``) behind a second one.  What is NOT here: the SigLIP *text tower* and tokenizer
(``:1365-1373``; hub weights, outside the path) — captions / class prompts enter as feature tensors
(``task_input["text_features"]``, ``set_label_embeddings``) — and the other task heads, which raise
``NotImplementedError`` (out of scope).
"""
from __future__ import annotations

import copy
import math
from typing import Dict, Optional

import torch
from torch import nn

from .configuration import StreamformerConfig
from .heads import LocalizationHead, RetrievalHead
from .modeling import TimesformerMultiTaskingModelSigLIP

RETRIEVAL_TASKS = ("MSRVTT", "WebVid", "TaskRetrieval")                                        # modeling:1401
LOCALIZATION_TASKS = ("THUMOS14Grounding", "ActivityNetGrounding", "FineActionGrounding", "HACSGrounding",
                      "TaskLocalization")                                                       # modeling:1384-1390


class _HeadLossFn(torch.autograd.Function):
    """loss = head.loss(pooler, ...) with the kernel's own gradients (d pooler, d logit_scale, d logit_bias) replayed."""

    @staticmethod
    def forward(ctx, pooler, logit_scale, logit_bias, run):
        loss, gp, gs = run(pooler.detach(), logit_scale.detach(), logit_bias.detach())
        ctx.save_for_backward(gp, gs)
        ctx.shapes = (logit_scale.shape, logit_bias.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        gp, gs = ctx.saved_tensors
        return gp * g, (gs[0] * g).reshape(ctx.shapes[0]), (gs[1] * g).reshape(ctx.shapes[1]), None


class _TaskHead(nn.Module):
    def __init__(self):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.tensor(math.log(10.0)))
        self.logit_bias = nn.Parameter(torch.tensor(-2.0))

    def prepare_multi_task(self, text_encoder=None, text_tokenizer=None, logit_scale=None, logit_bias=None, vision_model=None):
        """modeling:2199-2205 / :2296-2301: every head deep-copies the wrapper's scale / bias pair."""
        if logit_scale is not None:
            self.logit_scale = copy.deepcopy(logit_scale)
        if logit_bias is not None:
            self.logit_bias = copy.deepcopy(logit_bias)


class TimesformerVideoRetrievalHead(_TaskHead):
    """modeling:2285-2351.  ``task_input["text_features"]`` [B, D] stands in for ``encode_captions`` (:2307-2316);
    with torch.distributed initialised, every rank's captions are negatives (distributed SigLipLoss, :239-297)."""

    def __init__(self, config: Optional[StreamformerConfig] = None, gather_negatives: bool = True, process_group=None):
        super().__init__()
        self.config = config
        self.gather_negatives = gather_negatives
        self.group = process_group

    def forward(self, task_head_input, task_specific_input: Optional[dict] = None):
        pooler = task_head_input.pooler_output
        text = task_specific_input["text_features"].to(pooler.device)
        if not self.training:
            img = pooler[:, -1, :]
            return img / img.norm(p=2, dim=-1, keepdim=True), text / text.norm(p=2, dim=-1, keepdim=True)
        rank = 0
        if self.gather_negatives:
            from .parallel import all_gather_rows, world
            rank, ws = world(self.group)
            if ws > 1:
                text = all_gather_rows(text.contiguous(), group=self.group)
            else:
                rank = 0
        text = text.detach()              # frozen text tower (:1374-1375)

        def run(p, ls, lb):
            return RetrievalHead(ls, lb).loss(p, text, rank=rank)
        loss = _HeadLossFn.apply(pooler, self.logit_scale, self.logit_bias, run)
        with torch.no_grad():             # "logits for debugging" (:2346-2350): scale only, as in the reference
            img = pooler[:, -1, :]
            img = img / img.norm(p=2, dim=-1, keepdim=True)
            t = task_specific_input["text_features"].to(pooler.device)
            logits = img @ (t / t.norm(p=2, dim=-1, keepdim=True)).t() * self.logit_scale.exp()
        return loss, logits


class TimesformerUniversalLocalizationHead(_TaskHead):
    """modeling:2186-2282.  ``label2id``: ``{dataset_name: {label: id}}``; the prompt-ensemble class embeddings of
    ``prepare_multi_task`` (:2207-2223, text tower) are supplied through :meth:`set_label_embeddings` ([L, D], unit norm)."""

    def __init__(self, config: Optional[StreamformerConfig] = None, label2id: Optional[dict] = None):
        super().__init__()
        self.config = config
        self.label2id = label2id or {}
        self.dataset_label_embeddings: Dict[str, torch.Tensor] = {}

    def set_label_embeddings(self, dataset_name: str, embeddings: torch.Tensor) -> None:
        self.dataset_label_embeddings[dataset_name] = embeddings.detach()

    def forward(self, task_head_input, task_specific_input: Optional[dict] = None):
        pooler = task_head_input.pooler_output                     # [B, T, D]
        datasets = list(task_specific_input["dataset"])
        labels = task_specific_input["label"].to(pooler.device).long()
        B = pooler.shape[0]
        with torch.no_grad():
            img = pooler / pooler.norm(p=2, dim=-1, keepdim=True)
            all_logits = [img[i] @ self.dataset_label_embeddings[d].to(pooler.device).t() * self.logit_scale.exp() + self.logit_bias
                          for i, d in enumerate(datasets)]
        if not self.training:
            return all_logits
        total = None
        for name in dict.fromkeys(datasets):                        # clips grouped by dataset: one kernel launch per table
            idx = [i for i, d in enumerate(datasets) if d == name]
            emb = self.dataset_label_embeddings[name].to(pooler.device)
            whole = len(idx) == B
            p_g = pooler if whole else pooler[idx]
            lab_g = labels if whole else labels[idx]

            def run(p, ls, lb, emb=emb, lab_g=lab_g):
                return LocalizationHead(emb, ls, lb).loss(p, lab_g)
            part = _HeadLossFn.apply(p_g, self.logit_scale, self.logit_bias, run) * (len(idx) / B)
            total = part if total is None else total + part
        return total, all_logits


class StreamformerForMultiTaskingSigLIP(nn.Module):
    """modeling:1356-1536 without the text tower: ``timesformer`` + per-task heads, one task per call."""

    def __init__(self, config: StreamformerConfig, multi_task_config: Optional[dict] = None, compute_dtype="fp32"):
        super().__init__()
        self.config = config
        self.timesformer = TimesformerMultiTaskingModelSigLIP(config, compute_dtype=compute_dtype)
        self.logit_scale = nn.Parameter(torch.log(torch.tensor(10.0)))
        self.logit_bias = nn.Parameter(torch.tensor(-2.0))
        self.task_heads = nn.ModuleDict()
        self.task_types = list(multi_task_config.keys()) if multi_task_config else []
        for task_type in self.task_types:
            if task_type in LOCALIZATION_TASKS:
                self.task_heads[task_type] = TimesformerUniversalLocalizationHead(config, (multi_task_config[task_type] or {}).get("label2id"))
            elif task_type in RETRIEVAL_TASKS:
                self.task_heads[task_type] = TimesformerVideoRetrievalHead(config)
            else:
                raise NotImplementedError(f"Task type {task_type} not implemented (this build covers the retrieval and "
                                          "localization heads of BASELINE config #3)")
        if config.add_lora_spatial:
            self.add_lora_spatial()
        self.train()                      # an nn.Module is born in train mode; the encoder child alone is born in eval mode

    def frozen_backbone(self):
        for p in self.timesformer.parameters():
            p.requires_grad = False
        print("Backbone frozen")

    def prepare_for_multi_tasks(self):
        for head in self.task_heads.values():
            head.prepare_multi_task(None, None, self.logit_scale, self.logit_bias, self.timesformer)

    def add_lora_spatial(self):
        """modeling:1448-1459: rank-32 factors on every spatial qkv / output.dense; `_add_lora` freezes the base qkv / dense
        (modeling:519-522)."""
        self.timesformer.add_lora_spatial()

    def frozen_spatial(self):
        """modeling:1461-1476: spatial ``attention.qkv`` AND ``output.dense`` (weight + bias) stop training."""
        for i in range(self.config.num_hidden_layers):
            for n in ("attention.attention.qkv", "attention.output.dense"):
                for leaf in ("weight", "bias"):
                    p = self.timesformer._named.get(f"encoder.layer.{i}.{n}.{leaf}")
                    if p is not None:
                        p.requires_grad = False

    def forward(self, pixel_values=None, labels=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                multi_task_input: Optional[dict] = None):
        c = self.config
        if pixel_values.dim() != 5:           # modeling:1497-1503 flattens to clips of config.num_frames; 5-D input keeps its T
            pixel_values = pixel_values.reshape(-1, c.num_frames, 3, c.image_size, c.image_size)
        backbone_outputs = self.timesformer(pixel_values, output_attentions=output_attentions,
                                            output_hidden_states=output_hidden_states, return_dict=True)
        task_name = multi_task_input["task_name"]           # one task at a time (modeling:1514)
        if not self.training:
            return {task_name: self.task_heads[task_name](backbone_outputs, multi_task_input["task_input"])}
        loss, out = self.task_heads[task_name](backbone_outputs, multi_task_input["task_input"])
        return {task_name: loss}, {task_name: out}

    @torch.no_grad()
    def forward_features(self, pixel_values, pooling_method="mean"):
        return self.timesformer.forward_features(pixel_values, pooling_method)
