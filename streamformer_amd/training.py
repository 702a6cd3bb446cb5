"""Training step of the multitask pre-training (BASELINE configs #3 / #4) on the HIP library.

Host-side mirror of the reference's step:

* ``train_one_epoch_multi_task`` (``tools/finetune_tools.py:395-573``): one task per micro-batch,
  ``loss /= update_freq``, backward every micro-step, optimizer step + zero_grad every
  ``update_freq``-th micro-step, lr / weight-decay values written per step from the cosine tables
  (``utils.py:574-605``, :func:`cosine_scheduler` here);
* what trains: everything except the spatial ``attention.attention.qkv`` / ``attention.output.dense``
  base weights when ``freeze_spatial`` (``modeling:1471-1484``); LoRA factors, the temporal branch,
  the MLPs, embeddings, the pooling head and each task head's ``logit_scale`` / ``logit_bias`` train;
* AdamW with the ``optim_factory.py:59-104`` grouping (no decay for 1-D and ``*.bias``);
* data parallel: one process per GPU, gradients summed by RCCL all-reduce in ``L + 2`` slices that are
  issued as the staged backward finishes them, averaged inside the optimizer kernel
  (``run_finetuning_multi_task.py:421`` wraps the model in DDP; same mathematics, fewer, larger
  collectives over xGMI).

All state lives in four flat fp32 device tensors (params / grads / exp_avg / exp_avg_sq) whose layout
the library defines (``sf_trainer_param_info``); the compute path is the C ABI — there is no torch
autograd and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from . import _native as nat
from .configuration import StreamformerConfig

_ACT = {"gelu": 0}


def cosine_scheduler(base_value: float, final_value: float, epochs: int, niter_per_ep: int, warmup_epochs: int = 0,
                     start_warmup_value: float = 0.0, warmup_steps: int = -1) -> List[float]:
    """Per-iteration value table of the reference (``utils.py:574-605``): linear warm-up then half-cosine."""
    warmup_iters = int(warmup_epochs * niter_per_ep)
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    sched: List[float] = []
    if warmup_epochs > 0:            # np.linspace(start, base, warmup_iters): both endpoints included
        sched = [start_warmup_value if warmup_iters == 1 else
                 start_warmup_value + (base_value - start_warmup_value) * i / (warmup_iters - 1) for i in range(warmup_iters)]
    n = epochs * niter_per_ep - warmup_iters
    sched += [final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / n)) for i in range(n)]
    if len(sched) != epochs * niter_per_ep:
        raise ValueError("warmup_steps without warmup_epochs leaves the table short (the reference asserts here too)")
    return sched


def scaled_lr(lr: float, batch_size: int, update_freq: int, world_size: int) -> float:
    """``args.lr * total_batch_size / 256`` (``run_finetuning_multi_task.py:386-388``)."""
    return lr * batch_size * update_freq * world_size / 256.0


def drop_path_factors(rate: float, num_layers: int, B: int, T: int, N: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Keep / drop factors of one training forward, [L, B*N + B*T + B] fp32 on the CPU.

    The reference (modeling:846-856, 460-486) gives layer i the rate ``linspace(0, drop_path_rate, L)[i]`` and draws, per
    residual branch, one Bernoulli(keep) per dim-0 entry of the tensor the branch returns — (B*N, T, D) temporal,
    (B*T, N, D) spatial, (B, N*T, D) MLP — as ``floor(keep + rand)``, dividing the kept entries by ``keep``."""
    per = B * N + B * T + B
    out = torch.ones(num_layers, per, dtype=torch.float32)
    rates = torch.linspace(0, rate, num_layers).tolist()
    for i, r in enumerate(rates):
        if r <= 0.0:
            continue
        keep = 1.0 - r
        out[i] = torch.floor(keep + torch.rand(per, generator=generator)) / keep
    return out


def bucket_ranges(stage_ranges: Sequence[Tuple[int, int]], min_floats: int) -> List[Tuple[int, int, int]]:
    """Group consecutive backward stages into all-reduce buckets of at least ``min_floats`` floats.

    Stages finish in order 0, 1, ...; their slices are adjacent going DOWN the buffer (head first, then
    layers L-1..0, then embeddings), so a bucket is one contiguous slice.  Returns
    ``(last_stage, offset, numel)`` per bucket: issue it once ``last_stage`` has been enqueued.
    xGMI rings are per-link bound (7 links x ~153 GB/s): few large slices beat many small ones.
    """
    out, lo, hi, count = [], None, None, 0
    for st, (off, n) in enumerate(stage_ranges):
        lo = off if lo is None else min(lo, off)
        hi = off + n if hi is None else max(hi, off + n)
        count += n
        if count >= min_floats or st == len(stage_ranges) - 1:
            out.append((st, lo, hi - lo))
            lo = hi = None
            count = 0
    return out


class StreamformerTrainer:
    """Owns the flat training state of one rank and runs micro-steps through ``libstreamformer_hip``."""

    def __init__(self, config: StreamformerConfig, state_dict: Dict[str, torch.Tensor], task_heads: Sequence[str],
                 freeze_spatial: bool = True, device="cuda", lr: float = 1e-3, weight_decay: float = 0.05,
                 betas=(0.9, 0.999), eps: float = 1e-8, process_group=None, bucket_mb: float = 64.0,
                 grad_reduce_dtype: str = "fp32", collectives_at_world_1: bool = False, with_optimizer: bool = True,
                 drop_path_seed: int = 0, nonfinite_guard: bool = True, task_sync_check: str = "first"):
        if not torch.cuda.is_available():
            raise RuntimeError("StreamformerTrainer needs an AMD GPU: the training step runs only on the HIP library")
        if config.hidden_act not in _ACT:
            raise NotImplementedError(f"training supports hidden_act in {sorted(_ACT)}")
        if config.attention_type != "divided_space_time":
            raise NotImplementedError("only divided_space_time is implemented")
        # dropout (modeling:374, 378, 752, 761, 822, 835 hidden; 556, 603, 669, 705 attention probabilities): counter-based masks, a
        # fresh seed per forward from this rank's generator; the shipped recipes leave both probabilities at 0
        self.hidden_dropout = float(getattr(config, "hidden_dropout_prob", 0.0) or 0.0)
        self.attention_dropout = float(getattr(config, "attention_probs_dropout_prob", 0.0) or 0.0)
        if not (0.0 <= self.hidden_dropout < 1.0 and 0.0 <= self.attention_dropout < 1.0):
            raise ValueError("dropout probabilities must be in [0, 1)")
        if self.attention_dropout > 0.0:
            # the masked variants exist for the DMA-staged attention kernels only (sf_train.hip: T <= 16, N <= 224 patches per frame);
            # say so here, not as SF_ERR_INVALID at the first forward
            n_patches = (config.image_size // config.patch_size) ** 2
            if n_patches > 224 or config.num_frames > 16:
                raise NotImplementedError(f"attention_probs_dropout_prob > 0 needs <= 224 patches per frame and <= 16 frames "
                                          f"(this config: {n_patches} patches, {config.num_frames} frames)")
        self.dropout = True                         # False: forwards without dropout (evaluation through the trainer)
        self.last_dropout: Optional[Tuple[int, float, float]] = None    # (seed, hidden_p, attention_p) of the last forward, for replay in tests
        self.drop_path_rate = float(getattr(config, "drop_path_rate", 0.0) or 0.0)
        self.drop_path = True                       # False: forwards without stochastic depth (evaluation through the trainer)
        self.last_drop_path: Optional[torch.Tensor] = None     # the factors of the last forward (CPU), for replay in tests
        self._dp_dev = None
        if grad_reduce_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_reduce_dtype must be 'fp32' or 'bf16'")
        self.grad_reduce_dtype = grad_reduce_dtype
        self.config = config
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.task_heads = list(task_heads)
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.group = process_group
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world = torch.distributed.get_world_size(process_group) if dist_on else 1
        self.rank = torch.distributed.get_rank(process_group) if dist_on else 0
        self._dp_gen = torch.Generator().manual_seed(int(drop_path_seed) * 1000003 + self.rank)      # ranks draw different masks
        # world_size 1 normally skips every collective; with this flag they are issued anyway (a 1-rank all-reduce /
        # all-gather is the identity), which is how the RCCL branches get executed and tested on a one-GPU box
        self._collectives = dist_on and (self.world > 1 or bool(collectives_at_world_1))
        self.comm_enabled = True            # bench.py switches the gradient all-reduce off to measure its exposed time
        # Device sharing (profiles/r06_device_sharing.txt): workgroups of the temporal attention backward disturb ANY wave of
        # another kernel that shares their CU and runs an LDS-fed packed-FMA accumulate (synthetic victim, no library code; one process
        # on two streams is enough).  With world > 1 the collective's kernels run beside the backward, so that kernel then takes the
        # workgroups that own their CU (bit-identical, +0.17 ms per step); SF_TBWD_OWN_CU=0/1 in the environment overrides.
        if self.world > 1 and "SF_TBWD_OWN_CU" not in os.environ:
            os.environ["SF_TBWD_OWN_CU"] = "1"
            nat.lib.sf_reload_switches()
        # every rank must run the SAME task in a micro-step (the reference's sampler guarantees it, sampler.py:218-337): a
        # retrieval rank issues the caption all-gather, a localization rank does not, and mismatched collectives hang.
        # "first": verify the first micro-step of this trainer (one tiny all-gather + host read), "always": every micro-step
        # (debug), "never": trust the caller.
        if task_sync_check not in ("first", "always", "never"):
            raise ValueError("task_sync_check must be 'first', 'always' or 'never'")
        self.task_sync_check = task_sync_check
        self._task_checked = False
        c = config
        sc = nat.SfConfig(c.image_size, c.patch_size, c.num_channels, c.num_frames, c.hidden_size, c.num_hidden_layers,
                          c.num_attention_heads, c.intermediate_size, _ACT[c.hidden_act], int(c.qkv_bias),
                          int(c.enable_causal_temporal), int(c.add_lora_spatial), float(c.layer_norm_eps))
        h = C.c_void_p()
        nat.check(nat.lib.sf_trainer_create(C.byref(sc), self.device.index or 0, int(freeze_spatial), 2 * len(self.task_heads),
                                            C.byref(h)))
        self._h = h
        # ---- layout -------------------------------------------------------------------------------------
        self.layout: Dict[str, dict] = {}
        name = C.create_string_buffer(256)
        off, numel, nd, tr, dec = C.c_int64(), C.c_int64(), C.c_int(), C.c_int(), C.c_int()
        shape = (C.c_int64 * 4)()
        for i in range(nat.lib.sf_trainer_num_params(h)):
            nat.check(nat.lib.sf_trainer_param_info(h, i, name, 256, C.byref(off), C.byref(numel), shape, C.byref(nd),
                                                    C.byref(tr), C.byref(dec)))
            self.layout[name.value.decode()] = dict(offset=off.value, numel=numel.value, shape=tuple(shape[:nd.value]),
                                                    trainable=bool(tr.value), decay=bool(dec.value))
        total, ntrain = C.c_int64(), C.c_int64()
        nat.check(nat.lib.sf_trainer_total_floats(h, C.byref(total), C.byref(ntrain)))
        self.total, self.n_train = total.value, ntrain.value
        self.extra_slot = {}
        for i, t in enumerate(self.task_heads):
            self.extra_slot[f"task_heads.{t}.logit_scale"] = f"extra.{2 * i}"
            self.extra_slot[f"task_heads.{t}.logit_bias"] = f"extra.{2 * i + 1}"
        self.stage_ranges = []
        for st in range(nat.lib.sf_trainer_num_stages(h)):
            nat.check(nat.lib.sf_trainer_stage_range(h, st, C.byref(off), C.byref(numel)))
            self.stage_ranges.append((off.value, numel.value))
        self.buckets = bucket_ranges(self.stage_ranges, int(bucket_mb * (1 << 20) / 4))
        # ---- state ----------------------------------------------------------------------------------------
        dev = self.device
        self.params = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        # the autograd bridge (autograd.py) leaves the update to torch.optim: no moment buffers then
        self.exp_avg = torch.zeros(self.n_train, dtype=torch.float32, device=dev) if with_optimizer else None
        self.exp_avg_sq = torch.zeros(self.n_train, dtype=torch.float32, device=dev) if with_optimizer else None
        self._scratch = torch.zeros(4, dtype=torch.float32, device=dev)
        # non-finite guard (tools/finetune_tools.py:533-541, utils.py:515-551): checked ON THE DEVICE inside the optimizer
        # kernel — {sticky flag, skipped steps}; the host looks at it only in check_finite() / at checkpoints
        # non-finite guard: one {sticky flag, skipped steps} pair PER SET OF HEADS that stepped together (row 0 = "every head" / no heads),
        # so that reset_nonfinite() can take a skipped step back out of exactly the heads it involved; 32 rows, views of one tensor
        self._guard = torch.zeros(32, 2, dtype=torch.int32, device=dev) if nonfinite_guard else None
        self._guard_rows = {}               # frozenset(active heads) -> row of self._guard
        self._last_loss: Optional[torch.Tensor] = None
        # sum of the losses of the current accumulation window (update_freq > 1) — a non-finite micro-step loss stays visible to the
        # guard of the optimizer step that closes the window; with world > 1 it is all-reduced next to the gradients so that every
        # rank takes the same skip / step decision
        self._loss_acc = torch.zeros(1, dtype=torch.float32, device=dev)
        self._loss_acc_used = False
        self._wire: Optional[torch.Tensor] = None       # bf16 wire format of the gradient all-reduce (grad_reduce_dtype="bf16")
        self.step_count = 0
        self.micro = 0
        # torch.optim.AdamW keeps `step` per parameter and skips parameters whose grad is None: a head whose task was not
        # scheduled in an accumulation window is left alone (no decay, no moment decay) and keeps its own step count
        self.head_steps = {t: 0 for t in self.task_heads}
        self._touched = set()
        self._ws = None
        self._ws_key = None
        self._pooler = self._lhs = None
        for t in self.task_heads:           # modeling:1363-1364: each head deep-copies log(10) / -2 ...
            self._view(f"task_heads.{t}.logit_scale").fill_(math.log(10.0))
            self._view(f"task_heads.{t}.logit_bias").fill_(-2.0)
        self.load_state_dict(state_dict)    # ... unless the checkpoint carries trained values for this trainer's heads
        self.sync_weights()

    # ---- parameter access --------------------------------------------------------------------------------
    def _entry(self, key: str) -> dict:
        return self.layout[self.extra_slot.get(key, key)]

    def _view(self, key: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        e = self._entry(key)
        buf = self.params if buf is None else buf
        return buf[e["offset"]: e["offset"] + e["numel"]].view(e["shape"])

    def parameter_names(self, trainable_only: bool = False) -> List[str]:
        inv = {v: k for k, v in self.extra_slot.items()}
        return [inv.get(k, k) for k, e in self.layout.items() if e["trainable"] or not trainable_only]

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Accepts encoder keys as well as wrapper / DDP checkpoints (``timesformer.`` / ``module.`` prefixes,
        ``task_heads.<task>.logit_scale|logit_bias`` of this trainer's heads; other head entries are ignored)."""
        from .modeling import normalize_checkpoint_keys
        heads = {}
        for k, v in sd.items():
            kk = k[len("module."):] if k.startswith("module.") else k
            if kk in self.extra_slot:
                heads[kk] = v
        sd = normalize_checkpoint_keys(sd)
        for k, v in heads.items():
            self._view(k).copy_(v.to(torch.float32).reshape(self._entry(k)["shape"]))
        missing = []
        for k, e in self.layout.items():
            if k.startswith("extra."):
                continue
            if k not in sd:
                missing.append(k)
                continue
            self._view(k).copy_(sd[k].to(torch.float32).reshape(e["shape"]))
        if missing:
            raise KeyError(f"state_dict lacks {len(missing)} tensors, e.g. {missing[:3]}")

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {k: self._view(k).detach().clone() for k in self.layout if not k.startswith("extra.")}
        for k in self.extra_slot:
            out[k] = self._view(k).detach().clone()
        return out

    # ---- checkpoint / resume (reference layout: utils.py:608-636) ---------------------------------------
    def _optimizer_order(self) -> List[str]:
        """Names in the order the reference WRAPPER yields trainable parameters from ``named_parameters()``: its own
        ``logit_scale`` / ``logit_bias`` first (modeling:1363-1364 — trainable, never used by a head, so they never get a
        gradient or optimizer state), then the encoder tree under ``timesformer.``, then the task heads' deep copies.
        torch.optim numbers its state in that order, group by group (optim_factory.py:59-104).  Pinned by
        tests/golden/f12_param_order.json (``wrapper_lora``: generated from the reference's wrapper class)."""
        from .modeling import expected_keys
        enc = ["timesformer." + k for k in expected_keys(self.config) if self.layout[k]["trainable"]]
        heads = [k for t in self.task_heads for k in (f"task_heads.{t}.logit_scale", f"task_heads.{t}.logit_bias")]
        return ["logit_scale", "logit_bias"] + enc + heads

    _WRAPPER_SCALARS = ("logit_scale", "logit_bias")

    def _opt_entry(self, name: str) -> Optional[dict]:
        if name in self._WRAPPER_SCALARS:
            return None
        return self._entry(name[len("timesformer."):] if name.startswith("timesformer.") else name)

    def _opt_step_of(self, name: str) -> int:
        if name.startswith("task_heads."):
            return self.head_steps[name.split(".")[1]]
        return self.step_count

    @staticmethod
    def _no_decay(name: str, shape) -> bool:
        return len(shape) == 1 or name.endswith(".bias")           # optim_factory.py:72-77, skip_list = ()

    def _optimizer_groups(self) -> Dict[str, List[str]]:
        """{"decay": [...], "no_decay": [...]} in order of first appearance, as get_parameter_groups fills them."""
        groups: Dict[str, List[str]] = {}
        for n in self._optimizer_order():
            e = self._opt_entry(n)
            shape = () if e is None else e["shape"]
            groups.setdefault("no_decay" if self._no_decay(n, shape) else "decay", []).append(n)
        return groups

    def optimizer_state_dict(self) -> dict:
        """``torch.optim.AdamW.state_dict()`` of the reference's optimizer over the same parameters: two groups in order
        of first appearance ("decay" / "no_decay"), ids running through the groups, per-id ``step / exp_avg / exp_avg_sq``.
        Parameters that never received a gradient have no state entry, exactly as torch leaves them: the wrapper's
        own scalars always, a head whose task has not run yet."""
        groups = self._optimizer_groups()
        state, param_groups, pid = {}, [], 0
        for gname, members in groups.items():
            ids = []
            for n in members:
                e = self._opt_entry(n)
                step = 0 if e is None else self._opt_step_of(n)
                if step > 0:
                    sl = slice(e["offset"], e["offset"] + e["numel"])
                    state[pid] = {"step": torch.tensor(float(step)),
                                  "exp_avg": self.exp_avg[sl].view(e["shape"]).detach().cpu().clone(),
                                  "exp_avg_sq": self.exp_avg_sq[sl].view(e["shape"]).detach().cpu().clone()}
                ids.append(pid)
                pid += 1
            param_groups.append({"weight_decay": 0.0 if gname == "no_decay" else self.weight_decay, "lr_scale": 1.0, "lr": self.lr,
                                 "betas": tuple(self.betas), "eps": self.eps, "amsgrad": False, "maximize": False, "foreach": None,
                                 "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True,
                                 "params": ids})
        return {"state": state, "param_groups": param_groups, "param_names": [n for g in groups.values() for n in g]}

    def load_optimizer_state_dict(self, osd: dict) -> None:
        """Inverse of :meth:`optimizer_state_dict`; also accepts the reference's own ``optimizer.state_dict()``: ids are
        mapped by position in the reference's enumeration (the head names of a reference file — ``TaskRetrieval`` … — need
        not match this trainer's), ids without state are skipped, the encoder's common step count becomes ``step_count``
        and every head keeps its own.  Files written before round 3 (no ids for the wrapper's scalars) still load."""
        names = [n for g in self._optimizer_groups().values() for n in g]
        n_ids = sum(len(g["params"]) for g in osd["param_groups"])
        # Which enumeration the file uses is decided by what the file SAYS, not by a count that a reference checkpoint with one
        # task head fewer would also match: files of this trainer carry "param_names"; a file without them is a
        # reference optimizer.state_dict() (wrapper scalars included) unless it is exactly the round-2 layout of this trainer,
        # which had no ids for the wrapper's scalars AND no "param_names".
        stored = osd.get("param_names")
        if stored is not None:
            if len(stored) != n_ids:
                raise ValueError(f"optimizer state lists {len(stored)} parameter names for {n_ids} ids")
            legacy = not any(n in self._WRAPPER_SCALARS for n in stored)
            want = [n for n in names if n not in self._WRAPPER_SCALARS] if legacy else names
            if len(stored) != len(want):
                raise ValueError(f"optimizer state covers {len(stored)} parameters, this trainer enumerates {len(want)}")
            # head names of a reference-side file need not match this trainer's: compare everything but the task-head entries
            diff = [(a, b) for a, b in zip(stored, want) if a != b and not (a.startswith("task_heads.") and b.startswith("task_heads."))]
            if diff:
                raise ValueError(f"optimizer state was written for a different parameter enumeration, e.g. {diff[0][0]!r} where this trainer has {diff[0][1]!r}")
            names = want
        elif n_ids == len(names) - 2 and self._round2_layout_matches(osd, [n for n in names if n not in self._WRAPPER_SCALARS]):
            names = [n for n in names if n not in self._WRAPPER_SCALARS]      # round-2 file of this trainer
        elif n_ids != len(names):
            raise ValueError(f"optimizer state covers {n_ids} parameters, this trainer's reference-side enumeration has {len(names)} "
                             f"(2 wrapper scalars + {len(names) - 2} trainable)")
        ids = [i for g in osd["param_groups"] for i in g["params"]]
        by_id = dict(zip(ids, names))
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        enc_steps = set()
        head_steps = {t: set() for t in self.task_heads}
        for pid, st in osd.get("state", {}).items():
            n = by_id[int(pid)]
            e = self._opt_entry(n)
            if e is None:
                continue                                            # state on the wrapper's unused scalars: nothing to restore
            sl = slice(e["offset"], e["offset"] + e["numel"])
            for key in ("exp_avg", "exp_avg_sq"):
                if tuple(st[key].shape) != tuple(e["shape"]):
                    raise ValueError(f"optimizer state {pid}: {key} has shape {tuple(st[key].shape)}, {n} is {tuple(e['shape'])}")
                if not st[key].dtype.is_floating_point:
                    raise ValueError(f"optimizer state {pid}: {key} has dtype {st[key].dtype}")
            self.exp_avg[sl].copy_(st["exp_avg"].to(torch.float32).reshape(-1))
            self.exp_avg_sq[sl].copy_(st["exp_avg_sq"].to(torch.float32).reshape(-1))
            step = int(float(st["step"]))
            if n.startswith("task_heads."):
                head_steps[n.split(".")[1]].add(step)
            else:
                enc_steps.add(step)
        if len(enc_steps) > 1:
            raise ValueError(f"encoder parameters carry different step counts ({sorted(enc_steps)}): the fused AdamW keeps one")
        self.step_count = enc_steps.pop() if enc_steps else 0
        for t, ss in head_steps.items():
            if len(ss) > 1:
                raise ValueError(f"logit_scale / logit_bias of head {t!r} carry different step counts {sorted(ss)}")
            self.head_steps[t] = ss.pop() if ss else 0
        g0 = osd["param_groups"][0]
        self.lr = float(g0.get("lr", self.lr))
        decays = [float(g["weight_decay"]) for g in osd["param_groups"] if float(g.get("weight_decay", 0.0)) > 0]
        if decays:
            self.weight_decay = decays[0]

    def _round2_layout_matches(self, osd: dict, names: List[str]) -> bool:
        """True when every state entry's shape fits the parameter the round-2 enumeration (no wrapper scalars) assigns to its id."""
        ids = [i for g in osd["param_groups"] for i in g["params"]]
        by_id = dict(zip(ids, names))
        for pid, st in osd.get("state", {}).items():
            n = by_id.get(int(pid))
            e = None if n is None else self._opt_entry(n)
            if e is None or tuple(st["exp_avg"].shape) != tuple(e["shape"]):
                return False
        return True

    def _gather_stochastic_states(self) -> list:
        """Every rank's generator state, in rank order (a collective when world > 1: call it on ALL ranks)."""
        mine = self._dp_gen.get_state()
        if self.world > 1 and torch.distributed.is_initialized():
            states = [None] * self.world
            torch.distributed.all_gather_object(states, mine, group=self.group)
            return states
        return [mine]

    def checkpoint(self, epoch: int = 0, args=None, stochastic_states: Optional[list] = None) -> dict:
        """The dict the reference's ``save_model`` writes on rank 0: wrapper-keyed weights (``timesformer.*``,
        ``task_heads.*``), optimizer state, epoch.  ``scaler`` is empty: bf16 operands need no loss scaling.
        With world > 1 call :meth:`save_checkpoint` (every rank) rather than this method on rank 0 alone: the finite check and the
        per-rank generator states are collectives."""
        self.check_finite(collective=False)
        if self.world > 1 and stochastic_states is None:
            import warnings
            warnings.warn("StreamformerTrainer.checkpoint() on one rank of a world > 1 job stores no per-rank generator states and runs a "
                          "non-collective finite check: a resumed run will not continue the drop_path / dropout draws. Call "
                          "save_checkpoint() on every rank instead.", RuntimeWarning, stacklevel=2)
        model = OrderedDict()
        model["logit_scale"] = torch.tensor(math.log(10.0))        # the wrapper's own pair (modeling:1363-1364): never trained,
        model["logit_bias"] = torch.tensor(-2.0)                   # kept so that the reference's load_state_dict finds its keys
        for k, v in self.state_dict().items():
            model[k if k.startswith("task_heads.") else "timesformer." + k] = v.detach().cpu()
        # "stochastic_state": this rank's generator of drop_path factors and dropout seeds, so that a resumed run continues the
        # sequence of masks instead of replaying it from the start; an extra key the reference's loader ignores
        # per-rank states: the ranks draw from generators seeded seed * 1000003 + rank; a resume must hand every rank
        # ITS state back, not rank 0's
        states = stochastic_states if stochastic_states is not None else ([self._dp_gen.get_state()] if self.world == 1 else None)
        return {"model": model, "optimizer": self.optimizer_state_dict(), "epoch": int(epoch), "scaler": {}, "args": args,
                "stochastic_state": states}

    def save_checkpoint(self, path: str, epoch: int = 0, args=None) -> None:
        """Call on EVERY rank: the non-finite check raises on all of them together (a rank-0-only raise would leave the others
        hanging in their next collective), the generator states are gathered, rank 0 writes."""
        self.check_finite()
        states = self._gather_stochastic_states()
        if self.rank == 0:
            tmp = path + ".tmp"
            torch.save(self.checkpoint(epoch, args, stochastic_states=states), tmp)
            os.replace(tmp, path)

    def load_checkpoint(self, path_or_dict) -> int:
        """Resume: weights, Adam moments, step count.  Returns the stored epoch.  (The reference's pre-training driver only
        reloads weights, run_finetuning_multi_task.py:353-357; the optimizer entry is what its downstream ``auto_load_model``
        restores, utils.py:670-877.)"""
        ck = torch.load(path_or_dict, map_location="cpu", weights_only=False) if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
        self.load_state_dict(ck["model"])
        if ck.get("optimizer"):
            self.load_optimizer_state_dict(ck["optimizer"])
        st = ck.get("stochastic_state")
        if isinstance(st, (list, tuple)):
            if len(st) == self.world:
                self._dp_gen.set_state(st[self.rank])         # this rank's own sequence continues
            # a file written at another world size: keep the generator this trainer was constructed with (seed * 1000003 + rank)
        elif st is not None and self.world == 1:
            self._dp_gen.set_state(st)                         # round-4 file: one state, meaningful for a one-rank run only
        self.micro = 0
        self.grads.zero_()
        self.sync_weights()
        return int(ck.get("epoch", 0))

    def grad(self, key: str) -> torch.Tensor:
        e = self._entry(key)
        if not e["trainable"]:
            raise KeyError(f"{key} is frozen")
        return self.grads[e["offset"]: e["offset"] + e["numel"]].view(e["shape"])

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and nat is not None and getattr(nat, "lib", None) is not None:    # interpreter shutdown: globals may be gone
            nat.lib.sf_trainer_destroy(h)
            self._h = None

    # ---- guards ------------------------------------------------------------------------------------------
    def nonfinite_steps(self) -> int:
        """Optimizer steps the device-side guard has skipped so far (reads the flag: synchronises with the GPU)."""
        return 0 if self._guard is None else int(self._guard[:, 1].sum().item())

    def _guard_row(self, active) -> torch.Tensor:
        key = frozenset(active)
        if key not in self._guard_rows:
            if len(self._guard_rows) >= self._guard.shape[0] - 1:          # more distinct head sets than rows: share the last row
                return self._guard[-1]
            self._guard_rows[key] = len(self._guard_rows)
        return self._guard[self._guard_rows[key]]

    def reset_nonfinite(self) -> int:
        """Acknowledge skipped steps and continue: clears the sticky flag and takes the skipped steps back out of the host-side step
        counts (the device skipped the update, the host had already counted it).  Returns the number of steps that had been skipped."""
        n = self.nonfinite_steps()
        if n:
            counts = self._guard[:, 1].cpu().tolist()
            self.step_count = max(0, self.step_count - n)
            for key, row in self._guard_rows.items():        # a head's own step count rewinds only by the skipped steps it took part in
                for t in key:
                    self.head_steps[t] = max(0, self.head_steps[t] - counts[row])
            if counts[-1] and len(self._guard_rows) >= self._guard.shape[0] - 1:      # shared overflow row: heads unknown, rewind all (as before)
                for t in self.head_steps:
                    self.head_steps[t] = max(0, self.head_steps[t] - counts[-1])
            self._guard.zero_()
        return n

    def check_finite(self, collective: bool = True) -> None:
        """With world > 1 every rank must call this at the same point (the flag is all-reduced first, so that all ranks raise together).
        The reference stops the run when a loss is not finite (``tools/finetune_tools.py:533-541``: ``sys.exit(1)``) and its
        GradScaler skips a step with inf gradients (``utils.py:515-551``).  Here the optimizer kernel makes both checks on the
        device (the step is sync-free) and skips the update; this host-side read — call it wherever the loop already touches
        the host (logging, checkpoints; ``checkpoint()`` does) — raises once that has happened.  Weights and moments are
        those of the last finite step."""
        n = self.nonfinite_steps()
        if collective and self._guard is not None and self._collectives and self.world > 1:
            flag = torch.tensor([float(n)], dtype=torch.float32, device=self.device)
            if torch.distributed.get_backend(self.group) != "nccl":
                flag = flag.cpu()
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=self.group)
            n = int(flag.item())
        if n:
            raise FloatingPointError(f"loss or gradients were not finite in {n} optimizer step(s): those updates were skipped on the "
                                     "device; stopping as the reference does (tools/finetune_tools.py:533-541)")

    def _check_same_task(self, task: str) -> None:
        import hashlib
        code = int.from_bytes(hashlib.sha1(task.encode()).digest()[:7], "little")
        mine = torch.tensor([[float(code % (1 << 24)), float((code >> 24) % (1 << 24))]], dtype=torch.float32, device=self.device)
        from .parallel import all_gather_rows
        allr = all_gather_rows(mine, group=self.group, at_world_1=True).cpu()
        if not bool((allr == allr[0]).all()):
            raise RuntimeError(f"rank {self.rank} runs task {task!r} in this micro-step but another rank runs a different one: every rank "
                               "must schedule the same task per micro-step (sampler.py:218-337), or its collectives do not match")

    # ---- the step ---------------------------------------------------------------------------------------
    def _stream(self) -> int:
        return nat.current_stream_handle(self.device)

    def sync_weights(self) -> None:
        with torch.cuda.device(self.device):
            nat.check(nat.lib.sf_trainer_sync_weights(self._h, self.params.data_ptr(), self._stream()))

    def _workspace(self, B: int, T: int) -> torch.Tensor:
        if self._ws_key != (B, T):
            n = C.c_size_t()
            nat.check(nat.lib.sf_trainer_workspace_bytes(self._h, B, T, C.byref(n)))
            self._ws = None
            self._ws = torch.empty(n.value, dtype=torch.uint8, device=self.device)
            self._ws_key = (B, T)
        return self._ws

    def forward(self, pixel_values: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Forward with saved activations.  Returns (last_hidden_state [B,T,N,D], pooler_output [B,T,D])."""
        x = pixel_values
        if x.device != self.device:
            raise RuntimeError("pixel_values must already be on the trainer's GPU")
        if x.dtype not in (torch.float32, torch.bfloat16, torch.uint8):
            x = x.float()
        x = x.contiguous()
        B, T, Cc, H, W = x.shape
        c = self.config
        if (Cc, H, W) != (c.num_channels, c.image_size, c.image_size):
            raise ValueError(f"training takes {c.num_channels}x{c.image_size}x{c.image_size} frames, got {Cc}x{H}x{W}")
        N = (H // c.patch_size) * (W // c.patch_size)
        ws = self._workspace(B, T)
        if self.drop_path_rate > 0.0 and self.drop_path:
            self.last_drop_path = drop_path_factors(self.drop_path_rate, c.num_hidden_layers, B, T, N, self._dp_gen)
            self._dp_dev = self.last_drop_path.to(self.device, non_blocking=False)       # kept alive until the backward has run
            nat.check(nat.lib.sf_trainer_set_drop_path(self._h, self._dp_dev.data_ptr(), B, T))
        else:
            self.last_drop_path = None
            nat.check(nat.lib.sf_trainer_set_drop_path(self._h, None, 0, 0))
        if (self.hidden_dropout > 0.0 or self.attention_dropout > 0.0) and self.dropout:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=self._dp_gen))
            self.last_dropout = (seed, self.hidden_dropout, self.attention_dropout)
            nat.check(nat.lib.sf_trainer_set_dropout(self._h, self.hidden_dropout, self.attention_dropout, seed))
        else:
            self.last_dropout = None
            nat.check(nat.lib.sf_trainer_set_dropout(self._h, 0.0, 0.0, 0))
        lhs = torch.empty(B, T, N, c.hidden_size, dtype=torch.float32, device=self.device)
        pool = torch.empty(B, T, c.hidden_size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(nat.lib.sf_trainer_forward(self._h, x.data_ptr(), {torch.bfloat16: nat.SF_BF16, torch.uint8: nat.SF_U8}.get(x.dtype, nat.SF_F32),
                                                 B, T, lhs.data_ptr(), pool.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        self._lhs, self._pooler = lhs, pool
        return lhs, pool

    def backward(self, d_pooler: torch.Tensor, d_last_hidden: Optional[torch.Tensor] = None, reduce: bool = False) -> None:
        """grads += d loss / d params.  ``reduce``: all-reduce each bucket as soon as its stages are enqueued."""
        dp = d_pooler.to(torch.float32).contiguous()
        dl = None if d_last_hidden is None else d_last_hidden.to(torch.float32).contiguous()
        ws = self._ws
        if ws is None or self._pooler is None:
            raise RuntimeError("backward() needs a preceding forward() (its saved activations live in the workspace)")
        works = []
        nst = len(self.stage_ranges)
        with torch.cuda.device(self.device):
            first = 0
            reduce = reduce and self._collectives and self.comm_enabled
            plan = self.buckets if reduce else [(nst - 1, 0, 0)]
            for last, off, n in plan:
                nat.check(nat.lib.sf_trainer_backward(self._h, dp.data_ptr(), nat.ptr(dl), self.grads.data_ptr(), first, last,
                                                      ws.data_ptr(), ws.numel(), self._stream()))
                first = last + 1
                if reduce:
                    sl = self.grads[off: off + n]
                    if self.grad_reduce_dtype == "bf16":
                        # half the bytes on the xGMI links (204 MB instead of 407 MB per step for SigLIP-base);
                        # the sum over <= 8 ranks is taken in bf16, the optimizer state stays fp32
                        # persistent wire buffer (no allocation per bucket and step); the copy back after wait() stays: the clip /
                        # guard passes and AdamW read the fp32 buffer
                        if self._wire is None:
                            self._wire = torch.empty(self.n_train, dtype=torch.bfloat16, device=self.device)
                        half = self._wire[off: off + n]
                        half.copy_(sl)
                        works.append((torch.distributed.all_reduce(half, group=self.group, async_op=True), sl, half))
                    else:
                        works.append((torch.distributed.all_reduce(sl, group=self.group, async_op=True), None, None))
            if reduce and self._loss_acc_used and self.world > 1:
                # the window's loss sum rides along: every rank's guard sees the same (possibly non-finite) value
                la = self._loss_acc if torch.distributed.get_backend(self.group) == "nccl" else self._loss_acc.cpu()
                works.append((torch.distributed.all_reduce(la, group=self.group, async_op=True), self._loss_acc if la is not self._loss_acc else None, la))
        for w, sl, half in works:
            w.wait()
            if half is not None and sl is not None:
                sl.copy_(half)

    def time_bucket_allreduce(self, iters: int = 3) -> float:
        """Milliseconds for ONE round of the bucket all-reduces of a step on an otherwise idle GPU (HIP events on the
        current stream around `iters` rounds; a scratch buffer stands in for the gradients)."""
        if not self._collectives:
            return 0.0
        wire = torch.bfloat16 if self.grad_reduce_dtype == "bf16" else torch.float32
        buf = torch.zeros(self.n_train, dtype=wire, device=self.device)

        def one_round():
            for _, off, n in self.buckets:
                torch.distributed.all_reduce(buf[off: off + n], group=self.group)
        one_round()
        torch.cuda.synchronize(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            one_round()
        e1.record()
        torch.cuda.synchronize(self.device)
        return e0.elapsed_time(e1) / iters

    def loss_and_grad(self, task: str, pooler: torch.Tensor, task_input: dict):
        """Task-head loss (HIP kernels of heads.py) -> (loss [1], d loss/d pooler, d loss/d (scale, bias)).

        The heads read ``logit_scale`` / ``logit_bias`` straight from the flat parameter buffer on the device:
        nothing here synchronises with the host, so the next micro-step can be enqueued behind this one.
        Retrieval with world > 1 uses every rank's captions as negatives (the reference's distributed SigLipLoss,
        modeling:239-297) unless ``task_input["gather_negatives"]`` is False.
        Localization takes either one table for the whole batch (``label_emb`` [L, D]) or, as the reference head does, a dataset
        name per clip (``datasets``: B names, ``label_embs``: {name: [L_name, D]}) with tables of different sizes."""
        from .heads import LocalizationHead, RetrievalHead
        ls, lb = self._view(f"task_heads.{task}.logit_scale"), self._view(f"task_heads.{task}.logit_bias")
        if task_input["kind"] == "retrieval":
            text = task_input["text"].to(self.device)
            rank = 0
            if task_input.get("gather_negatives", True) and self._collectives:
                from .parallel import all_gather_rows
                text = all_gather_rows(text.contiguous(), group=self.group, at_world_1=True)
                rank = self.rank
            return RetrievalHead(ls, lb).loss(pooler, text, rank=rank)
        if "datasets" in task_input:
            # The reference head walks the batch sample by sample with each clip's own dataset table (modeling:2250-2276: tables of
            # different L in one batch): group the clips by dataset, one loss launch per table on that group's rows, weighted by the
            # group's share of the batch — loss = mean over clips — and scatter the gradients back to the clips' rows.
            names = list(task_input["datasets"])
            tables = task_input["label_embs"]
            labels = task_input["labels"]
            B = pooler.shape[0]
            if len(names) != B:
                raise ValueError(f"{len(names)} dataset names for {B} clips")
            loss = torch.zeros(1, dtype=torch.float32, device=self.device)
            gp = torch.zeros_like(pooler, dtype=torch.float32)
            gs = torch.zeros(2, dtype=torch.float32, device=self.device)
            for name in dict.fromkeys(names):
                idx = [i for i, d in enumerate(names) if d == name]
                w = len(idx) / B
                whole = len(idx) == B
                ix = None if whole else torch.tensor(idx, device=self.device)
                l_g, gp_g, gs_g = LocalizationHead(tables[name], ls, lb).loss(pooler if whole else pooler.index_select(0, ix),
                                                                              labels if whole else labels.to(self.device).index_select(0, ix))
                loss.add_(l_g, alpha=w)
                gs.add_(gs_g, alpha=w)
                if whole:
                    gp.add_(gp_g, alpha=w)
                else:
                    gp.index_add_(0, ix, gp_g, alpha=w)
            return loss, gp, gs
        return LocalizationHead(task_input["label_emb"], ls, lb).loss(pooler, task_input["labels"])

    def micro_step(self, task: str, pixel_values: torch.Tensor, task_input: dict, update_freq: int = 1,
                   lr: Optional[float] = None, weight_decay: Optional[float] = None,
                   clip_grad: Optional[float] = None) -> torch.Tensor:
        """One micro-batch of ``train_one_epoch_multi_task``; returns the (unscaled) loss tensor [1]."""
        if self._collectives and (self.task_sync_check == "always" or (self.task_sync_check == "first" and not self._task_checked)):
            self._check_same_task(task)         # before the first collective of the micro-step
            self._task_checked = True
        _, pooler = self.forward(pixel_values)
        loss, gp, gs = self.loss_and_grad(task, pooler, task_input)
        # d loss / d (scale, bias) into the two scalar slots of this head: ONE in-place multi-tensor add with the 1 / update_freq
        # factor folded in (no temporaries, no separate scaling launches); d pooler is scaled only when update_freq > 1
        inv = 1.0 / update_freq
        torch._foreach_add_([self.grad(f"task_heads.{task}.logit_scale"), self.grad(f"task_heads.{task}.logit_bias")],
                            [gs[0], gs[1]], alpha=inv)
        self._touched.add(task)
        self._last_loss = loss
        if update_freq > 1 or (self._collectives and self.world > 1):
            torch.add(self._loss_acc, loss.reshape(1), out=self._loss_acc)       # inf / NaN survive the sum
            self._loss_acc_used = True
        self.micro += 1
        last = self.micro % update_freq == 0
        self.backward(gp if update_freq == 1 else gp.mul_(inv), reduce=last)
        if last:
            self.optimizer_step(lr=lr, weight_decay=weight_decay, clip_grad=clip_grad)
        return loss

    def grad_norm(self) -> torch.Tensor:
        with torch.cuda.device(self.device):
            nat.check(nat.lib.sf_trainer_grad_sumsq(self._h, self.grads.data_ptr(), self._scratch.data_ptr(), self._stream()))
        return self._scratch[0].sqrt() / self.world

    def optimizer_step(self, lr: Optional[float] = None, weight_decay: Optional[float] = None,
                       clip_grad: Optional[float] = None) -> None:
        """AdamW on the trainable prefix (gradient cleared by the same kernel), then refresh the bf16 working
        weights.  ``clip_grad``: torch.nn.utils.clip_grad_norm_ semantics, coefficient computed on the device."""
        if self.exp_avg is None:
            raise RuntimeError("this trainer was built without optimizer state (with_optimizer=False)")
        self.step_count += 1
        scale = 1.0 / self.world                                  # all-reduce summed; DDP averages
        sumsq = 0
        active = ()
        if self.task_heads:
            # heads that received a gradient since the last step (micro_step records them; a caller that adds head
            # gradients by hand and never says which gets the plain behaviour: every head steps)
            active = self._touched or set(self.task_heads)
            steps = []
            for t in self.task_heads:
                if t in active:
                    self.head_steps[t] += 1
                steps += [self.head_steps[t] if t in active else 0] * 2
            nat.check(nat.lib.sf_trainer_set_extra_steps(self._h, (C.c_int32 * len(steps))(*steps), len(steps)))
            self._touched = set()
        with torch.cuda.device(self.device):
            if self._guard is not None:
                ll = self._loss_acc if self._loss_acc_used else self._last_loss
                nat.check(nat.lib.sf_trainer_set_nonfinite_guard(
                    self._h, self._guard_row(active).data_ptr(), ll.data_ptr() if (ll is not None and ll.dtype == torch.float32 and ll.is_cuda) else None))
            if clip_grad is not None:
                nat.check(nat.lib.sf_trainer_grad_sumsq(self._h, self.grads.data_ptr(), self._scratch.data_ptr(), self._stream()))
                sumsq = self._scratch.data_ptr()
            nat.check(nat.lib.sf_trainer_adamw_step(
                self._h, self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                self.step_count, self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps,
                self.weight_decay if weight_decay is None else weight_decay, scale, sumsq,
                float(clip_grad) if clip_grad is not None else 0.0, 1, self._stream()))
        # the guard has read its loss: a later optimizer_step() without a micro_step() must not re-check a stale one
        self._last_loss = None
        if self._loss_acc_used:
            self._loss_acc.zero_()
            self._loss_acc_used = False
        self.sync_weights()

    def zero_grad(self) -> None:
        self.grads.zero_()
        self.micro = 0
        self._touched = set()
        self._loss_acc.zero_()
        self._loss_acc_used = False
