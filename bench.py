#!/usr/bin/env python
"""bench.py — frames/s of the StreamFormer encoder forward on 16x224^2 clips (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run)

A "step" is one forward of the hot path over one batch of 8 synthetic clips per GPU
(BASELINE.json configs[1]: "1xMI355X bf16 forward, batch 8x16x224^2, SigLIP-base"), inputs already
resident in HBM.  Clips are independent, so N GPUs run N data-parallel replicas with no data-path
collective ("scaling": "weak"); the only collectives here are the start/stop barriers and the
max-over-ranks reduction of the elapsed time.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the dominant kernel of the forward (sf_gemm_panel_kernel, the N = 768 residual projections: ~38 % of kernel
                time, MFMA-bound at K = 3072, HBM-bound at K = 768): algorithmic FLOPs of its three launches per layer over
                their launch times measured with HIP events on the launch stream; the 256^2 kernel's shapes under other_gemms
  attention     the two attention kernels against the HBM roofline ("fraction of the attention roofline")
  accuracy      max-abs deviation of last_hidden_state / pooler_output vs the CPU oracle, both modes
  accurate_mode frames/s of the fp32-accurate (bf16x3) mode on the same workload
  cpu_baseline  the CPU oracle (a restatement pinned against the reference, kind "port") timed on the
                host cores on a bounded sample: B=1 clips of the same shape
  h2d_inclusive frames/s with the host -> device copy of the batch inside every step (pinned memory), for fp32 frames
                and for raw uint8 frames (normalisation fused into the patch kernel); never the headline `value`
  streaming     BASELINE configs[4]: per-frame latency (p50 / p99) of the KV-cached streaming path, 64-frame online
                clip at B = 1, the achieved HBM rate against the algorithmic bytes of a frame, `first_pass` (a fresh cache)
                and `roofline_streaming` (its dominant kernel, HIP-event timed)
  latency_b1    README.md:55-71 is one clip per call: ms per call at 1 / 2 / 4 clips (bf16) and 1 clip (fp32-accurate)
  train_step    BASELINE configs[2]: frames/s of one multitask pre-training step (forward + loss + backward +
                AdamW) on the same clip shape, with its own CPU baseline; `--mode train` makes that step THE
                timed step (configs[2] at N=1, configs[3] with the gradient all-reduce at N>1)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0         # HBM3E spec, MI355X_MICROARCH.md
GFLOP_PER_FRAME = 49.40       # 790.48 GFLOP per 16-frame clip


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (BASELINE config: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = usable host cores (affinity, cgroup quota)")
    ap.add_argument("--profile", action="store_true", help="timed steps + roofline hooks only (for rocprofv3 runs)")
    ap.add_argument("--streams", type=int, default=1, help="batches in flight: consecutive steps alternate over this many HIP streams")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--same-device", action="store_true", help="dry run of the N>1 path with every rank on cuda:0")
    ap.add_argument("--init-dist", action="store_true",
                    help="N=1 only: create the process group anyway (world_size 1) and run every collective of the N>1 path "
                         "through it — barriers, max-over-ranks, bucketed gradient all-reduce, caption all-gather; the way to "
                         "execute the RCCL branches on a one-GPU box")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="forward = BASELINE configs[1] (the headline metric); train = configs[2]/[3]: one multitask "
                         "pre-training step (forward + loss + backward + gradient all-reduce + AdamW) per GPU batch")
    ap.add_argument("--no-train", action="store_true", help="forward mode: skip the short training-step measurement")
    return ap.parse_args()


def usable_cores() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def timed_steps(model, x, steps, warmup, dist, world, nstreams=1):
    """K forwards over batches of clips.  With nstreams > 1, consecutive (independent) steps are issued
    round-robin on separate HIP streams, each with its own workspace: the HBM-bound phases of one batch
    overlap the MFMA-bound phases of the other.  Still K steps of the same batch size."""
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream()]
    def step(i):
        if nstreams > 1:
            with torch.cuda.stream(streams[i % nstreams]):
                model(x)
        else:
            model(x)
    # settle (untimed, in front of the W warm-up steps): a device that comes out of idle — or out of a rocprofv3 counter pass, which leaves
    # it in the profiling power state for a few seconds — runs its first tenths of a second at a lower clock (a 9.6 ms headline was
    # measured that way right behind the PMC passes of profiles/collect_r06.sh, with 7.96 ms seconds later in the same process)
    t_settle = time.perf_counter()
    j = 0
    while time.perf_counter() - t_settle < 1.5:
        step(j); j += 1
        if j % 16 == 0:
            torch.cuda.synchronize()
    for i in range(max(warmup, nstreams)):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return max_over_ranks(dt, dist)


def power_under_load(model, x, seconds=2.5):
    """Package power (W) and shader clock (MHz) from rocm-smi while the forward runs back to back: the forward is bound by the
    power envelope (docs/history.md H.4), this is the evidence on the line itself.  None if rocm-smi is absent / unreadable."""
    import json as _json, shutil, subprocess, threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    samples = []
    def poll():
        time.sleep(0.8 * seconds / 2.5)
        for _ in range(3):
            try:
                r = subprocess.run([exe, "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10)
                d = _json.loads(r.stdout)
                card = d[sorted(d.keys())[0]]
                w = [float(v) for k, v in card.items() if "Power (W)" in k and "Max" not in k]
                cap = [float(v) for k, v in card.items() if "Max Graphics Package Power" in k]
                sclk = [v for k, v in card.items() if k.startswith("sclk")]
                mhz = float("".join(ch for ch in str(sclk[0]) if ch.isdigit() or ch == ".")) if sclk else None
                if w:
                    samples.append((w[0], mhz, cap[0] if cap else None))
            except Exception:
                pass
            time.sleep(0.3)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    with torch.no_grad():
        while th.is_alive() or time.perf_counter() - t0 < seconds:
            for _ in range(10):
                model(x)
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > 20:
                break
    th.join()
    if not samples:
        return None
    return {"package_W": round(sum(s[0] for s in samples) / len(samples), 1), "cap_W": samples[0][2],
            "sclk_MHz": round(sum(s[1] for s in samples if s[1]) / max(1, sum(1 for s in samples if s[1])), 0) if any(s[1] for s in samples) else None,
            "samples": len(samples), "source": "rocm-smi --showpower --showclocks, polled under the back-to-back forward (rank 0)"}



def max_over_ranks(dt, dist):
    """MAX all-reduce of the elapsed time (on the GPU for RCCL, on the host for gloo)."""
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def ranks_seen(dist, rank, dev):
    """What every rank of the job says about itself, gathered on all ranks (one all_gather_object)."""
    me = {"rank": rank, "device": int(dev.index or 0), "gpu": torch.cuda.get_device_name(dev), "pid": os.getpid()}
    if dist is None:
        return [me]
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, me)
    return got


# forward + input-gradient + weight-gradient products, minus the weight-gradient GEMMs of the frozen spatial
# qkv / output.dense (12 x (11.098 + 3.699) GF per clip; their rank-32 LoRA factors cost < 1 %): SURVEY.md §8(d)
TRAIN_GFLOP_PER_FRAME = (3 * 790.48 - 12 * (11.098 + 3.699)) / 16


def dp_consistency_check(tr, task, x, ti, dist, world):
    """BASELINE configs[3] invariants, checked on the job itself (every rank calls this; collectives inside):
    (1) after the timed optimizer steps every rank holds bit-identical parameters (DDP's contract,
    run_finetuning_multi_task.py:421-423); (2) the gradient buffer the bucketed all-reduce leaves equals the SUM of the
    ranks' local gradients (the optimizer kernel divides by world), compared on a strided ~1M-element sample that
    crosses every bucket: the local gradients come from a backward with the collectives off, are all-gathered
    independently of the bucket path, and summed in float64."""
    from streamformer_amd.parallel import all_gather_rows
    dev = tr.device
    bits = tr.params.view(torch.int32).to(torch.int64)
    w = (torch.arange(bits.numel(), device=dev, dtype=torch.int64) % 65521) + 1
    mine = (int(bits.sum().item()), int((bits * w).sum().item()))          # order-free + position-weighted checksum of the raw bits
    del bits, w
    sums = [None] * world
    dist.all_gather_object(sums, mine)
    idx = torch.arange(0, tr.n_train, max(1, tr.n_train // (1 << 20)), device=dev)

    def one_backward(reduce):
        tr.zero_grad()
        _, pooler = tr.forward(x)
        _, gp, _ = tr.loss_and_grad(task, pooler, ti)
        tr.backward(gp, reduce=reduce)
        g = tr.grads[idx].clone()
        tr.zero_grad()
        return g

    # The comparison needs two backward passes of the same rank to agree bit for bit.  They do on a device of their own (150 of 150 passes,
    # tools/train_det.py).  With several ranks SHARING one device (the same-device dry runs of the test suite) the pooling head's forward
    # kernels were disturbed by another process's temporal attention backward on the same CU ("Device sharing"); they now keep
    # their CUs to themselves, and as a belt to those braces a mismatch is still re-measured, up to three attempts, every rank deciding
    # on the all-reduced maximum; the first attempt's error is reported beside the last.
    errs = []
    for attempt in range(3):
        local = one_backward(False)
        reduced = one_backward(True)
        want = all_gather_rows(local[None].contiguous(), group=tr.group, at_world_1=True).double().sum(0)
        scale = float(want.abs().max())
        diff = (reduced.double() - want).abs()
        err = float(diff.max()) / max(scale, 1e-30)
        errs.append(err)
        if max_over_ranks(err, dist if world > 1 else None) <= 1e-5:
            break
    local2 = one_backward(False)                # (3) the backward itself repeats bit for bit
    repeat_diff = (local2 - local).abs()

    def owner(flat_index):                      # the trainable parameter a position of the gradient buffer belongs to
        for name, e in tr.layout.items():
            if e["trainable"] and e["offset"] <= flat_index < e["offset"] + e["numel"]:
                return name
        return "?"
    worst = {"param": owner(int(idx[int(diff.argmax())])), "abs": float(diff.max())}
    rep = {"max_abs": float(repeat_diff.max()), "elements": int((repeat_diff > 0).sum()),
           "param": owner(int(idx[int(repeat_diff.argmax())])) if float(repeat_diff.max()) > 0 else None}
    # which buckets the sample touched
    touched = sorted({b for b, (_, off, n) in enumerate(tr.buckets) if bool(((idx >= off) & (idx < off + n)).any())})
    return {"params_identical_on_all_ranks": all(s == sums[0] for s in sums), "param_checksums_distinct": len(set(sums)),
            "reduced_equals_sum_of_local_rel_err": err, "sampled_elements": int(idx.numel()), "buckets_sampled": len(touched),
            "buckets": len(tr.buckets), "ways": world, "local_grad_absmax": float(local.abs().max()), "task": task,
            "worst_element": worst, "local_backward_repeat": rep, "attempts": len(errs), "first_attempt_rel_err": errs[0],
            "how": "params: int64 checksums of the raw fp32 bits, all_gather_object; gradients: bucketed all-reduce result vs float64 sum "
                   "of the all-gathered local gradients on a strided sample"}


def train_bench(args, dev, dist, world, rank, steps, warmup, with_cpu):
    """Time `steps` training micro-steps (update_freq = 1) of the LoRA recipe: SigLIP-base, add_lora_spatial,
    spatial base weights frozen, tasks alternating retrieval / localization, AdamW, gradients all-reduced."""
    import streamformer_amd as sa
    from streamformer_amd.training import StreamformerTrainer, scaled_lr
    cfg = sa.siglip_base(add_lora_spatial=True)
    sd = sa.make_state_dict(cfg, seed=0, lora=True)
    B, T, D = args.batch, cfg.num_frames, cfg.hidden_size
    tr = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], freeze_spatial=True, device=dev,
                             lr=scaled_lr(2e-5, B, 1, world), weight_decay=0.05,
                             collectives_at_world_1=dist is not None and world == 1)
    g = torch.Generator().manual_seed(2000 + rank)
    x = torch.randn(B, T, 3, cfg.image_size, cfg.image_size, generator=g).to(dev)
    lab = torch.randn(20, D, generator=g)
    lab = (lab / lab.norm(dim=-1, keepdim=True)).to(dev)
    tasks = [("retrieval", {"kind": "retrieval", "text": torch.randn(B, D, generator=g).to(dev)}),
             ("localization", {"kind": "localization", "label_emb": lab,
                               "labels": torch.randint(-1, 20, (B, T), generator=g).to(dev)})]

    def run(n, w):
        out = []
        for i in range(w):
            tr.micro_step(tasks[i % 2][0], x, tasks[i % 2][1])
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            out.append(tr.micro_step(tasks[i % 2][0], x, tasks[i % 2][1]))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return max_over_ranks(time.perf_counter() - t0, dist), out

    dt, losses = run(steps, warmup)
    value = world * B * T * steps / dt
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2):
        tr.micro_step(tasks[i % 2][0], x, tasks[i % 2][1])
    host_ms = 1e3 * (time.perf_counter() - t0) / 2          # wall time to ENQUEUE one micro-step (the GPU runs behind)
    torch.cuda.synchronize()
    per_task = {}
    for i, l in enumerate(losses):                 # the two tasks have different loss scales: report them apart
        per_task.setdefault(tasks[i % 2][0], []).append(round(float(l), 4))
    res = {"value": round(value, 1), "unit": "frames/s", "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
           "clips_per_gpu": B, "host_enqueue_ms_per_step": round(host_ms, 3),
           "losses_per_task_first_last": {k: [v[0], v[-1]] for k, v in per_task.items()},
           "trainable_params": int(sum(e["numel"] for e in tr.layout.values() if e["trainable"])),
           "grad_allreduce_MB": round(tr.n_train * 4 / 1e6, 1), "allreduce_buckets": len(tr.buckets),
           "workspace_GiB": round(tr._ws.numel() / 2**30, 2),
           "e2e_mfma_frac": round(value / world * TRAIN_GFLOP_PER_FRAME / 1e3 / PEAK_BF16_TFLOPS, 4),
           "recipe": "SigLIP-base + LoRA r=32 on spatial attention, spatial base frozen, retrieval/localization alternating, "
                     "AdamW (fp32 master weights, bf16 MFMA operands), update_freq 1"}
    if dist is not None:
        # data-parallel invariants of the job itself, BEFORE the no-collective timing below lets the ranks' parameters drift apart
        try:
            res["dp_check"] = dp_consistency_check(tr, "localization", x, tasks[1][1], dist, world)
        except Exception as e:          # symmetric on all ranks (no rank-dependent branch inside), so no rank is left in a collective
            res["dp_check"] = {"error": repr(e)}
        # How much of the gradient all-reduce is exposed: the same K steps with the collectives switched off (every rank
        # steps on its local gradients), and the bucket all-reduces alone on an idle GPU, HIP-event timed.
        n2 = max(2, min(steps, 6))
        tr.comm_enabled = False
        dt_nc, _ = run(n2, 1)
        tr.comm_enabled = True
        iso = tr.time_bucket_allreduce(iters=3)
        iso_ms = max_over_ranks(iso, dist)
        step_ms, nocomm_ms = 1e3 * dt / steps, 1e3 * dt_nc / n2
        exposed = max(0.0, step_ms - nocomm_ms)
        nbytes = tr.n_train * (2 if tr.grad_reduce_dtype == "bf16" else 4)
        res["allreduce_ms"] = {"isolated": round(iso_ms, 3), "exposed": round(exposed, 3),
                               "overlapped": round(max(0.0, iso_ms - exposed), 3),
                               "step_ms_without_collectives": round(nocomm_ms, 3),
                               "wire_dtype": tr.grad_reduce_dtype,
                               "busbw_GBps": round(2.0 * (world - 1) / max(world, 1) * nbytes / max(iso_ms, 1e-6) / 1e6, 1),
                               "how": "isolated = HIP-event time of the bucket all-reduces on an idle GPU (max over ranks); exposed = "
                                      "step time minus the same steps with the collectives switched off; overlapped = isolated - exposed",
                               # what `isolated` is to be read against: an all-reduce moves 2 (N - 1) / N x bytes per GPU; a ring drives ONE
                               # xGMI link per direction (~153 GB/s), a direct reduce-scatter + all-gather all N - 1 links at once
                               "bound": {"bytes_per_rank": nbytes, "xgmi_link_GBps": 153.0,
                                         "ring_ms": round(2.0 * (world - 1) / max(world, 1) * nbytes / 153e9 * 1e3, 3),
                                         "direct_ms": round(2.0 / max(world, 1) * nbytes / 153e9 * 1e3, 3)}}
        res["backend"] = dist.get_backend()
    if with_cpu and rank == 0:
        # CPU baseline of the same step: the oracle's autograd + torch.optim.AdamW on ONE clip (bounded sample)
        from oracle import train_oracle as TO
        cores = args.cpu_threads or usable_cores()
        torch.set_num_threads(cores)
        orc = TO.OracleTrainer(sd, cfg, ["retrieval", "localization"], freeze_spatial=True, lr=1e-5)
        x1 = x[:1].cpu()
        ti = {"kind": "localization", "label_emb": lab.cpu(), "labels": tasks[1][1]["labels"][:1].cpu()}
        t0 = time.perf_counter(); orc.micro_step("localization", x1, ti); t1 = time.perf_counter() - t0
        ts = []
        for _ in range(max(1, min(3, int(20.0 / max(t1, 1e-3))))):
            t0 = time.perf_counter(); orc.micro_step("localization", x1, ti); ts.append(time.perf_counter() - t0)
        ts.sort()
        res["cpu_baseline"] = {"value": round(T / ts[len(ts) // 2], 2), "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": f"{len(ts)} training steps of one [1,16,3,224,224] clip (torch autograd + AdamW, fp32) after 1 warm-up"}
        res["speedup_vs_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
    del tr
    torch.cuda.empty_cache()
    return res


def streaming_bench(dev):
    """Per-frame latency of the streaming path: num_frames = 64, B = 1, one frame per
    call, cache reset between repeats; p50 / p99 over the frames of 3 timed repeats, and the achieved HBM rate
    against the algorithmic bytes of a frame (weights 255 MB + KV read 7.225 MB x (t+1) + KV write + activations).
    `first_pass`: the same 64 calls on a FRESH cache (lazy kernel set-up on frame 0, the one graph capture on frame 1)."""
    import streamformer_amd as sa
    from streamformer_amd import _native as nat
    cfg = sa.siglip_base(num_frames=64)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
    m.load_state_dict(sa.make_state_dict(cfg, seed=0))
    m.to(dev)
    x = torch.randn(1, 64, 3, cfg.image_size, cfg.image_size, generator=torch.Generator().manual_seed(64)).to(dev)
    m(x[:, :2])                              # library warm-up outside the stream under test
    cache = m.new_cache(1, 64)
    lat, first = [], []
    for rep in range(4):
        cache.reset()
        for t in range(64):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m(x[:, t:t + 1], use_cache=True, past_key_values=cache)
            torch.cuda.synchronize()
            (lat if rep else first).append(time.perf_counter() - t0)
    lat.sort()
    mean = sum(lat) / len(lat)
    fs = sorted(first)
    gb = (255.0 + 7.225 * (64 + 1) / 2 + 7.225 + 12.0) / 1e3          # mean algorithmic GB per frame over t = 0..63
    # the dominant kernel of a streamed frame (rocprof: profiles/r03_streaming_kernel_stats.txt): the K-parallel skinny GEMM of
    # the N = 768 residual projections, 37 launches per frame; HIP-event timed here at M = 196 back to back (weights L2-warm:
    # inside the stream each launch starts on cold weights and measures ~6.4 us)
    ws = torch.randn(1 << 25, dtype=torch.bfloat16, device=dev).view(torch.uint8)
    ms, fl = nat.C.c_float(), nat.C.c_double()
    N1 = cfg.num_patches
    dom = {}
    for which, name, K in ((3, "out_proj_K768", 768), (1, "mlp_down_K3072", 3072)):
        nat.check(nat.lib.sf_bench_gemm(m._handle, N1, which, 50, ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev),
                                        nat.C.byref(ms), nat.C.byref(fl)))
        by = 768 * K * 2 + N1 * K * 2 + N1 * 768 * (4 + 4 + 2)          # W + A + fp32 residual in / out + bf16 copy
        dom[name] = {"us": round(1e3 * ms.value, 2), "algorithmic_MB": round(by / 1e6, 2), "GBps": round(by / ms.value / 1e6, 1),
                     "frac_of_hbm_peak": round(by / ms.value / 1e6 / PEAK_HBM_GBS, 4)}
    del ws
    del cache
    # what a chain of dependent launches costs on this box when the kernels do nothing: the same number of EMPTY 256-workgroup
    # kernels in one hipGraph: the floor a ~100-launch frame sits on, next to its 64 us of bytes at the HBM peak
    floor = None
    try:
        us = nat.C.c_float()
        nat.check(nat.lib.sf_bench_launch_floor(int(dev.index or 0), 104, 50, nat.current_stream_handle(dev), nat.C.byref(us)))
        floor = {"us_per_launch": round(us.value, 3), "launches_per_frame": 104, "ms_per_frame_of_empty_launches": round(104 * us.value / 1e3, 4),
                 "how": "104 dependent empty kernels (256 x 256 threads) in one hipGraph, 50 replays, HIP events (sf_bench_launch_floor)"}
    except Exception as e:
        floor = {"error": repr(e)}
    # the serving shape of the vision tower (vqa_enc:1494-1500: one cache per stream): 8 streams advance one frame per call
    S = 8
    xs = x.expand(S, -1, -1, -1, -1).contiguous()
    cache = m.new_cache(S, 64)
    lat8 = []
    for rep in range(2):
        cache.reset()
        for t in range(64):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m(xs[:, t:t + 1], use_cache=True, past_key_values=cache)
            torch.cuda.synchronize()
            if rep:
                lat8.append(time.perf_counter() - t0)
    lat8.sort()
    del m, cache, xs
    torch.cuda.empty_cache()
    # the DEFAULT compute mode (fp32-accurate, bf16x3) on the same stream: what a caller gets who only changes the import line
    acc_stream = None
    try:
        ma = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="fp32")
        ma.load_state_dict(sa.make_state_dict(cfg, seed=0))
        ma.to(dev)
        ma(x[:, :2])
        cache = ma.new_cache(1, 64)
        lata = []
        for rep in range(3):
            cache.reset()
            for t in range(64):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ma(x[:, t:t + 1], use_cache=True, past_key_values=cache)
                torch.cuda.synchronize()
                if rep:
                    lata.append(time.perf_counter() - t0)
        lata.sort()
        acc_stream = {"p50_ms": round(1e3 * lata[len(lata) // 2], 3), "p99_ms": round(1e3 * lata[int(len(lata) * 0.99)], 3),
                      "dtype": "bf16x3 (fp32-accurate, the default compute_dtype)"}
        del ma, cache
        torch.cuda.empty_cache()
    except Exception as e:
        acc_stream = {"error": repr(e)}
    return {"p50_ms": round(1e3 * lat[len(lat) // 2], 3), "p99_ms": round(1e3 * lat[int(len(lat) * 0.99)], 3),
            "mean_ms": round(1e3 * mean, 3), "frames_per_s": round(1.0 / mean, 1), "algorithmic_GB_per_frame": round(gb, 3),
            "GBps": round(gb / mean, 1), "frac_of_hbm_peak": round(gb / mean / PEAK_HBM_GBS, 4),
            "first_pass": {"p50_ms": round(1e3 * fs[len(fs) // 2], 3), "mean_ms": round(1e3 * sum(first) / len(first), 3),
                           "frame0_ms": round(1e3 * first[0], 3), "frame1_ms": round(1e3 * first[1], 3), "max_ms": round(1e3 * fs[-1], 3),
                           "note": "fresh cache: frame 0 runs eagerly (lazy per-kernel set-up), frame 1 captures the one "
                                   "position-free hipGraph of the cache, every later frame replays it"},
            "config": "SigLIP-base, num_frames=64, B=1, one 224^2 frame per call, bf16 mode, KV-cache of 64 frames",
            "launch_floor": floor,
            "roofline_streaming": {"bound": "hbm", "kernel": "sf_gemm_skinny_kg_kernel (N = 768 residual projections, 36 of 104 launches "
                                   "per frame, ~32 % of its kernel time)", "launches": dom, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "whole_frame": {"achieved": round(gb / mean, 1), "frac": round(gb / mean / PEAK_HBM_GBS, 4)},
                                   "note": "a streamed frame is ~100 dependent launches of 4-10 us: latency-bound, not bandwidth-bound "
                                           "(docs/history.md A.4.1)"},
            "accurate_mode": acc_stream,
            "eight_streams": {"p50_ms_per_call": round(1e3 * lat8[len(lat8) // 2], 3),
                              "frames_per_s": round(S / (sum(lat8) / len(lat8)), 1),
                              "config": "same model, 8 independent streams advance one frame per call (one cache, B = 8)"}}


def small_batch_latency(dev):
    """README.md:55-71 is ONE clip per call: latency of 1 and 2 clips of 16 x 224^2 (bf16 mode), and of one clip in the
    fp32-accurate mode."""
    import streamformer_amd as sa
    cfg = sa.siglip_base()
    sd = sa.make_state_dict(cfg, seed=0)
    out = {}
    for mode, batches in (("bf16", (1, 2, 4)), ("fp32", (1,))):
        m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
        m.load_state_dict(sd)
        m.to(dev).eval()
        for B in batches:
            x = torch.randn(B, cfg.num_frames, 3, cfg.image_size, cfg.image_size, generator=torch.Generator().manual_seed(B)).to(dev)
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                m(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            out[f"{'bf16' if mode == 'bf16' else 'bf16x3'}_B{B}"] = {"ms": round(1e3 * dt, 3), "frames_per_s": round(B * cfg.num_frames / dt, 1)}
        del m
    torch.cuda.empty_cache()
    out["config"] = "SigLIP-base, B clips x 16 x 224^2 per call, 20 back-to-back calls after 3 warm-ups"
    return out


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here (torch.distributed.run, one per GPU,
    rendezvous on 127.0.0.1) with the same command line; rank 0 of the children prints the one JSON line."""
    import socket
    import subprocess
    if not args.same_device and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node "
                         "(--same-device puts every rank on cuda:0 for a dry run)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's P2P buffers over xGMI need it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not launched and args.gpus > 1:
        self_launch(args)
    args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.init_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if not launched:                    # --init-dist at N = 1 without a launcher: a private rendezvous
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            kw = dict(init_method=f"tcp://127.0.0.1:{s.getsockname()[1]}", rank=0, world_size=1)
            s.close()
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(args.backend, **kw)
    seen = ranks_seen(dist, rank, dev)

    import streamformer_amd as sa
    from streamformer_amd import _native as nat

    if args.mode == "train":
        r = train_bench(args, dev, dist, world, rank, args.steps, args.warmup, with_cpu=(world == 1 and not args.no_cpu_baseline))
        if rank == 0:
            out = {"metric": "frames/s (multitask pre-training step, 16x224^2 clips)", "value": r["value"], "unit": "frames/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                   "config": {"workload": f"BASELINE configs[{2 if world == 1 else 3}]: multitask pre-training step, "
                                          f"{args.batch} clips x 16 x 224^2 per GPU, " + r["recipe"],
                              "global_batch_clips": args.batch * world, "parallelism": f"dp{world}",
                              "collective": (f"{'RCCL' if args.backend == 'nccl' else args.backend} all-reduce of the fp32 gradient buffer in "
                                             f"{r['allreduce_buckets']} buckets issued behind the staged backward + one all-gather of the "
                                             "caption features per retrieval step") if dist is not None else "none"}}
            out.update({k: v for k, v in r.items() if k not in ("value", "unit", "ms_per_step", "steps", "recipe")})
            if dist is not None:
                out["ranks_seen"] = seen
            print(json.dumps(out), flush=True)
        if dist is not None and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    cfg = sa.siglip_base()
    sd = sa.make_state_dict(cfg, seed=0)
    B, T = args.batch, cfg.num_frames
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(B, T, 3, cfg.image_size, cfg.image_size, generator=g).to(dev)

    model = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
    model.load_state_dict(sd)
    model.to(dev).eval()
    dt = timed_steps(model, x, args.steps, args.warmup, dist, world, args.streams)
    frames = world * B * T * args.steps
    value = frames / dt
    # host headroom (8 ranks = 8 Python processes on one host): wall time this process needs to ENQUEUE one step (~107 launches
    # through ctypes), measured on a drained stream with the GPU running behind
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        model(x)
    host_ms = 1e3 * (time.perf_counter() - t0) / 4
    torch.cuda.synchronize()

    out = {
        "metric": "frames/s (16x224^2 clips)", "value": round(value, 1), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"SigLIP-base StreamFormer encoder forward, {B} clips x 16 x 224^2 per GPU, "
                               "random-init de-trivialised weights, causal temporal attention",
                   "global_batch_clips": B * world, "frames_per_clip": T, "parallelism": f"dp{world}",
                   "steps_in_flight": args.streams,
                   "collective": "none on the data path (clips are independent); start/stop barriers and the max-over-ranks "
                                 "reduction of the elapsed time only" + (f" [{args.backend}]" if dist is not None else "")},
        "e2e_mfma_frac": round(value / world * GFLOP_PER_FRAME / 1e3 / PEAK_BF16_TFLOPS, 4),
        "host_enqueue_ms_per_step": round(host_ms, 3),
        "streams_in_flight": args.streams,
    }
    if dist is not None:
        out["ranks_seen"] = seen
    if args.streams > 1:
        dt1 = timed_steps(model, x, args.steps, 2, dist, world, 1)
        out["single_stream"] = {"value": round(frames / dt1, 1), "ms_per_step": round(1e3 * dt1 / args.steps, 3)}
    elif not args.profile:
        # beside the headline (one step at a time), the same K steps with TWO batches in flight on two HIP streams (own workspaces):
        # one batch's HBM-bound phases and round tails run under the other's main loops.  Reported, not `value`: a step's latency doubles.
        dt2 = timed_steps(model, x, args.steps, 2, dist, world, 2)
        out["two_steps_in_flight"] = {"value": round(frames / dt2, 1), "ms_per_step": round(1e3 * dt2 / args.steps, 3),
                                      "note": "consecutive steps alternate over two HIP streams; throughput only"}

    # BASELINE configs[3] on the driver's standard command: at N > 1 every rank also runs a short training leg (6 micro-steps of the
    # multitask step with the bucketed RCCL gradient all-reduce + the caption all-gather) right behind the headline forward timing,
    # while all ranks are still in lock step; the forward `value` above is untouched.  north_star scopes the xGMI traffic to exactly
    # this collective, and the forward alone would show an embarrassingly parallel curve.
    train_leg = None
    if world > 1 and not args.no_train and not args.profile:
        del model
        torch.cuda.empty_cache()
        try:
            train_leg = train_bench(args, dev, dist, world, rank, 6, 2, with_cpu=False)
            train_leg["config"] = (f"BASELINE configs[3]: data-parallel multitask pre-training step, {args.batch} clips per GPU x {world} GPUs = "
                                   f"{args.batch * world} clips global, gradient all-reduce in {train_leg.get('allreduce_buckets')} buckets + caption all-gather")
            train_leg["ranks_seen"] = seen
        except Exception as e:
            train_leg = {"error": repr(e)}
        torch.cuda.empty_cache()
        if rank == 0:                       # rank 0's roofline / accuracy legs below use it
            model = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
            model.load_state_dict(sd)
            model.to(dev).eval()
            model(x)                        # weights are uploaded / finalised by the first forward
            torch.cuda.synchronize()
    if args.profile and rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return
    if rank == 0:
        # ---- roofline of the dominant kernel (live HIP-event timing on the launch stream) ------
        M = B * T * cfg.num_patches
        ws = torch.randn(1 << 29, dtype=torch.bfloat16, device=dev).view(torch.uint8)   # 1 GiB of random bf16
        ms, fl = nat.C.c_float(), nat.C.c_double()
        stream = nat.current_stream_handle(dev)
        gemms = {}
        for which, name in ((0, "mlp_up"), (1, "mlp_down"), (2, "qkv"), (3, "out_proj")):
            nat.check(nat.lib.sf_bench_gemm(model._handle, M, which, 20, ws.data_ptr(), ws.numel(), stream,
                                            nat.C.byref(ms), nat.C.byref(fl)))
            gemms[name] = {"ms": round(ms.value, 4), "tflops": round(fl.value / ms.value / 1e9, 1)}
        # The kernel with the largest share of the forward (in-forward rocprof profile, profiles/r02_forward_kernel_stats.txt)
        # is sf_gemm_panel_kernel: per layer two K = 768 launches (temporal / spatial attention output projection) and one
        # K = 3072 launch (MLP down-projection), each with the residual read-modify-write (hi + lo bf16 planes since round 3) and the
        # LayerNorm statistics of the next GEMM in its epilogue.  Its launches are MFMA-work-weighted here exactly as the
        # forward runs them: achieved = algorithmic FLOPs of (2 x out_proj + 1 x mlp_down) / their summed launch times.
        L = cfg.num_hidden_layers
        panel_flop = (2 * 2.0 * M * 768 * 768 + 2.0 * M * 768 * 3072)               # per layer
        panel_ms_isolated = 2 * gemms["out_proj"]["ms"] + gemms["mlp_down"]["ms"]
        # IN-SITU durations: HIP events around this kernel's launches inside real forwards (sf_forward_profile;
        # the event pair also spans the launch boundary in front of the kernel).  Isolated back-to-back launches of one kernel work
        # on an Infinity-Cache-warm 115-190 MB set and read a few percent faster; they stay on the line as `isolated`.
        insitu = None
        try:
            prof = (nat.C.c_float * 8)()
            lhs_p = torch.empty(B, T, cfg.num_patches, cfg.hidden_size, dtype=torch.float32, device=dev)
            pool_p = torch.empty(B, T, cfg.hidden_size, dtype=torch.float32, device=dev)
            nb = nat.C.c_size_t()
            nat.check(nat.lib.sf_workspace_bytes(model._handle, B, T, cfg.image_size, cfg.image_size, nat.C.byref(nb)))
            wsp = torch.empty(nb.value, dtype=torch.uint8, device=dev)
            acc_ms = [0.0] * 4
            cnt = [0] * 4
            reps = 5
            for _ in range(reps):
                nat.check(nat.lib.sf_forward_profile(model._handle, x.data_ptr(), nat.SF_F32, B, T, cfg.image_size, cfg.image_size, lhs_p.data_ptr(),
                                                     pool_p.data_ptr(), wsp.data_ptr(), wsp.numel(), stream, prof))
                for c in range(4):
                    acc_ms[c] += prof[2 * c] * prof[2 * c + 1]
                    cnt[c] += int(prof[2 * c + 1])
            insitu = {n: {"ms": round(acc_ms[c] / max(cnt[c], 1), 4), "launches_per_forward": cnt[c] // reps}
                      for c, n in enumerate(("out_proj_K768", "mlp_down_K3072", "spatial_attention", "temporal_attention"))}
            del wsp, lhs_p, pool_p
        except Exception as e:      # the line still carries the isolated numbers
            insitu = None
            insitu_err = repr(e)
        if insitu and insitu["out_proj_K768"]["launches_per_forward"] and insitu["mlp_down_K3072"]["launches_per_forward"]:
            panel_ms = 2 * insitu["out_proj_K768"]["ms"] + insitu["mlp_down_K3072"]["ms"]
            timing = "in-situ: HIP events around the kernel's launches inside 5 real forwards (sf_forward_profile)"
        else:
            panel_ms = panel_ms_isolated
            timing = "isolated back-to-back launches (in-situ profile unavailable)"
        panel_tflops = panel_flop / panel_ms / 1e9
        # Yardstick: the vendor library (torch.matmul -> hipBLASLt / rocBLAS, bf16, random data) on the same four shapes, PLAIN
        # (no bias / LayerNorm / GELU / residual), in this process right after the forward.  Never on the product path.  Beside it the product
        # kernels' isolated launches, which DO carry their epilogues (out_proj / mlp_down: + 154 MB of residual read-modify-write per launch).
        yardstick = None
        try:
            yardstick = {"note": "torch.matmul(a, w.t()) bf16 on random data, 30 launches between torch.cuda.Event pairs on the current stream; plain GEMM without "
                                 "any epilogue — the product launches beside it include theirs (residual planes in / out + row sums; LayerNorm fold + GELU)"}
            for name, n_, k_ in (("out_proj", 768, 768), ("mlp_down", 768, 3072), ("qkv", 2304, 768), ("mlp_up", 3072, 768)):
                a_ = torch.randn(M, k_, device=dev, dtype=torch.bfloat16)
                w_ = torch.randn(n_, k_, device=dev, dtype=torch.bfloat16)
                for _ in range(5):
                    torch.matmul(a_, w_.t())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    torch.matmul(a_, w_.t())
                e1.record()
                torch.cuda.synchronize()
                vms = e0.elapsed_time(e1) / 30
                yardstick[name] = {"vendor_plain_ms": round(vms, 4), "vendor_plain_tflops": round(2.0 * M * n_ * k_ / vms / 1e9, 1),
                                   "product_fused_ms": gemms[name]["ms"], "product_fused_tflops": gemms[name]["tflops"]}
                if name in ("out_proj", "mlp_down"):      # the same product kernel with a plain bf16 output: like for like
                    nat.check(nat.lib.sf_bench_gemm(model._handle, M, 4 if name == "out_proj" else 5, 20, ws.data_ptr(), ws.numel(), stream,
                                                    nat.C.byref(ms), nat.C.byref(fl)))
                    yardstick[name]["product_plain_ms"] = round(ms.value, 4)
                    yardstick[name]["product_plain_tflops"] = round(fl.value / ms.value / 1e9, 1)
                del a_, w_
        except Exception as e:
            yardstick = {"error": repr(e)}
        traffic, traffic_note, traffic_from = None, "no PMC file", None
        try:   # HBM-side bytes per launch IMPORTED from the committed PMC passes (rocprofv3 cannot run inside the bench)
            pmc_file = next(f for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")
                            if os.path.exists(os.path.join(ROOT, "profiles", f)))
            traffic_from = "profiles/" + pmc_file
            with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                pmc = json.load(f)
            tr = next(v["traffic_bytes_corrected"] for k, v in pmc["kernels"].items() if k.startswith("void sf_gemm_panel_kernel<13>"))
            import hashlib
            hsrc = hashlib.sha256(open(os.path.join(ROOT, "streamformer_amd", "csrc", "sf_gemm_panel.hip"), "rb").read()).hexdigest()[:16]
            if pmc.get("panel_source_sha16") == hsrc:
                traffic = tr
                traffic_note = ("IMPORTED, not measured by this run: bytes/launch, mean over the panel launches of a forward = 2*FETCH_SIZE + WRITE_SIZE "
                                f"({traffic_from}, separate --pmc passes; FETCH includes Infinity-Cache hits); algorithmic bytes/launch: 194e6 at K = 768 "
                                "(A 38.5 + W 1.2 + residual as hi + lo bf16 planes in / out 154 MB; the hi plane is the next GEMM's operand), "
                                "313e6 at K = 3072")
            else:
                traffic_note = f"{traffic_from} was taken on an older sf_gemm_panel.hip: traffic withheld until the PMC passes are re-run"
        except Exception as e:
            traffic_note = f"PMC file unreadable: {e!r}"
        # MFMA-busy share of the dominant kernel's SIMD-cycles (north_star: "MFMA utilisation against gfx950 peak"): IMPORTED from the
        # committed SQ / GRBM counter passes, guarded by the same source hash as `traffic`
        mfma_busy = {"value": None, "note": "no SQ counter file"}
        mfma_busy_other = None
        try:
            sq_name = next(f for f in ("r06_pmc_sq.json", "r05_pmc_sq.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            sqf = os.path.join(ROOT, "profiles", sq_name)
            with open(sqf) as f:
                sq = json.load(f)
            import hashlib
            hsrc = hashlib.sha256(open(os.path.join(ROOT, "streamformer_amd", "csrc", "sf_gemm_panel.hip"), "rb").read()).hexdigest()[:16]
            ent = next(v for k, v in sq["kernels"].items() if k.startswith("void sf_gemm_panel_kernel<13>"))
            if sq.get("panel_source_sha16") == hsrc:
                mfma_busy = {"value": ent.get("mfma_busy_frac_of_simd_cycles"), "effective_clock_GHz": ent.get("effective_clock_GHz_if_counter_sums_8_xcds"),
                             "lds_bank_conflict_over_active_lds": ent.get("lds_bank_conflict_over_active_lds"),
                             "wait_inst_any_over_wave_cycles": ent.get("wait_inst_any_over_wave_cycles"),
                             "imported_from": "profiles/" + sq_name,
                             "note": "IMPORTED, not measured by this run: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), mean over the kernel's "
                                     "dispatches of a forward (K = 768 and K = 3072 launches together); separate rocprofv3 --pmc passes"}
                # the two other big GEMMs of the forward, same passes: LayerNorm-folded qkv projection and MLP up-projection
                mfma_busy_other = {}
                for key, nm in (("void sf_gemm256_kernel<1, true, 224", "qkv_256x224_tiles"), ("void sf_gemm256_kernel<2, true, 256", "mlp_up_256x256_tiles")):
                    e2 = next((v for k, v in sq["kernels"].items() if k.startswith(key)), None)
                    if e2:
                        mfma_busy_other[nm] = {"value": e2.get("mfma_busy_frac_of_simd_cycles"), "effective_clock_GHz": e2.get("effective_clock_GHz_if_counter_sums_8_xcds"),
                                               "lds_bank_conflict_over_active_lds": e2.get("lds_bank_conflict_over_active_lds")}
            else:
                mfma_busy["note"] = f"profiles/{sq_name} was taken on an older sf_gemm_panel.hip: withheld until the counter passes are re-run"
        except Exception as e:
            mfma_busy["note"] = f"SQ counter file unreadable: {e!r}"
        out["roofline"] = {"kernel": "sf_gemm_panel_kernel<13> (N = 768 residual projections: 2 x attention out-proj K=768 + MLP down-proj K=3072 per layer, "
                                     "epilogue = residual read-modify-write on hi + lo bf16 planes + LayerNorm row sums; M=%d)" % M,
                           "bound": "mfma", "achieved": round(panel_tflops, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(panel_tflops / PEAK_BF16_TFLOPS, 4), "mfma_busy": mfma_busy, "traffic": traffic, "traffic_imported_from": traffic_from,
                           "traffic_note": traffic_note, "avg_launch_ms": round(panel_ms / 3, 4), "timing": timing,
                           "in_situ": insitu,
                           "isolated": {"achieved": round(panel_flop / panel_ms_isolated / 1e9, 1), "frac": round(panel_flop / panel_ms_isolated / 1e9 / PEAK_BF16_TFLOPS, 4),
                                        "note": "sf_bench_gemm: 20 back-to-back launches per shape"},
                           # live: the kernel's launches of one forward (per layer 2 x K=768 + 1 x K=3072, + the embedding GEMM, also
                           # K = 768) at their HIP-event times of this run, over this run's ms_per_step
                           "share_of_step_time_live": round((L * panel_ms + gemms["out_proj"]["ms"]) / (1e3 * dt / args.steps), 4),
                           "share_from_profile": "profiles/r06_forward_kernel_stats.txt (rocprofv3 --kernel-trace --stats of this command)",
                           "launches": {"out_proj_K768": gemms["out_proj"], "mlp_down_K3072": gemms["mlp_down"]},
                           "hbm_view_K768": {"algorithmic_GB": 0.1939,
                                             "GBps": round(0.1939 / (insitu["out_proj_K768"]["ms"] if insitu and insitu["out_proj_K768"]["ms"] > 0 else gemms["out_proj"]["ms"]) * 1e3, 1),
                                             "frac_of_hbm_peak": round(0.1939 / (insitu["out_proj_K768"]["ms"] if insitu and insitu["out_proj_K768"]["ms"] > 0 else gemms["out_proj"]["ms"]) * 1e3 / PEAK_HBM_GBS, 4)},
                           "other_gemms": {"mlp_up": gemms["mlp_up"], "qkv": gemms["qkv"]},
                           "mfma_busy_other_gemms": mfma_busy_other,
                           "yardstick_tflops": yardstick,
                           "clock_note": "peak is the nominal 2.4 GHz figure the contract asks for.  The clock these kernels actually run at is on this line: "
                                         "mfma_busy.effective_clock_GHz (GRBM_GUI_ACTIVE over the kernel's duration, counter passes) and power_under_load.sclk_MHz "
                                         "(rocm-smi under the back-to-back forward); frac x 2.4 / that clock is the fraction of the issue bound at the measured clock"}
        if mfma_busy.get("effective_clock_GHz"):
            out["roofline"]["frac_of_issue_bound_at_measured_clock"] = round(panel_tflops / (PEAK_BF16_TFLOPS * mfma_busy["effective_clock_GHz"] / 2.4), 4)
        if world == 1 and not args.profile:
            out["power_under_load"] = power_under_load(model, x)
        by = nat.C.c_double()
        att = {}
        for which, name in ((0, "spatial"), (1, "temporal")):
            nat.check(nat.lib.sf_bench_attention(model._handle, B, T, which, 20, ws.data_ptr(), ws.numel(), stream,
                                                 nat.C.byref(ms), nat.C.byref(by), nat.C.byref(fl)))
            gbs = by.value / ms.value / 1e6
            att[name] = {"ms": round(ms.value, 4), "algorithmic_GB": round(by.value / 1e9, 4), "GBps": round(gbs, 1),
                         "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4), "tflops": round(fl.value / ms.value / 1e9, 1),
                         "timing": "isolated back-to-back launches"}
            k = name + "_attention"
            if insitu and insitu.get(k, {}).get("ms", 0) > 0:      # the same kernel inside real forwards
                g2 = by.value / insitu[k]["ms"] / 1e6
                att[name]["in_situ"] = {"ms": insitu[k]["ms"], "GBps": round(g2, 1), "frac_of_hbm_peak": round(g2 / PEAK_HBM_GBS, 4)}
        out["attention"] = att
        del ws
        if args.profile:
            print(json.dumps(out), flush=True)
            if dist is not None and dist.is_initialized():
                dist.barrier()
                dist.destroy_process_group()
            return

        # ---- accuracy vs the CPU oracle + the fp32-accurate mode's throughput -------------------
        from oracle import streamformer_oracle as O
        torch.manual_seed(0)
        x1 = torch.randn(1, T, 3, cfg.image_size, cfg.image_size)
        want = O.forward(sd, cfg, x1)
        acc = {}
        o = model(x1.to(dev))
        acc["bf16"] = {"last_hidden_state": float((o.last_hidden_state.cpu() - want["last_hidden_state"]).abs().max()),
                       "pooler_output": float((o.pooler_output.cpu() - want["pooler_output"]).abs().max())}
        del model
        m2 = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="fp32")
        m2.load_state_dict(sd)
        m2.to(dev).eval()
        o = m2(x1.to(dev))
        acc["bf16x3"] = {"last_hidden_state": float((o.last_hidden_state.cpu() - want["last_hidden_state"]).abs().max()),
                         "pooler_output": float((o.pooler_output.cpu() - want["pooler_output"]).abs().max())}
        out["accuracy_max_abs_vs_cpu_oracle"] = acc
        if world == 1:
            dt2 = timed_steps(m2, x, max(args.steps // 2, 3), 2, None, 1)
            out["accurate_mode"] = {"value": round(B * T * max(args.steps // 2, 3) / dt2, 1), "unit": "frames/s",
                                    "dtype": "bf16x3 (fp32-accurate)", "max_abs_last_hidden_state": acc["bf16x3"]["last_hidden_state"]}
        del m2

        # ---- CPU baseline: the oracle on the host cores, bounded sample (N=1 only) ------------
        if world == 1 and not args.no_cpu_baseline:
            cores = args.cpu_threads or usable_cores()
            torch.set_num_threads(cores)
            t0 = time.perf_counter()
            O.forward(sd, cfg, x1)                       # warm-up 1 (also sizes the sample)
            t1 = time.perf_counter() - t0
            if t1 < 4.0:
                O.forward(sd, cfg, x1)                   # warm-up 2
            iters = max(1, min(args.cpu_iters, int(20.0 / max(t1, 1e-3))))   # ~10-30 s of CPU work in total
            ts = []
            for _ in range(iters):
                t0 = time.perf_counter()
                O.forward(sd, cfg, x1)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            med = ts[len(ts) // 2]
            cpu_model = ""
            try:
                with open("/proc/cpuinfo") as f:
                    for line in f:
                        if line.startswith("model name"):
                            cpu_model = line.split(":", 1)[1].strip()
                            break
            except Exception:
                pass
            out["cpu_baseline"] = {"value": round(T / med, 2), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": f"{iters} forwards of one [1,16,3,224,224] clip after warm-up "
                                             f"(median {med:.3f}s, best {ts[0]:.3f}s), fp32 eager torch {torch.__version__}",
                                   "cpu": cpu_model, "best": round(T / ts[0], 2)}
            try:      # the port's wall time against the reference's own CPU forward, measured ONCE in the build container
                with open(os.path.join(ROOT, "profiles", "port_over_reference.json")) as f:      # (tools/port_vs_reference.py; the GPU box has no reference)
                    por = json.load(f)
                out["cpu_baseline"]["port_over_reference_wall"] = {**por["port_over_reference_wall"], "threads": por["threads"], "pairs": por["pairs"],
                                                                   "where": "build container, tools/port_vs_reference.py (profiles/port_over_reference.json)"}
            except Exception:
                out["cpu_baseline"]["port_over_reference_wall"] = None
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
            try:        # scaling context: the same forward on ONE host thread, one timed call
                torch.set_num_threads(1)
                t0 = time.perf_counter()
                O.forward(sd, cfg, x1)
                t1 = time.perf_counter() - t0
                out["cpu_baseline"]["one_thread"] = {"value": round(T / t1, 2), "unit": "frames/s", "sample": "1 forward of one clip"}
            finally:
                torch.set_num_threads(cores)
        if world == 1 and not args.profile:
            # host -> device copy inside the step ("report H2D-inclusive number separately"): the batch
            # comes from pinned host memory every step, as normalised fp32 frames and as raw uint8 frames (the patch
            # kernel fuses the image processor's rescale + normalize); same model, same K steps, no overlap tricks
            try:
                mh = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
                mh.load_state_dict(sd)
                mh.to(dev).eval()
                h2d = {}
                for name, host in (("fp32_frames", x.cpu().pin_memory()),
                                   ("uint8_frames", ((x.cpu().clamp(-1, 1) * 127.5 + 127.5).round().to(torch.uint8)).pin_memory())):
                    for _ in range(3):
                        mh(host.to(dev, non_blocking=True))
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        mh(host.to(dev, non_blocking=True))
                    torch.cuda.synchronize()
                    dth = time.perf_counter() - t0
                    h2d[name] = {"value": round(B * T * args.steps / dth, 1), "ms_per_step": round(1e3 * dth / args.steps, 3),
                                 "MB_per_step": round(host.numel() * host.element_size() / 1e6, 1)}
                out["h2d_inclusive"] = h2d
                del mh
            except Exception as e:
                out["h2d_inclusive"] = {"error": repr(e)}
            try:
                out["latency_b1"] = small_batch_latency(dev)
            except Exception as e:
                out["latency_b1"] = {"error": repr(e)}
            # BASELINE configs[4]: 64-frame online clip, one frame per call through the KV-cache (B = 1)
            try:
                out["streaming"] = streaming_bench(dev)
            except Exception as e:
                out["streaming"] = {"error": repr(e)}
        if world == 1 and not args.no_train:
            # BASELINE configs[2]: the multitask pre-training step on the same clip shape (short measurement;
            # `--mode train` times it under the full contract, also at N > 1)
            try:
                torch.cuda.empty_cache()
                out["train_step"] = train_bench(args, dev, dist, world, rank, 6, 2, with_cpu=not args.no_cpu_baseline)
            except Exception as e:      # never lose the headline line to the extra measurement
                out["train_step"] = {"error": repr(e)}
        elif train_leg is not None:
            out["train_step"] = train_leg
        print(json.dumps(out), flush=True)
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
