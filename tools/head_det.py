"""Which buffer of the training forward's pooling head stops repeating under contention ("Device sharing").

N passes of forward (+ localization loss + backward unless SF_DET_FWD_ONLY=1) on the same inputs; after every forward the head's saved
buffers are copied out of the workspace (offsets restated from tcarve, sf_train.hip) and compared bit for bit with the first pass:
the first buffer in data-flow order that differs names the kernel.  Modes (environment):
  SF_DET_N          passes (default 200)
  SF_DET_B          clips (default 8)
  SF_DET_FWD_ONLY   1 = no loss / backward in this process
  SF_DET_NOISE      k > 0 = before every pass, k large matmuls are enqueued on a second stream of THIS process (what a concurrent
                    RCCL kernel would be to the pass: other kernels on the same CUs)
  SF_DET_TAG        label printed with every line
  SF_DET_SECONDS    stop after this many seconds;  SF_DET_TS=1: a heartbeat line with the wall clock every 200 passes
Run two of them concurrently on one device for the cross-process case (tools/head_det.sh); SF_POOL_SHARE_CU=1 restores the state the finding
was made in (the pooling-head kernels with their exact LDS sizes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
from streamformer_amd.training import StreamformerTrainer

B = int(os.environ.get("SF_DET_B", "8"))
NP = int(os.environ.get("SF_DET_N", "200"))
FWD_ONLY = os.environ.get("SF_DET_FWD_ONLY", "0") == "1"
NOISE = int(os.environ.get("SF_DET_NOISE", "0"))
TAG = os.environ.get("SF_DET_TAG", f"pid{os.getpid()}")
T = 16
cfg = sa.siglip_base(add_lora_spatial=True)
tr = StreamformerTrainer(cfg, sa.make_state_dict(cfg, seed=0, lora=True), ["retrieval", "localization"], freeze_spatial=True, device="cuda:0")
g = torch.Generator().manual_seed(5)
x = torch.randn(B, T, 3, 224, 224, generator=g).cuda()
lab = torch.randn(20, cfg.hidden_size, generator=g); lab = (lab / lab.norm(dim=-1, keepdim=True)).cuda()
ti = {"kind": "localization", "label_emb": lab, "labels": torch.randint(-1, 20, (B, T), generator=g).cuda()}


def head_regions():
    """(name, byte offset, bytes, rows) of the head's saved buffers: tcarve (sf_train.hip) restated for the forward-saved part."""
    D, I, L, H = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads
    N = (cfg.image_size // cfg.patch_size) ** 2
    Kp = cfg.num_channels * cfg.patch_size ** 2
    M, F = B * T * N, B * T
    off = 0

    def take(nbytes):
        nonlocal off
        off = (off + 255) & ~255
        o = off
        off += nbytes
        return o
    take(M * Kp * 2); take(T * D * 4)
    for _ in range(L + 1): take(M * D * 4)
    for _ in range(L):
        take(M * D * 4); take(M * D * 4)
        take(M * D * 2); take(M * 3 * D * 2); take(M * D * 2); take(M * D * 2)
        take(M * D * 2); take(M * 3 * D * 2); take(M * D * 2)
        take(M * D * 2); take(M * I * 2); take(M * I * 2)
        take(F * H * N * 4)
    S = 1
    while S < 8 and F * S < 256 and (N + 2 * S - 1) // (2 * S) >= 16: S *= 2          # sf_pool_splits
    out = []
    for name, nbytes in (("xn", M * D * 2), ("pc", F * D * 2), ("pz", F * H * D * 4), ("pprobs", F * H * N * 4), ("pml", F * S * H * 2 * 4),
                         ("pzpart", F * S * H * D * 4), ("attn_out", F * D * 4), ("hn", F * D * 2), ("hm_pre", F * I * 2), ("hm", F * I * 2)):
        out.append((name, take(nbytes), nbytes))
    return out, S


REG, S = head_regions()
# data-flow order of the head: probe -> (pprobs, pml, pzpart) -> ctx -> (pz, pc) -> head_out GEMM -> attn_out -> LayerNorm -> hn -> fc1 -> hm_pre -> gelu -> hm -> fc2 -> pooler
ORDER = ["xn", "pprobs", "pml", "pzpart", "pz", "pc", "attn_out", "hn", "hm_pre", "hm"]
F = B * T
noise_stream = torch.cuda.Stream() if NOISE else None
na = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16) if NOISE else None
ref = None
bad = 0
first_buf = {}
T_END = time.time() + float(os.environ.get("SF_DET_SECONDS", "1e9"))          # optional wall-clock limit
done = 0
for it in range(NP):
    if time.time() > T_END: break
    done = it
    if NOISE:
        with torch.cuda.stream(noise_stream):
            for _ in range(NOISE): nb = na @ na
    if not FWD_ONLY: tr.zero_grad()
    lhs, pooler = tr.forward(x)
    torch.cuda.synchronize()
    ws = tr._ws
    cur = {n: ws[o: o + nb_].clone() for n, o, nb_ in REG}
    cur["pooler"] = pooler.clone(); cur["lhs"] = lhs.clone()
    if not FWD_ONLY:
        _, gp, _ = tr.loss_and_grad("localization", pooler, ti)
        tr.backward(gp)
        torch.cuda.synchronize()
        cur["grads"] = tr.grads.clone()
    if ref is None:
        ref = cur
        if it == 0:
            sane = {n: float(cur[n].view(torch.bfloat16 if n in ("xn", "pc", "hn", "hm_pre", "hm") else torch.float32).float().abs().mean()) for n in ORDER}
            print(f"[{TAG}] S={S} regions (mean |.|): " + ", ".join(f"{k}={v:.3g}" for k, v in sane.items()), flush=True)
        continue
    diff = [n for n in ORDER + ["pooler", "lhs"] + ([] if FWD_ONLY else ["grads"]) if not torch.equal(cur[n], ref[n])]
    if diff:
        bad += 1
        first = diff[0]
        first_buf[first] = first_buf.get(first, 0) + 1
        detail = ""
        if first in ORDER:
            a, b = cur[first], ref[first]
            nz = (a != b).nonzero().flatten()
            rows = sorted(set((nz // (a.numel() // F)).tolist()))               # every head buffer is frame-major
            per = a.numel() // F
            within = sorted(set((nz % per).tolist()))
            detail = f"; {first}: {nz.numel()} bytes in frames {rows[:24]} ({len(rows)} frames), byte-in-frame range {within[0]}..{within[-1]} of {per}"
        if "pooler" in diff:
            d = (cur["pooler"] - ref["pooler"]).abs().amax(-1).flatten()
            detail += f"; pooler max {float(d.max()):.3e} in frames {d.nonzero().flatten().tolist()[:24]}"
        print(f"[{TAG}] pass {it} t={time.time():.2f}: differing {diff}{detail}", flush=True)
    if os.environ.get("SF_DET_TS") == "1" and it % 200 == 0: print(f"[{TAG}] heartbeat pass {it} t={time.time():.2f} bad so far {bad}", flush=True)
print(f"[{TAG}] {done} repeats, {bad} differing; first differing buffer counts {first_buf}", flush=True)
