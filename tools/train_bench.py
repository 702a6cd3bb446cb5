"""Time the training micro-step (BASELINE config #3 shape: SigLIP-base + LoRA recipe, 8 clips x 16 frames).
usage: python tools/train_bench.py [B] [steps] [--profile]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamformer_amd.configuration import siglip_base  # noqa: E402
from streamformer_amd.init_weights import make_state_dict  # noqa: E402
from streamformer_amd.training import StreamformerTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 5
cfg = siglip_base(add_lora_spatial=True)
dev = torch.device("cuda:0")
sd = make_state_dict(cfg, seed=0, lora=True)
tr = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], freeze_spatial=True, device=dev, lr=2e-5 * B / 256, weight_decay=0.05)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 16, 3, 224, 224, generator=g).to(dev)
text = torch.randn(B, 768, generator=g).to(dev)
lab = torch.randn(20, 768, generator=g)
lab = (lab / lab.norm(dim=-1, keepdim=True)).to(dev)
labels = torch.randint(-1, 20, (B, 16), generator=g).to(dev)
tasks = [("retrieval", {"kind": "retrieval", "text": text}), ("localization", {"kind": "localization", "label_emb": lab, "labels": labels})]


def step(i):
    t, ti = tasks[i % 2]
    return tr.micro_step(t, x, ti)


for i in range(2):
    step(i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
t0 = time.perf_counter()
for i in range(steps):
    loss = step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# phase split of one step
ev[0].record(); _, pooler = tr.forward(x); ev[1].record()
l, gp, gs = tr.loss_and_grad("retrieval", pooler, tasks[0][1]); tr.backward(gp); ev[2].record()
tr.optimizer_step(); ev[3].record()
torch.cuda.synchronize()
print(f"B={B}: {dt*1e3:.2f} ms/step = {B*16/dt:.0f} frames/s  (loss {float(loss):.4f}); "
      f"forward {ev[0].elapsed_time(ev[1]):.2f} ms, loss+backward {ev[1].elapsed_time(ev[2]):.2f} ms, "
      f"adamw+weight sync {ev[2].elapsed_time(ev[3]):.2f} ms; workspace {tr._ws.numel()/2**30:.2f} GiB")
