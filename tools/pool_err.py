"""Round 4: where the bf16 mode's pooler_output error comes from.  One SigLIP-base clip:
bf16 model end to end; the fp32-accurate HEAD on the bf16 model's last_hidden_state; the fp32-accurate post-LayerNorm + head on the
bf16 model's pre-LayerNorm hidden state; all against the CPU oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
from oracle import streamformer_oracle as O
cfg = sa.siglip_base()
sd = sa.make_state_dict(cfg, 0)
mb = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16"); mb.load_state_dict(sd); mb.to("cuda").eval()
ma = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="fp32"); ma.load_state_dict(sd); ma.to("cuda").eval()
x = torch.randn(1, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1))
want = O.forward(sd, cfg, x)
lhs = want["last_hidden_state"] if isinstance(want, dict) else want[0]
pool = want["pooler_output"] if isinstance(want, dict) else want[1]
with torch.no_grad():
    ob = mb(x.cuda(), output_hidden_states=True)
    oa = ma(x.cuda())
    T, N, D = 16, 196, cfg.hidden_size
    l = ob.last_hidden_state.reshape(T, N, D)
    p1 = ma.head(l)                                           # accurate head on bf16-mode tokens
    pre = ob.hidden_states[-1]                                # (B, N*T, D) patch-major, pre post-LayerNorm
    pre_fm = pre.reshape(1, N, T, D).permute(0, 2, 1, 3).reshape(T, N, D).contiguous()
    p2 = ma.head(ma.post_layernorm(pre_fm))
    p3 = mb.head(oa.last_hidden_state.reshape(T, N, D))       # bf16 head on accurate tokens
e = lambda a, b: float((a.float().cpu().reshape(b.shape) - b).abs().max())
print(f"bf16 end to end        : lhs {e(ob.last_hidden_state, lhs):.3e}  pooler {e(ob.pooler_output, pool):.3e}")
print(f"accurate end to end    : lhs {e(oa.last_hidden_state, lhs):.3e}  pooler {e(oa.pooler_output, pool):.3e}")
print(f"accurate head on bf16 lhs (hidden_states path)       : pooler {e(p1, pool):.3e}")
print(f"accurate post-LN + head on bf16 pre-LN hidden state  : pooler {e(p2, pool):.3e}")
print(f"bf16 head on accurate lhs                            : pooler {e(p3, pool):.3e}")
print(f"pooler scale: max |pooler| {float(pool.abs().max()):.3f}, rms {float(pool.pow(2).mean().sqrt()):.3f};  lhs max {float(lhs.abs().max()):.3f} rms {float(lhs.pow(2).mean().sqrt()):.3f}")
