import sys, torch
sys.path.insert(0,'/root/repo')
import streamformer_amd.configuration as Cn
from streamformer_amd.init_weights import make_state_dict
from oracle import streamformer_oracle as O
cfg = Cn.siglip_base(); sd = make_state_dict(cfg, 0)
x = torch.randn(1,16,3,224,224, generator=torch.Generator().manual_seed(1))
torch.set_num_threads(8)
ref = O.forward(sd,cfg,x)
lhs0 = ref["last_hidden_state"] if isinstance(ref,dict) else ref[0]
pl0 = ref["pooler_output"] if isinstance(ref,dict) else ref[1]
orig = O._mha
def mk(rq, rp):
    def f(q,k,v,heads,mask,prob_mask=None):
        G,Lq,D=q.shape; Lk=k.shape[1]; d=D//heads
        q,k,v = rq(q),rq(k),rq(v)
        qh=q.reshape(G,Lq,heads,d).transpose(1,2); kh=k.reshape(G,Lk,heads,d).transpose(1,2); vh=v.reshape(G,Lk,heads,d).transpose(1,2)
        s=(qh@kh.transpose(-2,-1))*(d**-0.5)
        if mask is not None: s=s.masked_fill(~mask,float('-inf'))
        p=s.softmax(-1)
        return (rp(p)@vh).transpose(1,2).reshape(G,Lq,D), p
    return f
bf=lambda t:t.bfloat16().float(); hf=lambda t:t.half().float(); idt=lambda t:t
def bf2(t):
    h=t.bfloat16().float(); return h+(t-h).bfloat16().float()
for name,rq,rp in [("bf16 qkv, bf16 p",bf,bf),("bf16 qkv, fp32 p",bf,idt),("fp16 qkv, fp16 p",hf,hf),("fp16 qkv, fp32 p", hf, idt),("bf16x2 qkv, bf16x2 p",bf2,bf2)]:
    O._mha = mk(rq,rp)
    out = O.forward(sd,cfg,x)
    l = out["last_hidden_state"] if isinstance(out,dict) else out[0]; p = out["pooler_output"] if isinstance(out,dict) else out[1]
    print(f"{name:24s}: lhs max-abs {float((l-lhs0).abs().max()):.3e}  pooler {float((p-pl0).abs().max()):.3e}", flush=True)
O._mha = orig
print("lhs absmax", float(lhs0.abs().max()))
def mk2(rqk, rv, rp):
    def f(q,k,v,heads,mask,prob_mask=None):
        G,Lq,D=q.shape; Lk=k.shape[1]; d=D//heads
        q,k,v = rqk(q),rqk(k),rv(v)
        qh=q.reshape(G,Lq,heads,d).transpose(1,2); kh=k.reshape(G,Lk,heads,d).transpose(1,2); vh=v.reshape(G,Lk,heads,d).transpose(1,2)
        s=(qh@kh.transpose(-2,-1))*(d**-0.5)
        if mask is not None: s=s.masked_fill(~mask,float('-inf'))
        p=s.softmax(-1)
        return (rp(p)@vh).transpose(1,2).reshape(G,Lq,D), p
    return f
def hf2(t):
    h=t.half().float(); return h+(t-h).half().float()
for name,a,b,c in [("fp16 qk only",hf,idt,idt),("fp16 v,p only",idt,hf,hf),("fp16 v only",idt,hf,idt),("bf16 v,p only",idt,bf,bf),("bf16 qk only",bf,idt,idt)]:
    O._mha = mk2(a,b,c)
    out = O.forward(sd,cfg,x)
    l = out["last_hidden_state"] if isinstance(out,dict) else out[0]; p = out["pooler_output"] if isinstance(out,dict) else out[1]
    print(f"{name:24s}: lhs max-abs {float((l-lhs0).abs().max()):.3e}  pooler {float((p-pl0).abs().max()):.3e}", flush=True)
