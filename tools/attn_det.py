import sys, torch
sys.path.insert(0, "/root/repo")
import streamformer_amd as sa
from streamformer_amd import _native as nat
dev = torch.device("cuda:0")
def run(groups, L, heads=12):
    g = torch.Generator().manual_seed(groups * 1000 + L)
    qkv = torch.randn(groups, L, 3 * heads * 64, generator=g).to(dev)
    nb = nat.lib.sf_op_attention_workspace_bytes(groups, L, heads, 64)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    outs = []
    for rep in range(6):
        ctx = torch.empty(groups, L, heads * 64, device=dev)
        nat.check(nat.lib.sf_op_attention(qkv.data_ptr(), ctx.data_ptr(), groups, L, heads, 64, 0, 0, 0, 0, ws.data_ptr(), nb, nat.current_stream_handle(dev)))
        torch.cuda.synchronize()
        outs.append(ctx.clone())
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    # reference
    q, k, v = [qkv[..., i * 768:(i + 1) * 768].reshape(groups, L, heads, 64).transpose(1, 2).double() for i in range(3)]
    qb, kb, vb = [t.bfloat16().double() for t in (q, k, v)]
    ref = (torch.softmax(qb @ kb.transpose(-1, -2) / 8.0, -1) @ vb).transpose(1, 2).reshape(groups, L, heads * 64)
    err = float((outs[0].double() - ref).abs().max())
    if not same:
        d = (outs[0] - outs[1]).abs()
        idx = d.nonzero()
        print("   differing elements:", idx.shape[0], "first:", idx[:3].tolist(), "max diff", float(d.max()))
    print(f"groups={groups} L={L}: reproducible={same} max err vs fp64 ref {err:.3e}")
for groups, L in ((1, 196), (2, 196), (8, 196), (128, 196), (1, 64), (3, 100), (1, 224), (5, 33)):
    run(groups, L)
