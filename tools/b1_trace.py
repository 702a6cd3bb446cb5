"""Kernel table of the forward at small batches:  rocprofv3 --kernel-trace -d D -o x -- python tools/b1_trace.py B ; then
python profiles/summarize.py <db>.  (SF_MODE=bf16|fp32)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=os.environ.get("SF_MODE", "bf16"))
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda").eval()
x = torch.randn(B, 16, 3, 224, 224).cuda()
for _ in range(12):
    m(x)
torch.cuda.synchronize()
