import os, sys
sys.path.insert(0, os.environ["R"])
import torch, streamformer_amd as sa
cfg = sa.siglip_base(num_frames=64)
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, seed=0)); m.to("cuda").eval()
S = int(os.environ.get("S", "8"))
x = torch.randn(S, 64, 3, 224, 224).cuda()
cache = m.new_cache(S, 64)
with torch.no_grad():
    for rep in range(2):
        cache.reset()
        for t in range(64):
            m(x[:, t:t + 1], use_cache=True, past_key_values=cache)
torch.cuda.synchronize()
