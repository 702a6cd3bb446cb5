"""cpu_baseline.kind = "port": how the oracle's wall time compares with the REFERENCE's own CPU forward (within +-5 %).
Runs in the BUILD container only (it imports /root/reference, which does not exist on the GPU box): SigLIP-base, one 16-frame clip, the same
weights in both, alternating pairs, median of the per-pair ratios.  Writes profiles/port_over_reference.json, which bench.py copies into
cpu_baseline.port_over_reference_wall.

    python tools/port_vs_reference.py [pairs=9] [threads=all]
"""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import torch
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 9
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 8)
torch.set_num_threads(threads)
from models import TimesformerMultiTaskingModelSigLIP as RefModel, StreamformerConfig as RefConfig   # the reference
import streamformer_amd.configuration as Cn
from streamformer_amd.init_weights import make_state_dict
from oracle import streamformer_oracle as O

cfg = Cn.siglip_base()
sd = make_state_dict(cfg, 0)
rcfg = RefConfig(**{k: v for k, v in cfg.to_dict().items() if k in RefConfig().to_dict()})
ref = RefModel(rcfg).eval()
missing, unexpected = ref.load_state_dict({k: v for k, v in sd.items()}, strict=False)
assert not [k for k in missing if "mask" not in k], missing[:5]
x = torch.randn(1, 16, 3, 224, 224, generator=torch.Generator().manual_seed(0))
with torch.no_grad():
    a = ref(x).last_hidden_state
    b = O.forward(sd, cfg, x)["last_hidden_state"]
    err = float((a - b).abs().max())
    for _ in range(2):
        ref(x); O.forward(sd, cfg, x)
    tr, to = [], []
    for i in range(pairs):
        order = (0, 1) if i % 2 == 0 else (1, 0)
        for which in order:
            t0 = time.perf_counter()
            if which == 0: ref(x)
            else: O.forward(sd, cfg, x)
            (tr if which == 0 else to).append(time.perf_counter() - t0)
ratios = [o / r for o, r in zip(to, tr)]
out = {"config": "SigLIP-base, [1,16,3,224,224], fp32, same weights", "threads": threads, "pairs": pairs, "host": os.uname().nodename,
       "torch": torch.__version__, "max_abs_oracle_vs_reference": err,
       "reference_s": {"median": round(statistics.median(tr), 4), "min": round(min(tr), 4)},
       "oracle_s": {"median": round(statistics.median(to), 4), "min": round(min(to), 4)},
       "port_over_reference_wall": {"median_of_pair_ratios": round(statistics.median(ratios), 3), "best_of": round(min(to) / min(tr), 3),
                                    "pair_ratios": [round(r, 3) for r in ratios]},
       "note": "measured in the build container (8 shared vCPUs, noisy); alternating pairs; the GPU box cannot run the reference"}
json.dump(out, open(os.path.join(ROOT, "profiles", "port_over_reference.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
