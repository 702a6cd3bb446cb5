"""Spatial attention backward, phase by phase: SF_ATTN_BWD_LAB = 1 (no dK/dV phase), 2 (no dQ phase), 4 (no output stores),
run under `rocprofv3 --kernel-trace --stats`; one process per setting (the switch is read once)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from streamformer_amd import _native
lib = _native.lib
F, L, H = 128, 196, 12
D = H * 64
qkv = torch.randn(F * L, 3 * D, device="cuda").bfloat16()
o = torch.randn(F * L, D, device="cuda").bfloat16()
do = torch.randn(F * L, D, device="cuda").bfloat16()
dqkv = torch.zeros_like(qkv)
for _ in range(20):
    assert lib.sf_op_attention_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), dqkv.data_ptr(), 0, F, L, 1, H, 0, None) == 0
torch.cuda.synchronize()
