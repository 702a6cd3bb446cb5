"""The C ABI driven from C++ with no Python in the process (examples/host_forward.cpp) gives the same bits as the
Python mirror on the same weights and clip: the drop-in boundary is the shared library, not the wrapper."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from streamformer_amd.init_weights import make_state_dict
from tests.conftest import ROOT
from tests.helpers import frames, small_cfg

BIN = os.path.join(ROOT, "examples", "host_forward")


def test_host_example_is_built():
    """CPU: build() produced the example binary and it links against the in-tree library."""
    assert os.path.exists(BIN), "run python streamformer_amd/build.py"
    out = subprocess.run([BIN], capture_output=True, text=True)
    assert out.returncode == 2 and "usage" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_cpp_host_matches_python_mirror(tmp_path, mode):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import streamformer_amd as sa
    from streamformer_amd.convert import dump_for_c_host
    cfg = small_cfg(add_lora_spatial=True)
    sd = make_state_dict(cfg, seed=14, lora=True)
    x = frames(14, (2, 6, 3, 48, 48))
    wfile, cfile, ofile = (str(tmp_path / n) for n in ("w.bin", "clip.bin", "out.bin"))
    dump_for_c_host(cfg, sd, wfile)
    with open(cfile, "wb") as f:
        f.write(struct.pack("<4i", 2, 6, 48, 48))
        f.write(x.numpy().tobytes())
    r = subprocess.run([BIN, wfile, cfile, ofile, mode], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(ofile, dtype=np.uint8)
    B, T, N, D = struct.unpack("<4i", raw[:16].tobytes())
    vals = raw[16:].view(np.float32)
    lhs = torch.from_numpy(vals[: B * T * N * D].reshape(B, T, N, D).copy())
    pool = torch.from_numpy(vals[B * T * N * D:].reshape(B, T, D).copy())
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda")
    out = m(x.cuda())
    assert torch.equal(out.last_hidden_state.cpu(), lhs) and torch.equal(out.pooler_output.cpu(), pool)
