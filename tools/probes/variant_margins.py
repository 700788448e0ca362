import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from oracle import cases
from fastfourierdiffusion_amd.utils import synthetic
from fastfourierdiffusion_amd import _native as N
import test_gpu_parity as T
from conftest import rel_err
lib = N.lib()
c = next(c for c in cases.MODEL_CASES if c["name"] == "ecg")
m, _ = T.make_model(None, c)
for B in (1, 5, 64, 200, 512):
    x = torch.from_numpy(next(synthetic.noise_stream((B, c["L"], c["C"]), 1, 777))).cuda()
    outs = {}
    for name, tunes in (("large", {"small_path": 0, "mid_path": 0, "attn_small": 0}), ("auto", {}), ("split", {"ffn_split": 1}), ("mid4", {"small_path": 0, "mid_path": 4})):
        for k in ("small_path", "mid_path", "attn_small"): lib.ffd_tune(k.encode(), 1)
        lib.ffd_tune(b"ffn_split", 0)
        for k, v in tunes.items(): lib.ffd_tune(k.encode(), v)
        outs[name] = m(T.batch_of(x, 0.41)).cpu()
    print(B, {k: f"{rel_err(v, outs['large']):.2e}" for k, v in outs.items() if k != "large"})
for k in ("small_path", "mid_path", "attn_small"): lib.ffd_tune(k.encode(), 1)
lib.ffd_tune(b"ffn_split", 0)
