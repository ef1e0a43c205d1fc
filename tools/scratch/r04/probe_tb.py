#!/usr/bin/env python
"""Fused temporal attention block (fmc_temporal_block_bf16) at the bench size: time per launch, MFMA fraction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from synfmc_amd import hip_ops as K
torch.manual_seed(0)
B, Fr, hw, C = 2, 16, 2560, 320
dev = "cuda"
h = torch.randn(B, Fr, hw, C, device=dev, dtype=torch.bfloat16)
pt = torch.randn(B, Fr, hw, C, device=dev, dtype=torch.bfloat16)
g = torch.randn(C, device=dev) * 0.2 + 1
bpe = torch.randn(Fr, C, device=dev)
wq = torch.randn(3 * C, C, device=dev, dtype=torch.bfloat16) * C ** -0.5
wo = torch.randn(C, C, device=dev, dtype=torch.bfloat16) * C ** -0.5
wm = torch.randn(C, C, device=dev, dtype=torch.bfloat16) * C ** -0.5
bo = torch.randn(C, device=dev, dtype=torch.bfloat16)
wqp, wot, wmt = K.pack_temporal_qkv(wq), K._w_tilemajor(wo), K._w_tilemajor(wm)
M = B * Fr * hw
for merge in (True, False):
    kw = dict(w_merge_tm=wmt, pose_term=pt, merge_scale=0.7) if merge else {}
    fn = lambda: K.temporal_block(h, g, bpe, 1e-5, wqp, wot, bo, 40 ** -0.5, **kw)
    fn(); torch.cuda.synchronize()
    ms = K._time_ms(fn)
    fl = 2.0 * M * C * ((C if merge else 0) + 4 * C) + 4.0 * M * Fr * C
    print(f"fused block merge={merge}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.0f} TF/s  frac {fl / ms / 1e9 / 2500:.3f}", flush=True)

from synfmc_amd import _lib
lib = _lib.load()
for merge in (True, False):
    kw = dict(w_merge_tm=wmt, pose_term=pt, merge_scale=0.7) if merge else {}
    buf = torch.zeros(256, 4, 8, dtype=torch.int64, device=dev)
    lib.fmc_temporal_block_set_debug(buf.data_ptr())
    K.temporal_block(h, g, bpe, 1e-5, wqp, wot, bo, 40 ** -0.5, **kw)
    torch.cuda.synchronize()
    lib.fmc_temporal_block_set_debug(None)
    t = buf.cpu().double() * 0.01                       # us
    t0 = t[:, 0, 0].min()
    names = ["start", "h landed", "LN done", "merge done", "D compute done", "D barrier", "E loop done", "tile done"]
    for slot in (0, 1):
        d = t[:, slot, :] - t0
        print(f"merge={merge} tile slot {slot}: " + " | ".join(f"{n} {d[:, i].mean():6.1f} (max {d[:, i].max():6.1f})" for i, n in enumerate(names)))
