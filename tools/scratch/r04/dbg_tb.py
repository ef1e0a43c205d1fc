import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch, torch.nn.functional as F
from synfmc_amd import hip_ops as K
torch.manual_seed(0)
B, Fr, hw, C, H = 1, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 10, 320, 8
h = (torch.randn(B, Fr, hw, C) * 1.5 + 0.2).bfloat16()
g = torch.randn(C) * 0.3 + 1; b = torch.randn(C) * 0.2; pe = torch.randn(Fr, C) * 0.7
wq = (torch.randn(3 * C, C) * C ** -0.5 * 1.5).bfloat16(); wo = (torch.randn(C, C) * C ** -0.5).bfloat16(); bo = (torch.randn(C) * 0.3).bfloat16()
wm = (torch.randn(C, C) * C ** -0.5).bfloat16(); pt = torch.randn(B, Fr, hw, C).bfloat16()
for merge in (False, True):
    x = F.layer_norm(h.float(), (C,), g, b, 1e-5) + pe[None, :, None, :]
    m = 0.7 * F.linear(x, wm.float()) + pt.float() + x if merge else x
    qkv = F.linear(m, wq.float())
    q, k, v = (t.reshape(B, Fr, hw, H, 40).permute(0, 2, 3, 1, 4) for t in qkv.chunk(3, dim=-1))
    p = torch.softmax(q @ k.transpose(-1, -2) * 40 ** -0.5, dim=-1)
    o = (p @ v).permute(0, 3, 1, 2, 4).reshape(B, Fr, hw, C)
    ref = F.linear(o, wo.float(), bo.float()) + h.float()
    kw = dict(w_merge_tm=K._w_tilemajor(wm.cuda()), pose_term=pt.cuda(), merge_scale=0.7) if merge else {}
    out = K.temporal_block(h.cuda(), g.cuda(), (b[None] + pe).cuda().contiguous(), 1e-5, K.pack_temporal_qkv(wq.cuda()), K._w_tilemajor(wo.cuda()), bo.cuda(), 40 ** -0.5, **kw)
    stop = int(os.environ.get("FMC_TB_STOP", "0"))
    if stop == 1: ref = x
    if stop in (2, 4): ref = m
    if stop == 3: ref = o
    if stop == 5: ref = qkv[..., :C]
    if stop == 6:
        # p: [B, hw, H, Fq, Fk] -> out layout [B, Fq, hw, 40 h + k]
        pr = p.permute(0, 3, 1, 2, 4)                        # [B, Fq, hw, H, Fk]
        got = out.float().cpu().reshape(B, Fr, hw, H, 40)[..., :16]
        err = (got - pr).abs()
        print("merge", merge, "P max err", err.max().item()); print(" per pixel:", [round(err[0, :, pp].max().item(), 3) for pp in range(hw)]); continue
    err = (out.float().cpu() - ref).abs()
    print("merge", merge, "max err", err.max().item(), "ref max", ref.abs().max().item())
    e = err[0]                                            # [F, hw, C]
    print(" per frame:", [round(e[f].max().item(), 2) for f in range(Fr)])
    print(" per pixel:", [round(e[:, p].max().item(), 2) for p in range(min(hw, 20))])
    print(" per 40-col block:", [round(e[:, :, c * 40:(c + 1) * 40].max().item(), 2) for c in range(8)])
