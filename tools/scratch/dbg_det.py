import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from einops import rearrange
from oracle import conditioning as OC
from tests import common_models as CM
from synfmc_amd import hip_ops as K
W4 = (64, 128, 256, 256)
ou, oe, oa = CM.build_oracle(W4)
pu, pe, pa = CM.build_product(ou, oe, oa, W4, dtype=torch.bfloat16)
clip = CM.synthetic_clip(B=1, Fr=16, H=128, W=128)
pose_emb = rearrange(OC.to_plucker_embedding(clip["c2w"], clip["K"], (128, 128)), "b f c h w -> b c f h w").cuda().bfloat16()
def run(mod, *a, **k):
    rec = []
    hs = [m.register_forward_hook(lambda m, i, o, n=n: rec.append((n, type(m).__name__, (o[0] if isinstance(o, (tuple, list)) else o))))
          for n, m in mod.named_modules() if n]
    with torch.no_grad():
        out = mod(*a, **k)
    for h in hs: h.remove()
    return rec
for name, mod, args, kw in (("encoder", pe, (pose_emb,), {}),):
    with torch.no_grad(): mod(*args, **kw); mod(*args, **kw)
    r1 = run(mod, *args, **kw); r2 = run(mod, *args, **kw)
    shown = 0
    for (n1, t1, o1), (n2, t2, o2) in zip(r1, r2):
        if torch.is_tensor(o1) and not torch.equal(o1, o2):
            d = float((o1.float() - o2.float()).abs().max() / o2.float().abs().max())
            print(name, "first differing module:", n1, t1, tuple(o1.shape), d); shown += 1
            if shown >= 6: break
    print(name, "modules", len(r1), "differing shown", shown)
print({k: v for k, v in K._choice.items()})
