import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd.models.vae import AutoencoderKL
torch.manual_seed(0)
vae = AutoencoderKL(block_out_channels=(64, 64, 64, 64)).cuda().eval().requires_grad_(False)
def hook(name):
    def f(m, i, o):
        torch.cuda.synchronize()
        print("ok", name, tuple(o.shape) if torch.is_tensor(o) else type(o), flush=True)
    return f
for n, m in vae.named_modules():
    if n and len(list(m.children())) == 0 or n.endswith(("resnets.0", "resnets.1", "resnets.2", "attentions.0", "upsamplers.0")):
        m.register_forward_hook(hook(n))
z = torch.randn(1, 4, 16, 16, device="cuda")
with torch.no_grad():
    y = vae.decode(z).sample
torch.cuda.synchronize()
print("done", y.shape, float(y.abs().max()))
