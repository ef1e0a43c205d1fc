import sys, os, torch
sys.path.insert(0, ".")
from synfmc_amd import hip_ops as K
M, C, cff = 81920, 320, 1280
x = torch.randn(M, C, device="cuda").bfloat16(); w = (torch.randn(2 * cff, C, device="cuda") * C ** -0.5).bfloat16(); b = torch.randn(2 * cff, device="cuda").bfloat16()
g, beta = torch.randn(C, device="cuda") * 0.2 + 1, torch.randn(C, device="cuda")
wp = K.pack_geglu_frag80(w)
with torch.no_grad():
    for _ in range(3):
        out = K.geglu_ln_direct(x, g, beta, 1e-5, wp, b, cff)
torch.cuda.synchronize()
t = out.view(-1)[: 2 * 64 * 4].view(torch.int64).cpu().view(2, 8, 8)[:, :4, :7]
names = ["loop", "settle+bias issue", "bias wait..gelu..S writes", "barrier 1", "stores", "barrier 2"]
for wg in range(2):
    print("workgroup", (0, 300)[wg], "(cycles of the 100 MHz s_memtime counter x ?)")
    for ch in range(4):
        r = t[wg, ch]
        d = [int(r[i + 1] - r[i]) for i in range(6)]
        print("  chunk", ch, "start", int(r[0] - t[wg, 0, 0]), " ".join(f"{n}={v}" for n, v in zip(names, d)), "next chunk starts +", int(t[wg, ch + 1, 0] - r[6]) if ch < 3 else "")
