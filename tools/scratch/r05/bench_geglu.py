import sys, os, torch
sys.path.insert(0, ".")
from synfmc_amd import hip_ops as K
def t_ms(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
M, C, cff = 81920, 320, 1280
x = torch.randn(M, C, device="cuda").bfloat16(); w = (torch.randn(2 * cff, C, device="cuda") * C ** -0.5).bfloat16(); b = torch.randn(2 * cff, device="cuda").bfloat16()
g, beta = torch.randn(C, device="cuda") * 0.2 + 1, torch.randn(C, device="cuda")
wp = K.pack_geglu_frag80(w)
with torch.no_grad():
    ms = t_ms(lambda: K.geglu_ln_direct(x, g, beta, 1e-5, wp, b, cff))
print(f"FMC_GEGLU320_ROWS160={os.environ.get('FMC_GEGLU320_ROWS160','1')}: geglu320 {M}x{2*cff}x{C}: {ms*1e3:.1f} us = {2.0*M*2*cff*C/ms/1e9:.0f} TF/s")

M, C, cff = 20480, 640, 2560
x = torch.randn(M, C, device="cuda").bfloat16(); w = (torch.randn(2 * cff, C, device="cuda") * C ** -0.5).bfloat16(); b = torch.randn(2 * cff, device="cuda").bfloat16()
g, beta = torch.randn(C, device="cuda") * 0.2 + 1, torch.randn(C, device="cuda")
wp = K.pack_geglu_frag80(w)
with torch.no_grad():
    ms = t_ms(lambda: K.geglu_ln_direct(x, g, beta, 1e-5, wp, b, cff))
print(f"geglu640 {M}x{2*cff}x{C}: {ms*1e3:.1f} us = {2.0*M*2*cff*C/ms/1e9:.0f} TF/s")
