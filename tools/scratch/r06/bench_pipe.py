import sys, os, torch
sys.path.insert(0, ".")
import torch.nn.functional as F
from synfmc_amd import hip_ops as K
def t_ms(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for M, C, cff in ((81920, 320, 1280), (20480, 640, 2560)):
    x = (torch.randn(M, C, device="cuda") * 1.5 + 0.2).bfloat16(); w = (torch.randn(2 * cff, C, device="cuda") * C ** -0.5).bfloat16(); b = (torch.randn(2 * cff, device="cuda") * 0.3).bfloat16()
    g, beta = torch.randn(C, device="cuda") * 0.2 + 1, torch.randn(C, device="cuda") * 0.2
    wp80 = K.pack_geglu_frag80(w)
    VAR = int(os.environ.get('VAR', '0')) if C == 320 else 0
    wp64 = K.pack_geglu_frag(w, 32 if VAR == 0 else 16)
    with torch.no_grad():
        ref = K.geglu_ln_direct(x, g, beta, 1e-5, wp80, b, cff)
        got = K.geglu_ln_pipe(x, g, beta, 1e-5, wp64, b, cff, variant=VAR)
        torch.cuda.synchronize()
        d = (got.float() - ref.float()).abs().max().item()
        n = F.layer_norm(x.float(), (C,), g, beta, 1e-5).bfloat16().float()
        y = F.linear(n, w.float(), b.float())
        exact = y[:, :cff] * F.gelu(y[:, cff:])
        e_ref = ((ref.float() - exact).abs().max() / exact.abs().max()).item(); e_got = ((got.float() - exact).abs().max() / exact.abs().max()).item()
        print(f"C={C}: pipe vs direct max|diff| {d:.3e} (bit-equal: {torch.equal(got, ref)}); rel-inf vs fp32: direct {e_ref:.3e} pipe {e_got:.3e}")
        if M % 160 == 0:
            gb = K.geglu_ln_pipe(x, g, beta, 1e-5, wp64, b, cff, blocked=True, variant=VAR)
            print("   blocked layout equal:", torch.equal(gb.view(M // 160, cff // 32, 160, 32).permute(0, 2, 1, 3).reshape(M, cff), got))
        ms_d = t_ms(lambda: K.geglu_ln_direct(x, g, beta, 1e-5, wp80, b, cff))
        ms_p = t_ms(lambda: K.geglu_ln_pipe(x, g, beta, 1e-5, wp64, b, cff, variant=VAR))
        ms_pb = t_ms(lambda: K.geglu_ln_pipe(x, g, beta, 1e-5, wp64, b, cff, blocked=True, variant=VAR)) if M % 160 == 0 else float("nan")
    fl = 2.0 * M * 2 * cff * C
    print(f"   direct {ms_d * 1e3:.1f} us = {fl / ms_d / 1e9:.0f} TF/s | pipe {ms_p * 1e3:.1f} us = {fl / ms_p / 1e9:.0f} TF/s | pipe blocked {ms_pb * 1e3:.1f} us")
