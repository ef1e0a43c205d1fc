"""Isolated timing of fmc_conv3x3_halo4_bf16 against the shipped conv3x3 arms on the small-level convolution shapes (and the 20x32 level for comparison)."""
import sys, torch
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

shapes = [(32, 40, 64, 320, 320, False), (32, 40, 64, 960, 320, False), (32, 40, 64, 640, 640, True), (32, 20, 32, 1920, 640, False),
          (32, 10, 16, 1280, 1280, False), (32, 10, 16, 640, 1280, False), (32, 10, 16, 2560, 1280, False), (32, 10, 16, 1920, 1280, False),
          (32, 10, 16, 1280, 1280, True), (32, 5, 8, 1280, 1280, False), (32, 5, 8, 2560, 1280, False),
          (32, 20, 32, 640, 640, False), (32, 20, 32, 1280, 640, False)]
for n, h, w, cin, cout, ups in shapes:
    hs, ws = (h // 2, w // 2) if ups else (h, w)
    x = torch.randn(n, hs, ws, cin, device="cuda").bfloat16()
    wt = (torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(cout, device="cuda").bfloat16()
    fl = 2.0 * n * h * w * cout * 9 * cin
    xn = x.permute(0, 3, 1, 2)
    ref = lambda: K.conv3x3(xn, wt, bias, upsample=ups)
    a = t_ms(ref)
    b = t_ms(lambda: K.conv3x3_halo4(x, wt, bias, upsample=ups))
    c = t_ms(lambda: K.conv3x3_halo4(x, wt, bias, upsample=ups, emit_gn=True))
    y0 = ref().permute(0, 2, 3, 1).float(); y1 = K.conv3x3_halo4(x, wt, bias, upsample=ups).float()
    err = ((y0 - y1).abs().max() / y0.abs().max()).item()
    sk = K.conv3x3_halo4_split(n, h, w, cin, cout)
    wd = float("nan")
    if K.conv3x3_halo4_supported(n, h, w, cin, cin, cout, ups, wide=True):
        wd = t_ms(lambda: K.conv3x3_halo4(x, wt, bias, upsample=ups, wide=True))
    print(f"{n}x{h}x{w} {cin}->{cout} ups={int(ups)} split {sk}: 8-wave {wd*1e3:7.1f} us ({fl/wd/1e9:6.0f} TF/s) | shipped {a*1e3:7.1f} us ({fl/a/1e9:6.0f} TF/s) | halo4 {b*1e3:7.1f} us ({fl/b/1e9:6.0f} TF/s = {fl/b/1e9/2500:.3f}) | "
          f"+stats {c*1e3:7.1f} us | diff vs shipped {err:.2e}", flush=True)
