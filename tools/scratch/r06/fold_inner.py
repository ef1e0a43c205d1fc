"""Would the FF2 + proj_out fold pay at the inner levels (library arm)?  pair of launches vs one K = 5C + C launch + the copy of h into the operand buffer."""
import torch, bench
from synfmc_amd import hip_ops as K
dev = torch.device("cuda:0")
dt = torch.bfloat16
for (M, C) in ((5120, 1280), (1280, 1280), (20480, 640)):
    g = torch.randn(M, 4 * C, device=dev, dtype=dt)
    h = torch.randn(M, C, device=dev, dtype=dt)
    x = torch.randn(M, C, device=dev, dtype=dt)
    w2 = torch.randn(C, 4 * C, device=dev, dtype=dt) * (4 * C) ** -0.5
    wp = torch.randn(C, C, device=dev, dtype=dt) * C ** -0.5
    b = torch.randn(C, device=dev, dtype=dt)
    wc, bc = K.fold_ff_tail(w2, b, wp, b)
    buf = torch.randn(M, 5 * C, device=dev, dtype=dt)
    with torch.no_grad():
        for _ in range(2):
            K.linear(g, w2, b, residual=h); K.linear(h, wp, b, residual=x); K.linear(buf, wc, bc, residual=x)
        t_ff2 = bench._time_launch(lambda: K.linear(g, w2, b, residual=h), 30)
        t_po = bench._time_launch(lambda: K.linear(h, wp, b, residual=x), 30)
        t_fold = bench._time_launch(lambda: K.linear(buf, wc, bc, residual=x), 30)
        t_copy = bench._time_launch(lambda: buf[:, 4 * C:].copy_(h), 30)
        t_pair = bench._time_launch(lambda: (K.linear(g, w2, b, residual=h), K.linear(h, wp, b, residual=x)), 30)
        t_new = bench._time_launch(lambda: (buf[:, 4 * C:].copy_(h), K.linear(buf, wc, bc, residual=x)), 30)
    print(f"M={M} C={C}: ff2 {t_ff2*1e3:.1f} us  proj_out {t_po*1e3:.1f}  folded {t_fold*1e3:.1f}  copy {t_copy*1e3:.1f} | pair {t_pair*1e3:.1f}  copy+folded {t_new*1e3:.1f}",
          "arms", K._choice.get(("lin", M, C, 4 * C, True, 1, 0)), K._choice.get(("lin", M, C, C, True, 1, 0)), K._choice.get(("lin", M, C, 5 * C, True, 1, 0)))
