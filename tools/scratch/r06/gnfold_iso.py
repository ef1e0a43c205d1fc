"""Isolated timing of the GroupNorm -> proj_in pair against the folded form (level 0: 32 images x 2560 pixels x 320 channels)."""
import torch, bench
from synfmc_amd import hip_ops as K
from synfmc_amd import _lib
dev = torch.device("cuda:0"); dt = torch.bfloat16
n, hw, C, N = 32, 2560, 320, 320
x = torch.randn(n, hw, C, device=dev, dtype=dt)
v = x.float().view(n, 16, 160, 32, 10)
part = torch.stack([v.sum(dim=(2, 4)), (v * v).sum(dim=(2, 4))], dim=-1).contiguous()
g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
w = torch.randn(N, C, device=dev, dtype=dt) * C ** -0.5
bias = torch.randn(N, device=dev, dtype=dt)
ln = K.LnSpec(gamma=g, beta=b, eps=1e-5, pe=None, pe_inner=1, pe_frames=1, key=("t",), stats_only=True)
with torch.no_grad():
    for spec in (None, ln):
        t_apply = bench._time_launch(lambda: K.groupnorm_apply(x, g, b, 32, 1e-6, False, part), 30)
        y = K.groupnorm_apply(x, g, b, 32, 1e-6, False, part)
        K.linear(y, w, bias, ln=spec)
        t_lin = bench._time_launch(lambda: K.linear(y, w, bias, ln=spec), 30)
        t_fold = bench._time_launch(lambda: K.linear_gnfold(x, (part, C), g, b, 32, 1e-6, w, bias, spec), 30)
        w_img = torch.empty(n, N, C, dtype=dt, device=dev); b_img = torch.empty(n, N, dtype=torch.float32, device=dev)
        lib = _lib.load()
        t_k = bench._time_launch(lambda: lib.fmc_groupnorm_fold_linear(part.data_ptr(), 16, g.data_ptr(), b.data_ptr(), w.data_ptr(), bias.data_ptr(), w_img.data_ptr(),
                                                                      b_img.data_ptr(), n, hw, C, 32, N, 1e-6, 1, K._stream()), 30)
        print("ln" if spec else "plain", "apply %.1f us  linear %.1f  | folded pair %.1f  (fold kernel %.1f)" % (t_apply * 1e3, t_lin * 1e3, t_fold * 1e3, t_k * 1e3),
              "arm", K._choice.get(("lin", n * hw, N, C, True, 0, 0)))
