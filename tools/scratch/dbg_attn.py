import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synfmc_amd import hip_ops as K
from tests.test_gpu_kernels import oracle_attention
B, S, H, D = 1, 256, 2, 40
C = H * D
q = torch.randn(B, S, C, generator=torch.Generator().manual_seed(16))
k = torch.randn(B, S, C, generator=torch.Generator().manual_seed(17))
v = torch.randn(B, S, C, generator=torch.Generator().manual_seed(18))
k[0, 200] = q[0, 7] * 6.0
ref = oracle_attention(q, k, v, H)
ref64 = oracle_attention(q.double(), k.double(), v.double(), H)
out = K.spatial_attention(q.cuda(), k.cuda(), v.cuda(), H).cpu()
err = (out - ref).abs()
print("max err vs fp32 oracle", err.max().item(), "rel", (err.max() / ref.abs().max()).item(), "at", (err == err.max()).nonzero()[0].tolist())
print("fp32 oracle vs fp64", ((ref - ref64).abs().max() / ref64.abs().max()).item(), " ours vs fp64", ((out - ref64).abs().max() / ref64.abs().max()).item())
rows = err.amax(dim=-1)[0]
print("worst rows", rows.topk(5))
