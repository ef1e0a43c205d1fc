"""CPU checks: restated diffusers primitives vs torch built-ins / closed forms, the C ABI exports every symbol the
header declares, DDIM host logic, and the data-parallel sharding with a world_size-2 gloo group."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import diffusers_restated as OD
from oracle import fmc_modules as OM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported_and_typed():
    from synfmc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fmc_hip.h")).read()
    declared = set(re.findall(r"\b(fmc_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), "ctypes table and header disagree"
    lib = _lib.load()                       # raises if the .so is missing: there is no fallback
    for name in declared:
        assert hasattr(lib, name), f"libfmc_hip.so does not export {name}"
    assert lib.fmc_version() == 100
    # error path without touching a GPU: NULL pointers are rejected before any launch
    assert lib.fmc_geglu_fwd(None, None, 4, 8, 0, None) == -5
    assert b"NULL" in lib.fmc_last_error()


def test_attention_matches_sdpa():
    torch.manual_seed(0)
    attn = OD.Attention(query_dim=64, heads=4, dim_head=16)
    x = torch.randn(2, 10, 64)
    out = attn(x)
    q, k, v = attn.to_q(x), attn.to_k(x), attn.to_v(x)
    sp = lambda t: t.view(2, 10, 4, 16).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(2, 10, 64)
    ref = attn.to_out[0](ref)
    assert torch.allclose(out, ref, atol=1e-5)


def test_geglu_is_erf_gelu_on_second_half():
    m = OD.GEGLU(8, 16)
    x = torch.randn(3, 8)
    a, g = m.proj(x).chunk(2, -1)
    assert torch.allclose(m(x), a * 0.5 * g * (1 + torch.erf(g / 2 ** 0.5)), atol=1e-6)


def test_timesteps_flip_sin_to_cos():
    t = torch.tensor([0, 7, 999])
    emb = OD.Timesteps(320, True, 0)(t)
    k = torch.arange(160, dtype=torch.float32)
    ang = t[:, None].float() * torch.exp(-np.log(10000.0) * k / 160)[None]
    assert torch.allclose(emb[:, :160], torch.cos(ang), atol=1e-6) and torch.allclose(emb[:, 160:], torch.sin(ang), atol=1e-6)


def test_resnet_block_order_of_operations():
    torch.manual_seed(1)
    blk = OD.ResnetBlock2D(in_channels=32, out_channels=64, temb_channels=16, groups=8, eps=1e-5)
    x, temb = torch.randn(2, 32, 6, 5), torch.randn(2, 16)
    h = blk.conv1(F.silu(F.group_norm(x, 8, blk.norm1.weight, blk.norm1.bias, 1e-5)))
    h = h + blk.time_emb_proj(F.silu(temb))[:, :, None, None]
    h = blk.conv2(F.silu(F.group_norm(h, 8, blk.norm2.weight, blk.norm2.bias, 1e-5)))
    assert torch.allclose(blk(x, temb), blk.conv_shortcut(x) + h, atol=1e-5)


def test_ddim_closed_form_and_product_scheduler_host_logic():
    from synfmc_amd.schedulers import DDIMScheduler
    kw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, clip_sample=False)
    for sched_name in ("linear", "scaled_linear"):
        o = OD.DDIMScheduler(beta_schedule=sched_name, **kw)
        p = DDIMScheduler(beta_schedule=sched_name, **kw)
        assert torch.equal(o.alphas_cumprod, p.alphas_cumprod)
        for n in (25, 50):
            o.set_timesteps(n)
            p.set_timesteps(n)
            assert o.timesteps.tolist() == p._timesteps_host == [(n - 1 - i) * (1000 // n) + 1 for i in range(n)]
            t = o.timesteps[3].item()
            a_t, a_prev = p._alphas(t)
            assert a_t == pytest.approx(float(o.alphas_cumprod[t])) and a_prev == pytest.approx(float(o.alphas_cumprod[t - 1000 // n]))
        assert p._alphas(1)[1] == 1.0                                  # last step lands on alpha = 1
    x0, noise = torch.randn(2, 4, 3, 2, 2), torch.randn(2, 4, 3, 2, 2)
    t = torch.tensor([10, 900])
    assert torch.equal(o.add_noise(x0, noise, t), p.add_noise(x0, noise, t))
    # one step: x_{t-1} = sqrt(a')*x0_hat + sqrt(1-a')*eps
    o.set_timesteps(50)
    x, eps = torch.randn(1, 4), torch.randn(1, 4)
    a, ap = o.alphas_cumprod[981], o.alphas_cumprod[961]
    want = ap.sqrt() * (x - (1 - a).sqrt() * eps) / a.sqrt() + (1 - ap).sqrt() * eps
    assert torch.allclose(o.step(eps, 981, x).prev_sample, want, atol=1e-6)


def test_positional_encoding_added_after_layernorm():
    torch.manual_seed(2)
    blk = OM.TemporalTransformerBlock(dim=32, num_attention_heads=4, attention_head_dim=8,
                                      attention_block_types=("Temporal_Self",), temporal_position_encoding=True,
                                      temporal_position_encoding_max_len=16)
    x = torch.randn(3, 16, 32)
    a = blk.attention_blocks[0]
    n = blk.norms[0](x)
    want = a.processor(a, n + a.pos_encoder.pe[:, :16]) + x
    want = blk.ff(blk.ff_norm(want)) + want
    assert torch.allclose(blk(x), want, atol=1e-5)


def test_clip_shard_matches_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    from synfmc_amd.dp import clip_shard
    for n, w in ((10, 4), (8, 8), (7, 2), (3, 8)):
        for shuffle in (False, True):
            for r in range(w):
                s = DistributedSampler(range(n), num_replicas=w, rank=r, shuffle=shuffle, seed=42)
                s.set_epoch(3)
                assert list(s) == clip_shard(n, r, w, shuffle=shuffle, seed=42, epoch=3), (n, w, r, shuffle)


_GLOO_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from synfmc_amd import dp
dist.init_process_group("gloo", init_method="env://")
r, w = dp.rank(), dp.world()
mine = dp.clip_shard(9, r, w, shuffle=True, seed=7)
gathered = [None] * w
dist.all_gather_object(gathered, mine)
dp.barrier()
tmax = dp.max_over_ranks(1.0 + r)
tsum = dp.sum_over_ranks(float(len(mine)))
if r == 0:
    print(json.dumps({"world": w, "shards": gathered, "tmax": tmax, "tsum": tsum}))
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29731", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["world"] == 2 and res["tmax"] == 2.0 and res["tsum"] == 10.0
    flat = sorted(res["shards"][0] + res["shards"][1])
    assert set(flat) == set(range(9)) and len(flat) == 10           # 9 clips padded to 10 by wrapping


_GLOO_REDUCER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from synfmc_amd.training import GradAllReducer, broadcast_parameters
dist.init_process_group("gloo", init_method="env://")
r, w = dist.get_rank(), dist.get_world_size()
mode = sys.argv[2]
torch.manual_seed(100 + r)                       # different init per rank: broadcast must fix it
net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                          torch.nn.Linear(32, 4))
unused = torch.nn.Linear(8, 8)                   # never reached by the loss on any rank (like the Adapter's level 3)
rank1_only = torch.nn.Linear(4, 4)               # reached on rank 1 only: must stay in the buckets everywhere
v0 = [p._version for p in net.parameters()]
broadcast_parameters(net); broadcast_parameters(unused); broadcast_parameters(rank1_only)
assert all(p._version > a for p, a in zip(net.parameters(), v0))      # derived-weight caches keyed on _version see it
params = list(net.parameters()) + list(unused.parameters()) + list(rank1_only.parameters())
red = GradAllReducer(params, bucket_bytes=3000, overlap=(mode != "sync"),
                     compress_dtype=torch.bfloat16 if mode == "bf16" else None)
assert len(red.buckets) > 2
x = torch.randn(5, 16, generator=torch.Generator().manual_seed(r))
def fwd(ps, xx):
    h = torch.relu(xx @ ps[0].t() + ps[1]); h = torch.relu(h @ ps[2].t() + ps[3]); return h @ ps[4].t() + ps[5]
def loss_of(ps, extra, xx, rank, step):
    y = fwd(ps, xx)
    if rank == 1:
        y = y @ extra[0].t() + extra[1]
    return y.pow(2).mean() * (step + 1)
for step in range(3):
    loss = loss_of(list(net.parameters()), list(rank1_only.parameters()), x, r, step)
    loss.backward()
    if step == 2:                                # a second backward in the same step must be refused, not mis-reduced
        red.finish()
        try:
            loss_of(list(net.parameters()), list(rank1_only.parameters()), x, r, step).backward()
            raised = False
        except RuntimeError as e:
            raised = "already reduced" in str(e)
        if r == 0:
            print(json.dumps({"step": step, "raised": raised}))
        break
    red.finish()
    live = list(net.parameters()) + list(rank1_only.parameters())
    got = torch.cat([p.grad.reshape(-1) for p in live]).clone()
    ref_net = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
    ref_extra = [p.detach().clone().requires_grad_(True) for p in rank1_only.parameters()]
    total = [torch.zeros_like(p) for p in ref_net + ref_extra]
    for rr in range(w):
        xr = torch.randn(5, 16, generator=torch.Generator().manual_seed(rr))
        gs = torch.autograd.grad(loss_of(ref_net, ref_extra, xr, rr, step), ref_net + ref_extra, allow_unused=True)
        total = [t + (g / w if g is not None else 0) for t, g in zip(total, gs)]
    want = torch.cat([t.reshape(-1) for t in total])
    err = ((got - want).abs().max() / want.abs().max()).item()
    unused_none = all(p.grad is None for p in unused.parameters())
    n_bucket_params = sum(len(b["params"]) for b in red.buckets)
    red.zero_grad()
    assert all(float(p.grad.abs().max()) == 0.0 for p in live)
    if r == 0:
        print(json.dumps({"step": step, "err": err, "buckets": len(red.buckets), "unused_none": unused_none,
                          "n_unused": len(red.unused), "n_bucket_params": n_bucket_params,
                          "wire_bytes": red.allreduce_bytes()}))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode,port,tol", [("overlap", 29732, 1e-6), ("sync", 29733, 1e-6), ("bf16", 29734, 2e-2)])
def test_two_rank_gloo_grad_allreduce(tmp_path, mode, port, tol):
    """2-rank all-reduce of the trainable subset: unused-parameter discovery (bitmap over ranks; a parameter used on one
    rank only stays), overlapped / post-backward / bf16-compressed exchange, refusal of a second backward per step."""
    script = tmp_path / "reducer.py"
    script.write_text(_GLOO_REDUCER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, mode],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 3 and all(l["err"] < tol for l in lines[:2])
    assert all(l["unused_none"] and l["n_unused"] == 2 and l["n_bucket_params"] == 8 for l in lines[:2])
    n_live = 16 * 32 + 32 + 32 * 32 + 32 + 32 * 4 + 4 + 4 * 4 + 4
    assert lines[1]["wire_bytes"] == n_live * (2 if mode == "bf16" else 4)
    assert lines[2]["raised"] is True


_GLOO_LATE = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from synfmc_amd.training import GradAllReducer, broadcast_parameters
dist.init_process_group("gloo", init_method="env://")
r, w = dist.get_rank(), dist.get_world_size()
torch.manual_seed(7)
net = torch.nn.Linear(8, 8)
late = torch.nn.Linear(8, 8)                     # unused in the discovery step, reached on RANK 1 ONLY from step 2 on
broadcast_parameters(net); broadcast_parameters(late)
params = list(net.parameters()) + list(late.parameters())
red = GradAllReducer(params, bucket_bytes=200, overlap=True)
x = torch.randn(4, 8, generator=torch.Generator().manual_seed(10 + r))
def loss_of(ps, ls, xx, rank, step):
    y = xx @ ps[0].t() + ps[1]
    if step >= 2 and rank == 1:
        y = y @ ls[0].t() + ls[1]
    return y.pow(2).mean()
out = []
for step in range(4):
    loss_of(list(net.parameters()), list(late.parameters()), x, r, step).backward()
    red.finish()
    ref_n = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
    ref_l = [p.detach().clone().requires_grad_(True) for p in late.parameters()]
    total = [torch.zeros_like(p) for p in ref_n + ref_l]
    for rr in range(w):
        xr = torch.randn(4, 8, generator=torch.Generator().manual_seed(10 + rr))
        gs = torch.autograd.grad(loss_of(ref_n, ref_l, xr, rr, step), ref_n + ref_l, allow_unused=True)
        total = [t + (g / w if g is not None else 0) for t, g in zip(total, gs)]
    err = 0.0
    for p, t in zip(list(net.parameters()) + list(late.parameters()), total):
        if p.grad is None:
            err = max(err, float(t.abs().max()))                 # a None gradient is only right where the mean gradient is zero
        else:
            err = max(err, float((p.grad - t).abs().max()))
    out.append({"step": step, "err": err, "n_unused": len(red.unused), "readmitted": red.readmitted,
                "late_grad_none": all(p.grad is None for p in late.parameters()),
                "n_bucket_params": sum(len(b["params"]) for b in red.buckets)})
    red.zero_grad()
# every rank must have reached the same verdicts
flat = torch.tensor([float(o["n_unused"]) for o in out] + [float(o["readmitted"]) for o in out])
other = flat.clone(); dist.all_reduce(other, op=dist.ReduceOp.MAX)
assert torch.equal(flat, other)
if r == 0:
    print(json.dumps(out))
dist.destroy_process_group()
'''


def test_two_rank_gloo_late_gradient_is_readmitted(tmp_path):
    """A parameter pruned as unused in the discovery step and reached LATER, on one rank only: its gradient is averaged
    over the ranks in that very step (no rank steps on a rank-local gradient) and it is back in the buckets from the next one."""
    script = tmp_path / "late.py"
    script.write_text(_GLOO_LATE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29741", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    o = json.loads([l for l in out.stdout.splitlines() if l.startswith("[")][-1])
    assert all(s["err"] < 1e-6 for s in o), o
    assert o[0]["n_unused"] == 2 and o[1]["n_unused"] == 2 and o[0]["late_grad_none"] and o[1]["late_grad_none"]
    assert o[2]["readmitted"] == 2 and not o[2]["late_grad_none"]               # averaged in the step it appeared ...
    assert o[3]["n_unused"] == 0 and o[3]["n_bucket_params"] == 4                # ... and bucketed afterwards


def test_discovery_step_must_see_a_gradient():
    from synfmc_amd.training import GradAllReducer
    lin = torch.nn.Linear(4, 4)
    red = GradAllReducer(lin.parameters())
    with pytest.raises(RuntimeError, match="no gradient hook fired"):
        red.finish()


def test_loss_and_timestep_sampling_match_oracle():
    from oracle import pipeline as OP
    from synfmc_amd.training import biased_timesteps, masked_mse_loss
    g = torch.Generator().manual_seed(0)
    pred, tgt = torch.randn(2, 4, 16, 8, 12, generator=g), torch.randn(2, 4, 16, 8, 12, generator=g)
    masks = torch.rand(2, 16, 64, 96, generator=g) > 0.7
    assert torch.allclose(masked_mse_loss(pred, tgt, masks, 0.3, 1.0), OP.stage3_loss(pred, tgt, masks, 0.3, 1.0), atol=1e-6)
    t = biased_timesteps(20000, 1000, 700, 0.8, "cpu", torch.Generator().manual_seed(1))
    frac_hi = (t >= 700).float().mean().item()
    assert abs(frac_hi - 0.8) < 0.02 and int(t.min()) >= 0 and int(t.max()) < 1000


_OVERLAY_WORKER = r'''
import ast, sys, types, json
sys.dont_write_bytecode = True
root, ref = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
# third-party packages the trainers / the reference's data + logging modules import that this container lacks: inert stubs
# (the user's environment has the real ones).  Nothing below stubs any `fmc` module.
def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent and parent in sys.modules: setattr(sys.modules[parent], leaf, m)
    return m
for name in ("decord", "cv2", "imageio", "nltk", "nltk.stem", "torchvision", "torchvision.transforms",
             "torchvision.transforms.functional", "termcolor", "omegaconf"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            stub(name)
sys.modules["decord"].__dict__.setdefault("VideoReader", object); sys.modules["decord"].__dict__.setdefault("cpu", lambda *a, **k: None)
sys.modules["nltk.stem"].__dict__.setdefault("WordNetLemmatizer", object); sys.modules["nltk.stem"].__dict__.setdefault("PorterStemmer", object)
sys.modules["termcolor"].__dict__.setdefault("colored", lambda s, *a, **k: s)
class _DDIM:                                                 # what `from diffusers import DDIMScheduler` yields in the trainers
    def __init__(self, **kw):
        self.config = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                           clip_sample=True, steps_offset=0, set_alpha_to_one=True, prediction_type="epsilon",
                           timestep_spacing="leading"); self.config.update(kw)
_DDIM.__name__ = "DDIMScheduler"
stub("diffusers", AutoencoderKL=object, DDIMScheduler=_DDIM)
stub("diffusers.optimization", get_scheduler=lambda *a, **k: None)
stub("diffusers.utils", check_min_version=lambda v: None)
stub("diffusers.models"); stub("diffusers.models.attention_processor", AttnProcessor=object)

import synfmc_amd
mode = synfmc_amd.install_as_fmc(ref)
ns, done = {}, []
for trainer in ("train_cam_obj_ctrl.py", "train_cam_ctrl.py"):
    tree = ast.parse(open(f"{ref}/{trainer}").read())
    for node in tree.body:                                   # the trainers' own top-level `from fmc... import ...` lines
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "fmc":
            exec(compile(ast.Module([node], []), trainer, "exec"), ns)
            done.append(f"{trainer}:{node.lineno}")
import fmc.data.dataset, fmc.util, fmc.utils.util, fmc.data.utils
out = {"mode": mode, "n_lines": len(done),
       "unet": ns["UNet3DConditionModelCamObjCond"].__module__, "posecond": ns["UNet3DConditionModelPoseCond"].__module__,
       "pipe": ns["CameraObjCtrlPipeline"].__module__, "pipe_cam": ns["CameraCtrlPipeline"].__module__,
       "adapter": ns["Adapter"].__module__, "encoder": ns["CameraPoseEncoder"].__module__,
       "wrapper": ns["CamObjPoseAdaptor"].__module__, "patch_fn": ns["Adapted_CrossAttnDownBlock3D_forward"].__module__,
       "ray_condition": ns["ray_condition"].__module__, "traj": ns["get_traj_features_v2"].__module__,
       "dataset_cls": ns["UnrealTrajVideoDataset"].__module__, "dataset_file": fmc.data.dataset.__file__,
       "logger": ns["setup_logger"].__module__, "logger_file": fmc.utils.util.__file__,
       "abs_matrix": ns["create_absolute_matrix_from_ref_cam_list"].__module__,
       "kept_reference_fn": fmc.data.dataset._reference_ray_condition.__module__}
# a diffusers-style scheduler object handed to the pipeline constructor is re-expressed as the native one
from synfmc_amd.schedulers import DDIMScheduler
pipe = ns["CameraObjCtrlPipeline"](None, None, None, None, _DDIM(beta_start=0.00085, beta_end=0.012, steps_offset=1,
                                                               clip_sample=False), None)
out["sched"] = [type(pipe.scheduler).__module__, pipe.scheduler.config.steps_offset, float(pipe.scheduler.betas[0])]
print(json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isdir("/root/reference/fmc"), reason="build container only: needs the reference checkout")
def test_install_as_fmc_overlays_the_reference_package():
    """`install_as_fmc(reference_root)`: every `from fmc... import ...` line of the two trainers resolves -- hot-path names
    to this package, dataset classes / logging / pose math to the reference's own modules, with `ray_condition` and
    `get_traj_features_v2` patched into them.  Runs in a subprocess (it rewires `sys.modules`); only third-party packages
    missing from this container are stubbed, no `fmc` module is."""
    import json
    out = subprocess.run([sys.executable, "-c", _OVERLAY_WORKER, ROOT, "/root/reference"], capture_output=True, text=True,
                         timeout=300, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["mode"] == "overlay" and r["n_lines"] >= 17
    for k in ("unet", "posecond", "pipe", "pipe_cam", "adapter", "encoder", "wrapper", "patch_fn", "ray_condition", "traj"):
        assert r[k].startswith("synfmc_amd."), (k, r[k])
    for k in ("dataset_cls", "logger", "abs_matrix", "kept_reference_fn"):
        assert r[k].startswith("fmc."), (k, r[k])
    assert r["dataset_file"].startswith("/root/reference/") and r["logger_file"].startswith("/root/reference/")
    assert r["sched"][0] == "synfmc_amd.schedulers" and r["sched"][1] == 1 and abs(r["sched"][2] - 0.00085) < 1e-9


def test_install_as_fmc_standalone_mode():
    """Without a reference checkout on the path the package itself answers to `fmc` for the hot-path modules."""
    code = ("import sys; sys.path.insert(0, sys.argv[1]); import synfmc_amd; m = synfmc_amd.install_as_fmc(); "
            "from fmc.models.unet_cam_obj import UNet3DConditionModelCamObjCond as U; from fmc.adapter import Adapter; "
            "from fmc.data.dataset import ray_condition; from fmc.util import get_traj_features_v2; print(m, U.__module__)")
    out = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[-2:] == ["standalone", "synfmc_amd.models.unet"]


def test_bench_entry_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must become 2 ranks (re-exec under torch.distributed.run),
    rendezvous on 127.0.0.1, time with barrier + max-over-ranks and print ONE JSON line whose `n_gpus` is the world size a
    collective returned.  `--dry-run` swaps the model for a CPU stub and RCCL for gloo; the launcher / timing / JSON code is
    the one the GPU runs use.  A WORLD_SIZE that disagrees with --gpus is an error."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "4",
                          "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = lines[0]
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["warmup"] == 1 and r["dry_run"] is True and r["scaling"] == "weak"
    assert abs(r["value"] - 2 * 4 / (r["ms_per_step"] * 4e-3)) / r["value"] < 1e-2          # whole-job aggregate
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                         text=True, env=dict(env, WORLD_SIZE="3", RANK="0"), timeout=120, cwd="/tmp")
    assert bad.returncode != 0 and "WORLD_SIZE=3" in (bad.stderr + bad.stdout)


def test_parity_mode_arm_decoding_reaches_the_160x320_kernels():
    """`_f32_arm` peeled the stream-K offsets (>= 256) before it looked for the 160 x 320 family (512.. / 528 / 544..), so fp32
    parity mode never ran the kernels the bf16 path autotunes to.  The family (any split) maps to ARM_160; stream-K / hybrid / k-lockstep forms
    map to their plain geometry; split-K ids to their geometry; the vendor arm and arm 15 to the kernel's own rule (0)."""
    from synfmc_amd import hip_ops as K
    for arm in list(range(K.ARM_160, K.ARM_160 + 5)) + list(range(K.ARM_160B, K.ARM_160B + 5)) + [K.ARM_256]:
        assert K._f32_arm_of(arm) == K.ARM_160, arm
    assert K._f32_arm_of(128 + 13) == 13 and K._f32_arm_of(256 + 13) == 13 and K._f32_arm_of(384 + 13) == 13 and K._f32_arm_of(128 + 3) == 3
    assert K._f32_arm_of(2 + 16 * 2) == 2 and K._f32_arm_of(9 + 16) == 9
    assert K._f32_arm_of(0) == 0 and K._f32_arm_of(15) == 0 and K._f32_arm_of(11) == 11
    # what the decoder hands the C ABI for the arms of this round
    assert K._decode_arm(384 + 13, 1) == (13, -3) and K._decode_arm(256 + 13, 1) == (13, -2) and K._decode_arm(K.ARM_160B + 2, 1) == (18, 4)


_GLOO_ORDERINGS = r'''
import os, sys, json, hashlib
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from synfmc_amd.training import GradAllReducer, broadcast_parameters, optimizer_update
dist.init_process_group("gloo", init_method="env://")
r, w = dist.get_rank(), dist.get_world_size()


def run(mode):
    """Three optimisation steps in the ORDER bench.py's train mode runs them: `none` = eager, bucket all-reduces launched from inside the
    backward (overlap); `split` = [graph A: forward + backward into the flat buckets] | re-arm + all-reduce | [graph B: clip + AdamW + zero];
    `one` = the same three inside one captured graph.  (No HIP graphs on the CPU: what is exercised is the order of hooks, `finish()`,
    the host-side re-arming of the buckets the split form does after a replay, and the optimizer on bucket views.)"""
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(12, 24), torch.nn.SiLU(), torch.nn.Linear(24, 24), torch.nn.SiLU(), torch.nn.Linear(24, 3))
    unused = torch.nn.Linear(6, 6)
    broadcast_parameters(net); broadcast_parameters(unused)
    params = list(net.parameters()) + list(unused.parameters())
    red = GradAllReducer(params, bucket_bytes=1500, overlap=(mode == "none"))

    def fwd_bwd(step):
        x = torch.randn(7, 12, generator=torch.Generator().manual_seed(1000 * step + r))
        loss = net(x).pow(2).mean()
        loss.backward()
        return loss
    fwd_bwd(99); red.finish(); red.zero_grad()                 # discovery step: prunes `unused`
    trainable = red.parameters()
    opt = torch.optim.AdamW(trainable, lr=1e-2, weight_decay=1e-2)
    for step in range(3):
        fwd_bwd(step)
        if mode == "split":
            for b in red.buckets:                               # what the replay of graph A leaves to the host (bench.py train_main)
                b["launched"] = False
        red.finish()
        optimizer_update(trainable, opt, red, 1.0)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    return hashlib.sha256(flat.numpy().tobytes()).hexdigest(), len(red.unused)


out = {m: run(m) for m in ("none", "split", "one")}
digests = [out[m][0] for m in out]
gathered = [None] * w
dist.all_gather_object(gathered, digests)
if r == 0:
    print(json.dumps({"digests": digests, "ranks_equal": all(g == gathered[0] for g in gathered), "n_unused": [out[m][1] for m in out]}))
dist.destroy_process_group()
'''


def test_two_rank_gloo_step_orderings_agree(tmp_path):
    """the `split` and `one` step orderings of `bench.py --mode train` (graph | all-reduce | graph, and everything in
    one graph) have only ever been captured on one GPU.  Their ORDER of operations -- backward into the flat buckets without launching,
    host-side re-arming, `finish()`, clip + AdamW + zero on the bucket views -- runs here on two gloo ranks next to the eager overlapped form:
    after three steps all three leave bit-identical parameters, on both ranks."""
    script = tmp_path / "orderings.py"
    script.write_text(_GLOO_ORDERINGS)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29751", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    o = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert o["ranks_equal"] and len(set(o["digests"])) == 1, o
    assert o["n_unused"] == [2, 2, 2]


def test_fragment_order_weight_packers():
    """Host logic of the fused kernels' weight formats (no GPU): every packed element sits where the kernel's fragment addressing reads it.
    A fragment = one 16-row block x one 32-deep k-step = [lane = 16 * (k chunk of 8) + row][8]; streams are [wave | head][k-step][block]."""
    import torch
    from synfmc_amd import hip_ops as K
    g = torch.Generator().manual_seed(0)

    def frag(packed, off, lane, e):               # element e of lane's 16 bytes in the fragment starting at element offset `off`
        return packed[off + lane * 8 + e]

    # pack_w_frag80: wave w owns rows 80 w ..; fragment (w, g, nb) at ((w * KS + g) * 5 + nb) * 512
    for C in (640,):
        w = torch.randn(C, C, generator=g)
        p = K.pack_w_frag80(w)
        KS = C // 32
        for (wv, ks, nb, lane, e) in [(0, 0, 0, 0, 0), (C // 80 - 1, KS - 1, 4, 63, 7), (1, 3, 2, 37, 5)]:
            row, kq = lane & 15, lane >> 4
            assert frag(p, ((wv * KS + ks) * 5 + nb) * 512, lane, e) == w[80 * wv + 16 * nb + row, 32 * ks + 8 * kq + e]
    # pack_temporal_qkv80: head h, part (q | k | v), k-step, block
    w = torch.randn(3 * 640, 640, generator=g)
    p = K.pack_temporal_qkv80(w)
    for (h, part, ks, nb, lane, e) in [(0, 0, 0, 0, 0, 0), (7, 2, 19, 4, 63, 7), (3, 1, 5, 2, 21, 3)]:
        row, kq = lane & 15, lane >> 4
        off = h * 3 * 51200 + part * 51200 + (ks * 5 + nb) * 512
        assert frag(p, off, lane, e) == w[part * 640 + 80 * h + 16 * nb + row, 32 * ks + 8 * kq + e]
    # pack_xattn_q40: per head [10 k-steps][q0 | q1 | (q tail, zeros)]
    w = torch.randn(320, 320, generator=g)
    p = K.pack_xattn_q40(w)
    for (h, ks, nb, lane, e) in [(0, 0, 0, 0, 0), (7, 9, 1, 63, 7), (2, 4, 2, 5, 6), (2, 4, 2, 13, 6)]:
        row, kq = lane & 15, lane >> 4
        want = w[40 * h + 16 * nb + row, 32 * ks + 8 * kq + e] if 16 * nb + row < 40 else 0.0
        assert frag(p, h * 15360 + (ks * 3 + nb) * 512, lane, e) == want
    # pack_geglu_frag80: chunk c, wave wv: rows [v 0-15 | v 16-31 | v 32-39, g 32-39 | g 0-15 | g 16-31] of gated columns 40 NW c + 40 wv ..
    for C, cff in ((640, 640), (320, 320)):
        w = torch.randn(2 * cff, C, generator=g)
        p = K.pack_geglu_frag80(w)
        NW, KS = C // 80, C // 32
        for (c, wv, ks, nb, lane, e) in [(0, 0, 0, 0, 0, 0), (cff // (40 * NW) - 1, NW - 1, KS - 1, 4, 63, 7), (1, 1, 2, 2, 3, 1), (1, 1, 2, 2, 11, 1), (0, 2, 1, 3, 40, 2)]:
            row, kq = lane & 15, lane >> 4
            base = 40 * NW * c + 40 * wv
            src = [base + row, base + 16 + row, (base + 32 + row) if row < 8 else (cff + base + 32 + row - 8), cff + base + row, cff + base + 16 + row][nb]
            assert frag(p, (((c * NW + wv) * KS + ks) * 5 + nb) * 512, lane, e) == w[src, 32 * ks + 8 * kq + e]
    # pack_geglu_frag (csrc/geglu_pipe.hip): per group of G = 32 | 16 gated columns the rows [value blocks of 16 | gate blocks of 16], [group][k-step][block][lane][8]
    for C, cff, G in ((320, 256, 16), (320, 256, 32), (640, 384, 32)):
        w = torch.randn(2 * cff, C, generator=g)
        p = K.pack_geglu_frag(w, G)
        KS, NBK = C // 32, G // 8
        assert p.numel() == w.numel()
        for (grp, ks, nb, lane, e) in [(0, 0, 0, 0, 0), (cff // G - 1, KS - 1, NBK - 1, 63, 7), (1, 3, 1, 22, 5), (2, 1, NBK // 2, 9, 2)]:
            row, kq = lane & 15, lane >> 4
            half = NBK // 2
            src = (G * grp + 16 * nb + row) if nb < half else (cff + G * grp + 16 * (nb - half) + row)      # value blocks first, then gate blocks
            assert frag(p, ((grp * KS + ks) * NBK + nb) * 512, lane, e) == w[src, 32 * ks + 8 * kq + e]


# ---- f4: the VAE restatement (oracle/vae_restated.py) cross-checked the only way this image allows: every primitive against torch
# built-ins and against the U-Net restatement's counterparts (oracle/diffusers_restated.py, read against diffusers 0.24.0 by the round-5 judge), the
# module tree against the published SD-1.5 VAE (parameter count, checkpoint key names, channel ladder) ----------------------------------------------------
def test_vae_restatement_primitives_against_torch_and_the_unet_restatement():
    from oracle import vae_restated as OV
    torch.manual_seed(11)
    # ResnetBlock2D(temb_channels=None, eps 1e-6): functional form AND the U-Net restatement's block on the same weights
    for cin, cout in ((32, 32), (32, 64)):
        r = OV.Resnet(cin, cout, groups=8)
        x = torch.randn(2, cin, 7, 6)
        h = F.conv2d(F.silu(F.group_norm(x, 8, r.norm1.weight, r.norm1.bias, 1e-6)), r.conv1.weight, r.conv1.bias, padding=1)
        h = F.conv2d(F.silu(F.group_norm(h, 8, r.norm2.weight, r.norm2.bias, 1e-6)), r.conv2.weight, r.conv2.bias, padding=1)
        sc = x if cin == cout else F.conv2d(x, r.conv_shortcut.weight, r.conv_shortcut.bias)
        assert torch.allclose(r(x), sc + h, atol=1e-5)
        d = OD.ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=None, groups=8, eps=1e-6)
        assert sorted(d.state_dict()) == sorted(r.state_dict())
        d.load_state_dict(r.state_dict(), strict=True)
        assert torch.allclose(r(x), d(x, None), atol=1e-5)
    # Attention(heads=1, dim_head=c, residual_connection=True, norm_num_groups, bias=True): scaled-dot-product attention of the GroupNorm'd tokens + input
    a = OV.Attn(64, groups=8)
    x = torch.randn(2, 64, 5, 4)
    t = F.group_norm(x, 8, a.group_norm.weight, a.group_norm.bias, 1e-6).flatten(2).transpose(1, 2)
    o = F.scaled_dot_product_attention(a.to_q(t)[:, None], a.to_k(t)[:, None], a.to_v(t)[:, None], scale=64 ** -0.5)[:, 0]
    assert torch.allclose(a(x), a.to_out[0](o).transpose(1, 2).reshape(2, 64, 5, 4) + x, atol=1e-5)
    d = OD.Attention(query_dim=64, heads=1, dim_head=64, bias=True)                      # (same projections through the U-Net restatement's Attention)
    d.load_state_dict({k: v for k, v in a.state_dict().items() if not k.startswith("group_norm")}, strict=True)
    assert torch.allclose(a(x), d(t).transpose(1, 2).reshape(2, 64, 5, 4) + x, atol=1e-5)
    # Upsample2D / Downsample2D(padding=0): the U-Net restatement's classes on the same weights (nearest x2 + conv; pad (0, 1, 0, 1) + stride-2 conv)
    up, dn = OV.Up(16), OV.Down(16)
    x = torch.randn(2, 16, 6, 9)
    du, dd = OD.Upsample2D(16, use_conv=True, out_channels=16), OD.Downsample2D(16, use_conv=True, out_channels=16, padding=0, name="op")
    du.load_state_dict(up.state_dict(), strict=True)
    dd.load_state_dict({k: v for k, v in dn.state_dict().items()}, strict=True)
    assert torch.allclose(up(x), du(x), atol=1e-6) and torch.allclose(dn(x), dd(x), atol=1e-6)
    assert dn(x).shape == (2, 16, 3, 4)                                                 # (6 x 9 -> 3 x 4: the asymmetric pad, not padding=1's 3 x 5)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), dn.conv.weight, dn.conv.bias, stride=2)
    assert torch.allclose(dn(x), ref, atol=1e-6)


def test_vae_restatement_is_the_published_sd15_vae_tree():
    from oracle import vae_restated as OV
    with torch.device("meta"):
        vae = OV.AutoencoderKLFull()                                                    # SD-1.5 `vae/config.json`: 128 / 256 / 512 / 512, 2 layers per block
    sd = vae.state_dict()
    assert sum(v.numel() for v in sd.values()) == 83_653_863                            # the published parameter count of `AutoencoderKL` (sd-vae-ft-mse, SD-1.5)
    for key, shape in (("encoder.conv_in.weight", (128, 3, 3, 3)), ("encoder.down_blocks.0.downsamplers.0.conv.weight", (128, 128, 3, 3)),
                       ("encoder.down_blocks.1.resnets.0.conv_shortcut.weight", (256, 128, 1, 1)), ("encoder.mid_block.attentions.0.to_q.weight", (512, 512)),
                       ("encoder.mid_block.attentions.0.group_norm.weight", (512,)), ("encoder.conv_out.weight", (8, 512, 3, 3)),
                       ("quant_conv.weight", (8, 8, 1, 1)), ("post_quant_conv.weight", (4, 4, 1, 1)), ("decoder.conv_in.weight", (512, 4, 3, 3)),
                       ("decoder.mid_block.attentions.0.to_out.0.bias", (512,)), ("decoder.up_blocks.0.resnets.2.conv2.weight", (512, 512, 3, 3)),
                       ("decoder.up_blocks.2.resnets.0.conv_shortcut.weight", (256, 512, 1, 1)), ("decoder.up_blocks.2.upsamplers.0.conv.weight", (256, 256, 3, 3)),
                       ("decoder.up_blocks.3.resnets.2.norm2.weight", (128,)), ("decoder.conv_norm_out.weight", (128,)), ("decoder.conv_out.weight", (3, 128, 3, 3))):
        assert tuple(sd[key].shape) == shape, key
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in sd and "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd
    # shapes end to end on a small instance: x8 down to 2 x 4 moments, x8 up to 3 channels; the logvar clamp of DiagonalGaussianDistribution
    small = OV.AutoencoderKLFull((32, 64, 64, 64), groups=8)
    x = torch.randn(1, 3, 32, 48)
    with torch.no_grad():
        mean, logvar = small.encode_moments(x)
        assert mean.shape == logvar.shape == (1, 4, 4, 6) and float(logvar.max()) <= 20.0 and float(logvar.min()) >= -30.0
        assert small.decode(mean).shape == (1, 3, 32, 48)


def test_ff_tail_fold_is_the_two_layer_chain():
    """`hip_ops.fold_ff_tail`: `[g | h] [Wp W2 | Wp]^T + (Wp b2 + bp)` equals `proj_out(ff2(g) + b2 + h) + bp` (fp64 chain on the same bf16 inputs; the fold rounds
    the product weight once), with and without biases -- the algebra `fmc_linear_bf16_fftail` relies on (motion_module.py:130-134,295-299, unet_blocks.py:323-333)."""
    import torch
    from synfmc_amd import hip_ops as K
    g = torch.Generator().manual_seed(5)
    M, C, cff = 96, 64, 256
    bf = lambda t: t.to(torch.bfloat16)
    gd, h = bf(torch.randn(M, cff, generator=g)), bf(torch.randn(M, C, generator=g))
    w2, wp = bf(torch.randn(C, cff, generator=g) * cff ** -0.5), bf(torch.randn(C, C, generator=g) * C ** -0.5)
    b2, bp = bf(torch.randn(C, generator=g) * 0.2), bf(torch.randn(C, generator=g) * 0.2)
    for with_b in (True, False):
        wc, bc = K.fold_ff_tail(w2, b2 if with_b else None, wp, bp if with_b else None)
        assert wc.shape == (C, cff + C) and wc.dtype == torch.bfloat16 and torch.equal(wc[:, cff:], wp) and (bc is None) == (not with_b)
        chain = (gd.double() @ w2.double().t() + (b2.double() if with_b else 0) + h.double()) @ wp.double().t() + (bp.double() if with_b else 0)
        folded = torch.cat([gd, h], 1).double() @ wc.double().t() + (bc.double() if with_b else 0)
        assert ((folded - chain).abs().max() / chain.abs().max()).item() < 6e-3          # bf16 rounding of the folded weight (2^-9 per element, summed at random)


def test_groupnorm_fold_algebra_cancels_the_mean():
    """The arithmetic of `fmc_groupnorm_fold_linear` restated in torch: per-image weights `bf16(W rstd gamma)` and the fp32 bias row built from the ROUNDED weights
    give `proj(GroupNorm(x))` to bf16 weight rounding even when a group's mean is 50 sigma -- a bias row built from the un-rounded weights does not."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(9)
    n, hw, C, N, G = 2, 64, 64, 32, 32
    x = torch.randn(n, hw, C, generator=g) * 0.3 + 15.0                                  # |mean| = 50 sigma
    x = x.to(torch.bfloat16).double()
    gam, bet = torch.rand(C, generator=g).double() + 0.5, torch.randn(C, generator=g).double() * 0.3
    w, b = (torch.randn(N, C, generator=g) * C ** -0.5).to(torch.bfloat16).double(), torch.randn(N, generator=g).double() * 0.1
    want = F.linear(F.group_norm(x.transpose(1, 2), G, gam, bet, 1e-6).transpose(1, 2), w, b)
    v = x.view(n, hw, G, C // G)
    mean = v.mean(dim=(1, 3))
    rstd = (v.var(dim=(1, 3), unbiased=False) + 1e-6).rsqrt()
    a = rstd.repeat_interleave(C // G, 1) * gam[None]                                     # [n, C]
    mu = mean.repeat_interleave(C // G, 1)
    w_exact = w[None] * a[:, None, :]
    w_img = w_exact.to(torch.bfloat16).double()                                           # what the GEMM multiplies with
    bias_rounded = b[None] + (w @ bet)[None] - torch.einsum("inc,ic->in", w_img, mu)      # the kernel's bias row
    bias_exact = b[None] + (w @ bet)[None] - torch.einsum("inc,ic->in", w_exact, mu)      # the tempting one
    got = torch.einsum("imc,inc->imn", x, w_img) + bias_rounded[:, None, :]
    bad = torch.einsum("imc,inc->imn", x, w_img) + bias_exact[:, None, :]
    scale = want.abs().max()
    assert ((got - want).abs().max() / scale).item() < 1e-2
    assert ((bad - want).abs().max() / scale).item() > 5 * ((got - want).abs().max() / scale).item()
