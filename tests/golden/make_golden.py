#!/usr/bin/env python
"""Generate golden vectors from the *importable* pieces of the reference.

Run in the build container only (it reads /root/reference; the GPU box has no
reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs `tests/golden/g*.npz` (inputs + the reference's outputs, a few MB in
total).  Only data is written -- no reference source or bytecode.  Unused
top-level imports of the reference (`decord`, `cv2`, `nltk`, `imageio`,
`torchvision`) are replaced by inert stubs; the functions exercised here never
touch them.

G1 `fmc.data.dataset.ray_condition`            (dataset.py:930-972)
G2 `fmc.adapter.Adapter`                       (adapter.py:109-192)
G3 `fmc.util.get_traj_features_v2`             (util.py:147-213)
G4 `fmc.data.utils.create_relative_matrix_of_cam_list` (data/utils.py:148-165)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import torch


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    _stub("decord", VideoReader=object, cpu=lambda *a, **k: None)
    _stub("cv2")
    _stub("imageio")
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.transforms.functional = _stub("torchvision.transforms.functional")
    nl = _stub("nltk")
    nl.stem = _stub("nltk.stem", WordNetLemmatizer=object, PorterStemmer=object)


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def reseed_module(m, seed, std=0.05):
    """zero-initialised layers would make every output 0: give all params seeded values."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * std)


def smooth_c2w(B, Fr, seed):
    """relative c2w trajectory: identity first frame, small rotations, translations / 1200-like."""
    rng = np.random.default_rng(seed)
    out = np.zeros((B, Fr, 4, 4))
    for b in range(B):
        ang, t = np.zeros(3), np.zeros(3)
        for f in range(Fr):
            if f:
                ang += rng.normal(0, 0.03, 3)
                t += rng.normal(0, 0.05, 3)
            cx, cy, cz = np.cos(ang)
            sx, sy, sz = np.sin(ang)
            Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
            Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
            out[b, f, :3, :3] = Rz @ Ry @ Rx
            out[b, f, :3, 3] = t
            out[b, f, 3, 3] = 1
    return out


def circle_masks(n_obj, Fr, H, W, seed):
    """Gaussian circle masks by the formula of dataset.py:5365-5380 (numpy only)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.ogrid[:H, :W]
    ctr = rng.uniform([W * 0.2, H * 0.2], [W * 0.8, H * 0.8], size=(n_obj, 2))
    rad = rng.uniform(min(H, W) * 0.15, min(H, W) * 0.35, size=n_obj)
    frames = []
    for f in range(Fr):
        ctr = ctr + rng.normal(0, 2.0, size=ctr.shape)
        ms = []
        for o in range(n_obj):
            c, r = ctr[o], rad[o]
            dist = np.sqrt((xx - c[0]) ** 2 + (yy - c[1]) ** 2)
            g = np.exp(-0.5 * (dist / (r / 2)) ** 2)
            g = g / g.max()
            disc = ((xx - int(c[0])) ** 2 + (yy - int(c[1])) ** 2) <= int(r) ** 2
            ms.append((disc * g)[None])
        frames.append(torch.from_numpy(np.stack(ms)).float())          # [n_obj,1,H,W]
    return frames


def main():
    install_stubs()
    sys.path.insert(0, REF)
    from fmc.data.dataset import ray_condition
    from fmc.adapter import Adapter
    from fmc.util import get_traj_features_v2
    from fmc.data.utils import create_relative_matrix_of_cam_list

    # ---- G1: Pluecker -------------------------------------------------------
    B, Fr, H, W = 2, 4, 16, 24
    c2w = torch.from_numpy(smooth_c2w(B, Fr, 11)).float()
    K = torch.tensor([W * 1.1, W * 0.9, W / 2.0, H / 2.0]).view(1, 1, 4).repeat(B, Fr, 1)
    K[1] = torch.tensor([float(W), float(H), 0.0, 0.0])           # reference-style intrinsics (dataset.py:5451)
    out = ray_condition(K, c2w, H, W, device="cpu")
    # one full-resolution frame, rows subsampled to keep the file small
    Hb, Wb = 320, 512
    c2w_b = torch.from_numpy(smooth_c2w(1, 2, 12)).float()
    K_b = torch.tensor([float(Wb), float(Wb), 0.0, 0.0]).view(1, 1, 4).repeat(1, 2, 1)
    out_b = ray_condition(K_b, c2w_b, Hb, Wb, device="cpu")[:, :, ::37]
    np.savez_compressed(os.path.join(HERE, "g1_plucker.npz"), K=K.numpy(), c2w=c2w.numpy(), H=H, W=W,
                        out=out.numpy(), K_b=K_b.numpy(), c2w_b=c2w_b.numpy(), H_b=Hb, W_b=Wb, row_step=37,
                        out_b=out_b.numpy())

    # ---- G2: Adapter ----------------------------------------------------------
    torch.manual_seed(0)
    small = dict(channels=[16, 32, 64, 64], nums_rb=2, cin=832, sk=True, use_conv=False,
                 use_pre_zero_conv=True, use_post_zero_conv=True)
    ad = Adapter(**small).eval()
    reseed_module(ad, 21)
    x = seeded((3, 13, 64, 64), 22)
    m = (seeded((3, 1, 64, 64), 23) > 0.3).float() * torch.rand(3, 1, 64, 64, generator=torch.Generator().manual_seed(24))
    with torch.no_grad():
        f_mask = ad(x, m)
        f_nomask = ad(x, None)
    sd = {k: v.numpy() for k, v in ad.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "g2_adapter_small.npz"), x=x.numpy(), mask=m.numpy(),
                        **{f"out_mask_{i}": t.numpy() for i, t in enumerate(f_mask)},
                        **{f"out_nomask_{i}": t.numpy() for i, t in enumerate(f_nomask)},
                        **{f"sd::{k}": v for k, v in sd.items()})
    # obj.yaml widths: record only key names/shapes + an output checksum (152.5 M params are not committed)
    full = Adapter(channels=[320, 640, 1280, 1280], nums_rb=2, cin=832, sk=True, use_conv=False,
                   use_pre_zero_conv=True, use_post_zero_conv=True)
    keys = np.array(list(full.state_dict().keys()))
    shapes = np.array([str(tuple(v.shape)) for v in full.state_dict().values()])
    np.savez_compressed(os.path.join(HERE, "g2_adapter_full_keys.npz"), keys=keys, shapes=shapes,
                        n_params=sum(p.numel() for p in full.parameters()))

    # ---- G3: get_traj_features_v2 --------------------------------------------
    Bt, Ft, Ht, Wt, n_obj = 1, 3, 64, 64, 3
    rng = np.random.default_rng(31)
    masks = [circle_masks(n_obj, Ft, Ht, Wt, 32)]                         # [B][F] -> [n_obj,1,H,W]
    infos = [[rng.normal(0, 0.5, size=(n_obj, 12)) for _ in range(Ft)]]   # float64 like the dataset
    with torch.no_grad():
        feats = get_traj_features_v2(infos, masks, ad, False, 0.0, [False], "cpu", torch.float32)

    class _Tap(torch.nn.Module):                                          # capture what the Adapter is fed
        def forward(self, x, m):
            self.x, self.m = x, m
            return [x]
    tap = _Tap()
    get_traj_features_v2(infos, masks, tap, False, 0.0, [False], "cpu", torch.float32)
    np.savez_compressed(os.path.join(HERE, "g3_traj.npz"),
                        masks=np.stack([torch.stack(mf).numpy() for mf in masks]),       # [B,F,n,1,H,W]
                        infos=np.stack([np.stack(i) for i in infos]),                    # [B,F,n,12]
                        raster=tap.x.numpy(), raster_mask=tap.m.numpy(),
                        **{f"feat_{i}": t.numpy() for i, t in enumerate(feats)})

    # G3b: the classifier-free "null object condition" branch (util.py:194-199) with ratio 1.0 -> every clip dropped: the
    # features are zeroed but the Adapter still gets the real mask
    with torch.no_grad():
        feats_null = get_traj_features_v2(infos, masks, ad, True, 1.0, [False], "cpu", torch.float32)
    np.savez_compressed(os.path.join(HERE, "g3_traj_null.npz"),
                        **{f"feat_{i}": t.numpy() for i, t in enumerate(feats_null)})

    # ---- G4: relative camera poses -------------------------------------------
    abs_rt = torch.from_numpy(smooth_c2w(1, 6, 41)[0][:, :3]).double()
    abs_rt[:, :, 3] *= 300.0
    rel = create_relative_matrix_of_cam_list([abs_rt[i] for i in range(6)], scale_T=1200)
    np.savez_compressed(os.path.join(HERE, "g4_relpose.npz"), abs_rt=abs_rt.numpy(), rel=rel.numpy(), scale_T=1200)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
