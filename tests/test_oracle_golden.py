"""The oracle against vectors produced by the reference itself (tests/golden/make_golden.py).
CPU only.  Tolerances: fp32 round-off of re-associated sums (1e-5 rel-inf unless stated)."""
import os

import numpy as np
import torch

from oracle import conditioning as C
from oracle import fmc_modules as M


def rel_inf(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_g1_plucker(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_plucker.npz"))
    out = C.ray_condition(torch.from_numpy(g["K"]), torch.from_numpy(g["c2w"]), int(g["H"]), int(g["W"]))
    assert out.shape == g["out"].shape
    assert rel_inf(out, g["out"]) < 1e-6
    out_b = C.ray_condition(torch.from_numpy(g["K_b"]), torch.from_numpy(g["c2w_b"]), int(g["H_b"]), int(g["W_b"]))
    assert rel_inf(out_b[:, :, :: int(g["row_step"])], g["out_b"]) < 1e-6


def test_g1_to_plucker_embedding_layout(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_plucker.npz"))
    emb = C.to_plucker_embedding(torch.from_numpy(g["c2w"][:, :, :3]), torch.from_numpy(g["K"]),
                                 (int(g["H"]), int(g["W"])))
    assert emb.shape == (2, 4, 6, 16, 24)
    assert rel_inf(emb.permute(0, 1, 3, 4, 2), g["out"]) < 1e-6


def _small_adapter(g):
    ad = M.Adapter(channels=[16, 32, 64, 64], nums_rb=2, cin=832, sk=True, use_conv=False,
                   use_pre_zero_conv=True, use_post_zero_conv=True).eval()
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    assert set(sd) == set(ad.state_dict()), "state-dict keys must equal the reference's"
    ad.load_state_dict(sd, strict=True)
    return ad


def test_g2_adapter(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_adapter_small.npz"))
    ad = _small_adapter(g)
    with torch.no_grad():
        fm = ad(torch.from_numpy(g["x"]), torch.from_numpy(g["mask"]))
        fn = ad(torch.from_numpy(g["x"]), None)
    for i in range(4):
        assert rel_inf(fm[i], g[f"out_mask_{i}"]) < 1e-5
        assert rel_inf(fn[i], g[f"out_nomask_{i}"]) < 1e-5


def test_g2_adapter_full_keys(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_adapter_full_keys.npz"))
    with torch.device("meta"):
        ad = M.Adapter(channels=[320, 640, 1280, 1280], nums_rb=2, cin=832, sk=True, use_conv=False,
                       use_pre_zero_conv=True, use_post_zero_conv=True)
    sd = ad.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert sum(p.numel() for p in ad.parameters()) == int(g["n_params"]) == 152510656


def test_g3_rasterise_and_traj_features(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_traj.npz"))
    g2 = np.load(os.path.join(golden_dir, "g2_adapter_small.npz"))
    masks = [[torch.from_numpy(g["masks"][b, f]) for f in range(g["masks"].shape[1])]
             for b in range(g["masks"].shape[0])]
    infos = [[g["infos"][b, f] for f in range(g["infos"].shape[1])] for b in range(g["infos"].shape[0])]
    feats, m = C.rasterize_objects(infos, masks)
    assert torch.equal(m, torch.from_numpy(g["raster_mask"]))
    assert rel_inf(feats, g["raster"]) < 1e-7
    # overlapping objects: a later object must overwrite an earlier one
    both = (g["masks"][0, 0, 0, 0] > 0) & (g["masks"][0, 0, 2, 0] > 0)
    assert both.any()
    ad = _small_adapter(g2)
    with torch.no_grad():
        out = C.get_traj_features(infos, masks, ad)
    for i in range(4):
        assert out[i].shape == g[f"feat_{i}"].shape
        assert rel_inf(out[i], g[f"feat_{i}"]) < 1e-5
    # G3b: cfg_random_null_om=True with ratio 1.0 (every clip dropped): zero features, REAL mask -> non-zero output
    gn = np.load(os.path.join(golden_dir, "g3_traj_null.npz"))
    with torch.no_grad():
        out_null = C.get_traj_features(infos, masks, ad, null_clips=range(len(infos)))
    for i in range(4):
        assert i > 0 or float(np.abs(gn[f"feat_{i}"]).max()) > 0      # (level 3 is one pixel whose nearest mask sample is 0)
        assert rel_inf(out_null[i], gn[f"feat_{i}"]) < 1e-5


def test_g4_relative_pose(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_relpose.npz"))
    rel = C.relative_cam_poses(g["abs_rt"], scale_T=float(g["scale_T"]))
    assert np.abs(rel - g["rel"]).max() < 1e-12


def test_g5_reference_unet_over_restated_diffusers(golden_dir):
    """G5: the reference's own `UNet3DConditionModelCamObjCond` (+ processors, motion modules, blocks, the
    `Adapted_*_forward` patch, `CameraPoseEncoder`), run by `tests/golden/make_golden_g5.py` over the restated diffusers
    primitives with the oracle's seeded weights (strict state-dict load).  The oracle must reproduce its outputs and
    its state-dict key set."""
    import numpy as np
    from einops import rearrange
    from oracle import conditioning as OC
    from tests import common_models as CM
    g = np.load(os.path.join(golden_dir, "g5_unet_cmc_omc.npz"))
    W4 = tuple(int(x) for x in g["widths"])
    ou, oe, oa = CM.build_oracle(W4, seed=int(g["seed"]))
    keys = open(os.path.join(golden_dir, "g5_unet_keys.txt")).read().split()
    assert sorted(ou.state_dict().keys()) == keys and len(keys) == int(g["n_keys"])
    clip = CM.synthetic_clip(B=1, Fr=16, H=128, W=128, seed=int(g["clip_seed"]))
    with torch.no_grad():
        plucker = OC.to_plucker_embedding(clip["c2w"], clip["K"], (128, 128))
        pose_emb = rearrange(plucker, "b f c h w -> b c f h w")
        pf = oe(pose_emb)
        assert np.allclose([float(x.double().sum()) for x in pf], g["enc_feat_sums"], rtol=1e-5, atol=1e-4)
        assert torch.allclose(pf[0][:2, :8], torch.from_numpy(g["enc_feat0"]), rtol=1e-5, atol=1e-6)
        assert torch.allclose(pf[3][:2, :8], torch.from_numpy(g["enc_feat3"]), rtol=1e-5, atol=1e-6)
        pose_feats = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in pf]
        traj = OC.get_traj_features(clip["infos"], clip["masks"], oa)
        for t, traj_in, key in ((801, traj, "out"), (801, None, "out_notraj"), (17, traj, "out_t17")):
            out = ou(clip["latents"], torch.tensor([t]), clip["text"], pose_embedding_features=pose_feats,
                     traj_features=traj_in).sample
            ref = torch.from_numpy(g[key])
            assert out.shape == ref.shape
            assert ((out - ref).abs().max() / ref.abs().max()).item() < 1e-5, key
    assert not np.allclose(g["out"], g["out_notraj"])             # the OMC features matter in the reference too


def test_g5_reference_unet_variants(golden_dir):
    """G5: `fmc.models.unet.UNet3DConditionModelPoseCond` (CMC only, un-patched blocks: CameraCtrlPipeline) on the same
    weights equals the CMC+OMC model called without `traj_features`; the processor-less base `UNet3DConditionModel`
    equals an un-conditioned oracle.  Both reference outputs come from make_golden_g5.py."""
    import numpy as np
    from einops import rearrange
    from oracle import conditioning as OC
    from tests import common_models as CM
    g = np.load(os.path.join(golden_dir, "g5_unet_cmc_omc.npz"))
    v = np.load(os.path.join(golden_dir, "g5_unet_variants.npz"))
    W4 = tuple(int(x) for x in g["widths"])
    clip = CM.synthetic_clip(B=1, Fr=16, H=128, W=128, seed=int(g["clip_seed"]))
    assert np.allclose(v["out_posecond"], g["out_notraj"], rtol=1e-6, atol=1e-6)          # reference vs reference
    ob, _, _ = CM.build_oracle(W4, conditioned=False, seed=int(v["base_seed"]))
    with torch.no_grad():
        out = ob(clip["latents"], torch.tensor([801]), clip["text"]).sample
    ref = torch.from_numpy(v["out_base"])
    assert ((out - ref).abs().max() / ref.abs().max()).item() < 1e-5


def test_g5_reference_pipeline_loop(golden_dir):
    """G5: the reference's own `CameraObjCtrlPipeline.__call__` (6 DDIM steps; CFG 2.0 with and without the
    `omcm_min_step` gate, and no CFG) on stub VAE / CLIP, run by make_golden_g5.py.  `oracle.pipeline.denoise` must
    return the same final latents."""
    import numpy as np
    from einops import rearrange
    from oracle import conditioning as OC
    from oracle import diffusers_restated as OD
    from oracle import pipeline as OP
    from tests import common_models as CM
    g = np.load(os.path.join(golden_dir, "g5_unet_cmc_omc.npz"))
    gp = np.load(os.path.join(golden_dir, "g5_pipeline.npz"))
    W4 = tuple(int(x) for x in g["widths"])
    ou, oe, oa = CM.build_oracle(W4, seed=int(g["seed"]))
    clip = CM.synthetic_clip(B=1, Fr=16, H=128, W=128, seed=int(g["clip_seed"]))
    emb_uncond = torch.from_numpy(gp["emb_uncond"])
    with torch.no_grad():
        plucker = OC.to_plucker_embedding(clip["c2w"], clip["K"], (128, 128))
        pose_emb = rearrange(plucker, "b f c h w -> b c f h w")
        traj = OC.get_traj_features(clip["infos"], clip["masks"], oa)
        for key, gs, min_step in (("cfg2_gate700", 2.0, 700), ("cfg2_nogate", 2.0, 0), ("nocfg", 1.0, 0)):
            sched = OD.DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                                     steps_offset=1, clip_sample=False)
            text = torch.cat([emb_uncond, clip["text"]], 0) if gs > 1.0 else clip["text"]
            out = OP.denoise(ou, sched, oe, text, pose_emb, clip["latents"].clone(), num_inference_steps=6,
                             guidance_scale=gs, traj_features=[x.clone() for x in traj], omcm_min_step=min_step)
            ref = torch.from_numpy(gp[key])
            # the stub VAE round trip ((x*0.01/2+0.5) in fp32, then back) costs ~6e-8 * 200 absolute
            assert ((out - ref).abs().max() / ref.abs().max()).item() < 2e-5, key
    assert not np.allclose(gp["cfg2_gate700"], gp["cfg2_nogate"], atol=1e-4)


def test_g5_full_width_oracle_matches_reference_code(golden_dir):
    """The oracle at the BENCHMARKED widths (320/640/1280/1280, text 768, 1.39 B parameters) against the stored output
    of the reference's own U-Net + camera-encoder code (make_golden_g5_full_width.py): same seeded weights and clip."""
    import os
    from einops import rearrange
    from oracle import conditioning as OC
    from tests import common_models as CM
    g = np.load(os.path.join(golden_dir, "g5_unet_full_width.npz"))
    H, W = (int(v) for v in g["hw"])
    ou, oe, oa, clip = CM.full_width_case(int(g["seed"]), int(g["clip_seed"]), H, W)
    assert sum(p.numel() for p in ou.parameters()) == int(g["n_params"])
    with torch.no_grad():
        pose_emb = rearrange(OC.to_plucker_embedding(clip["c2w"], clip["K"], (H, W)), "b f c h w -> b c f h w")
        feats = oe(pose_emb)
        assert np.allclose([float(x.double().sum()) for x in feats], g["enc_feat_sums"], rtol=1e-6)
        pose_feats = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in feats]
        traj = OC.get_traj_features(clip["infos"], clip["masks"], oa)
        out = ou(clip["latents"], torch.tensor([801]), clip["text"], pose_embedding_features=pose_feats,
                 traj_features=traj).sample
    ref = torch.from_numpy(g["out"])
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-5


def test_g7_bf16_format_fixture_is_consistent_with_g7(golden_dir):
    """`g7_bf16_format.npz` (the oracle of BASELINE configs[1] / configs[2] with every layer output and weight rounded to bf16) belongs to the G7 case: same
    seeds / clip / timestep, same shapes, and the stored format errors are the distances of its tensors from G7's reference outputs."""
    import numpy as np
    g7 = np.load(os.path.join(golden_dir, "g7_lora_cam_steps.npz"))
    gb = np.load(os.path.join(golden_dir, "g7_bf16_format.npz"))
    for k in ("lora_seed", "cam_seed", "clip_seed", "uncond_seed", "t"):
        assert int(g7[k]) == int(gb[k])
    assert tuple(g7["hw"]) == tuple(gb["hw"])
    for name in ("lora", "cam"):
        ref, r16 = g7[f"{name}_eps"].astype(np.float64), gb[f"{name}_eps_bf16_rounded_oracle"].astype(np.float64)
        assert ref.shape == r16.shape == (2, 4, 16, 40, 64)
        fmt = np.abs(r16 - ref).max() / np.abs(ref).max()
        assert abs(fmt - float(gb[f"{name}_bf16_format_err"])) < 1e-6 and 5e-3 < fmt < 4e-2
