"""Host-side logic of the product (module wiring, channels-last plumbing, processor installation, kwargs routing,
pipeline loop) checked against the CPU oracle with the HIP kernels replaced by CPU stand-ins
(`tests/fake_kernels.py`; test infrastructure, never a product fallback).  The same comparisons run for real on
the GPU in tests/test_gpu_model.py."""
import pytest
import torch
from einops import rearrange

from oracle import conditioning as OC
from oracle import diffusers_restated as OD
from oracle import pipeline as OP
from tests import common_models as CM
from tests import fake_kernels

W4 = (32, 64, 64, 64)          # head dims 4/8/8/8: fine for the stand-ins (the real kernels need D % 8 == 0)


def rel_inf(a, b):
    a, b = a.detach().double(), b.detach().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture()
def fake(monkeypatch):
    fake_kernels.install(monkeypatch)


@pytest.fixture(scope="module")
def stack():
    ou, oe, oa = CM.build_oracle(W4, cross_dim=32)
    clip = CM.synthetic_clip(B=1, Fr=16, H=128, W=128, cross_dim=32)
    with torch.no_grad():
        pose_emb = rearrange(OC.to_plucker_embedding(clip["c2w"], clip["K"], (128, 128)), "b f c h w -> b c f h w")
        pose_feats = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in oe(pose_emb)]
        traj = OC.get_traj_features(clip["infos"], clip["masks"], oa)
        t = torch.tensor([801])
        ref = ou(clip["latents"], t, clip["text"], pose_embedding_features=pose_feats, traj_features=traj).sample
        ref0 = ou(clip["latents"], t, clip["text"], pose_embedding_features=pose_feats, traj_features=None).sample
    return dict(ou=ou, oe=oe, oa=oa, clip=clip, pose_emb=pose_emb, pose_feats=pose_feats, traj=traj, t=t, ref=ref,
                ref0=ref0)


def test_product_fails_loudly_without_gpu():
    from synfmc_amd import hip_ops as K
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.groupnorm_silu(torch.zeros(1, 4, 32), torch.ones(32), torch.zeros(32), 32, 1e-5, True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.spatial_attention(torch.zeros(1, 4, 64), torch.zeros(1, 4, 64), torch.zeros(1, 4, 64), 8)


def test_state_dict_keys_match_oracle(stack):
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    assert set(pu.state_dict()) == set(stack["ou"].state_dict())
    assert list(pe.state_dict()) == list(stack["oe"].state_dict())
    assert list(pa.state_dict()) == list(stack["oa"].state_dict())
    assert len(pu.attn_processors) == 32 and len(pu.mm_attn_processors) == 40
    kinds = [type(p).__name__ for p in pu.mm_attn_processors.values()]
    assert kinds.count("PoseAdaptorAttnProcessor") == 20 and kinds.count("AttnProcessor") == 20
    with pytest.raises(ValueError, match="number of processors"):
        pu.set_attn_processor({"x": None})


def test_conditioning_encoders_plumbing(stack, fake):
    from synfmc_amd.data.dataset import to_plucker_embedding
    from synfmc_amd.models.pose_adaptor import features_to_video
    from synfmc_amd.util import get_traj_features_v2
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    clip = stack["clip"]
    emb = to_plucker_embedding(clip["c2w"], clip["K"], (128, 128), device="cpu")
    assert rel_inf(rearrange(emb, "b f c h w -> b c f h w"), stack["pose_emb"]) < 1e-6
    feats = features_to_video(pe(rearrange(emb, "b f c h w -> b c f h w")), 1)
    for got, want in zip(feats, stack["pose_feats"]):
        assert got.shape == want.shape and rel_inf(got, want) < 1e-4
    emb_u = to_plucker_embedding(clip["c2w"], clip["K"], (128, 128), device="cpu", layout="unshuffle8")
    for got, want in zip(features_to_video(pe.forward_unshuffled(emb_u, 1), 1), stack["pose_feats"]):
        assert rel_inf(got, want) < 1e-4
    traj = get_traj_features_v2(clip["infos"], clip["masks"], pa, False, 0.0, [False], "cpu", torch.float32)
    for got, want in zip(traj, stack["traj"]):
        assert got.shape == want.shape and rel_inf(got, want) < 1e-4
    planar = pa(*fake_kernels.omc_rasterize(*_stacked(clip), "planar"))
    for got, want in zip(features_to_video(planar, 1), stack["traj"]):
        assert rel_inf(got, want) < 1e-4


def _stacked(clip):
    from synfmc_amd.util import stack_object_inputs
    return stack_object_inputs(clip["infos"], clip["masks"], "cpu")


def test_unet_plumbing_cmc_omc(stack, fake):
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    clip = stack["clip"]
    with torch.no_grad():
        out = pu(clip["latents"], stack["t"], clip["text"], pose_embedding_features=stack["pose_feats"],
                 traj_features=stack["traj"]).sample
        out0 = pu(clip["latents"], stack["t"], clip["text"], pose_embedding_features=stack["pose_feats"],
                  traj_features=None).sample
    assert out.shape == stack["ref"].shape
    assert rel_inf(out, stack["ref"]) < 1e-3
    assert rel_inf(out0, stack["ref0"]) < 1e-3
    assert rel_inf(stack["ref"], stack["ref0"]) > 1e-3


def test_unpatched_block_rejects_traj_features(stack, fake):
    from synfmc_amd.models.unet import UNet3DConditionModelCamObjCond
    unet = UNet3DConditionModelCamObjCond(**CM.unet_kwargs(W4, 32))
    unet.set_all_attn_processor(**CM.processor_kwargs(W4))
    clip = stack["clip"]
    with pytest.raises(TypeError, match="traj_features"):
        unet(clip["latents"], 1, clip["text"], pose_embedding_features=stack["pose_feats"], traj_features=None)


def test_pose_adaptor_wrappers(stack, fake):
    from synfmc_amd.models.pose_obj_adaptor import CamObjPoseAdaptor
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    clip = stack["clip"]
    with torch.no_grad():
        out = CamObjPoseAdaptor(pu, pe)(clip["latents"], stack["t"], clip["text"], stack["pose_emb"], stack["traj"])
    assert rel_inf(out, stack["ref"]) < 1e-3


def test_unconditioned_base_unet(stack, fake):
    from synfmc_amd.models.unet import UNet3DConditionModel
    ou, _, _ = CM.build_oracle(W4, cross_dim=32, conditioned=False, seed=7)
    base = UNet3DConditionModel(**CM.unet_kwargs(W4, 32))
    base.load_state_dict(ou.state_dict(), strict=True)
    clip = stack["clip"]
    with torch.no_grad():
        assert rel_inf(base.eval()(clip["latents"], 500, clip["text"]).sample,
                       ou(clip["latents"], 500, clip["text"]).sample) < 1e-3


def test_denoising_loop_plumbing(stack, fake):
    from synfmc_amd.pipelines.pipeline_animation_cm_om import CameraObjCtrlPipeline
    from synfmc_amd.schedulers import DDIMScheduler
    kw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
              clip_sample=False)
    clip = stack["clip"]
    text2 = torch.cat([torch.randn(1, 77, 32, generator=torch.Generator().manual_seed(5)), clip["text"]])
    ref = OP.denoise(stack["ou"], OD.DDIMScheduler(**kw), stack["oe"], text2, stack["pose_emb"], clip["latents"],
                     num_inference_steps=4, guidance_scale=2.0, traj_features=stack["traj"], omcm_min_step=700)
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    pipe = CameraObjCtrlPipeline(None, None, None, pu, DDIMScheduler(**kw), pe)
    out = pipe(None, stack["pose_emb"], 16, traj_features=stack["traj"], height=128, width=128, num_inference_steps=4,
               guidance_scale=2.0, latents=clip["latents"], output_type="latent", prompt_embeds=text2,
               omcm_min_step=700, use_graph=False).videos
    assert rel_inf(out, ref) < 5e-3      # CFG multiplies fp32 round-off by ~g*sqrt(2) per step


def test_stage3_training_gradients_plumbing(stack, fake):
    """Adapter gradients through the frozen product U-Net (CPU stand-in kernels are differentiable torch ops)."""
    from tests import training_common as TC
    ou, oe, oa = CM.build_oracle(W4, cross_dim=32, seed=20, fan_in_gain=0.7)     # well-conditioned gradients
    pu, pe, pa = CM.build_product(ou, oe, oa, W4, cross_dim=32, device="cpu")
    clip = stack["clip"]
    noise = torch.randn(clip["latents"].shape, generator=torch.Generator().manual_seed(9))
    t = torch.tensor([801])
    l_ref, g_ref = TC.oracle_grads(ou, oe, oa, clip, stack["pose_emb"], t, noise)
    l_got, g_got = TC.product_grads(pu, pe, pa, clip, stack["pose_emb"], t, noise, "cpu")
    assert abs(float(l_ref) - float(l_got)) < 1e-4 * abs(float(l_ref))
    err, scale = TC.compare(g_ref, g_got)
    assert scale > 0 and err < 2e-3
    # level-3 Adapter blocks receive no gradient (DownBlock3D never consumes traj_features, unet_cam_obj.py:1227-1234)
    assert all(float(g_got[k].abs().max()) == 0.0 for k in g_got if k.startswith("body.6") or k.startswith("body.7")
               or k.startswith("zero_conv_out_list.3"))


LORA_SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                  steps_offset=1, clip_sample=False)


def test_lora_only_animation_pipeline_plumbing(fake):
    """BASELINE configs[1] wiring: `AnimationPipeline` (no pose encoder) over the LoRA-only U-Net, incl. sliding windows."""
    from synfmc_amd.pipelines.pipeline_animation import AnimationPipeline
    from synfmc_amd.schedulers import DDIMScheduler
    ou, pu = CM.build_lora_only(W4, 32, seed=50, fan_in_gain=0.7, device="cpu")
    kinds = {type(p).__name__ for p in pu.attn_processors.values()}, {type(p).__name__ for p in pu.mm_attn_processors.values()}
    assert kinds == ({"LoRAAttnProcessor"}, {"AttnProcessor"})
    g = torch.Generator().manual_seed(3)
    text2 = torch.randn(2, 77, 32, generator=g)
    pipe = AnimationPipeline(None, None, None, pu, DDIMScheduler(**LORA_SCHED))
    for frames, md in ((16, 1), (20, 2)):
        lat = torch.randn(1, 4, frames, 8, 8, generator=g)
        ref = OP.denoise_plain(ou, OD.DDIMScheduler(**LORA_SCHED), text2, lat, 16, 3, 2.0, multidiff_total_steps=md)
        out = pipe(None, 16, height=64, width=64, num_inference_steps=3, guidance_scale=2.0, latents=lat,
                   output_type="latent", prompt_embeds=text2, use_graph=False, multidiff_total_steps=md).videos
        assert out.shape == ref.shape and rel_inf(out, ref) < 2e-3
    with pytest.raises(RuntimeError, match="no VAE"):
        pipe(None, 16, height=64, width=64, num_inference_steps=1, latents=lat[:, :, :16], prompt_embeds=text2,
             use_graph=False)
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe(None, 16, height=60, width=64, prompt_embeds=text2, use_graph=False)


def test_camera_pipeline_rejects_mismatched_conditioning_batch(stack, fake):
    """`num_videos_per_prompt > 1` without repeated conditioning: the reference fails on the shape mismatch, so do we."""
    from synfmc_amd.pipelines.pipeline_animation_cm_om import CameraObjCtrlPipeline
    from synfmc_amd.schedulers import DDIMScheduler
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    pipe = CameraObjCtrlPipeline(None, None, None, pu, DDIMScheduler(**LORA_SCHED), pe)
    text4 = torch.randn(4, 77, 32)
    with pytest.raises(ValueError, match="clips"):
        pipe(None, stack["pose_emb"], 16, traj_features=stack["traj"], height=128, width=128, num_inference_steps=1,
             guidance_scale=2.0, num_videos_per_prompt=2, output_type="latent", prompt_embeds=text4, use_graph=False)


def test_lora_pose_adaptor_processor_plumbing(stack, fake):
    """a11: `LORAPoseAdaptorAttnProcessor` on the temporal attention (`add_motion_lora=True`): merged-LoRA product wiring
    against the oracle's un-merged `W x + s * up(down(x))`."""
    ou, oe, oa = CM.build_oracle(W4, cross_dim=32, seed=60, fan_in_gain=0.7, motion_lora=True)
    pu, pe, pa = CM.build_product(ou, oe, oa, W4, cross_dim=32, device="cpu", motion_lora=True)
    kinds = [type(p).__name__ for p in pu.mm_attn_processors.values()]
    assert kinds.count("LORAPoseAdaptorAttnProcessor") == 20 and kinds.count("LoRAAttnProcessor") == 20
    assert set(pu.state_dict()) == set(ou.state_dict())
    clip = stack["clip"]
    with torch.no_grad():
        pose_feats = [rearrange(x, "(b f) c h w -> b c f h w", b=1) for x in oe(stack["pose_emb"])]
        ref = ou(clip["latents"], stack["t"], clip["text"], pose_embedding_features=pose_feats,
                 traj_features=stack["traj"]).sample
        out = pu(clip["latents"], stack["t"], clip["text"], pose_embedding_features=pose_feats,
                 traj_features=stack["traj"]).sample
    assert rel_inf(out, ref) < 1e-3


def _product_small_adapter(golden_dir, device="cpu"):
    import os
    import numpy as np
    from synfmc_amd.adapter import Adapter
    g2 = np.load(os.path.join(golden_dir, "g2_adapter_small.npz"))
    ad = Adapter(channels=[16, 32, 64, 64], nums_rb=2, cin=832, sk=True, use_conv=False, use_pre_zero_conv=True,
                 use_post_zero_conv=True).eval()
    ad.load_state_dict({k[4:]: torch.from_numpy(g2[k]) for k in g2.files if k.startswith("sd::")}, strict=True)
    return ad.to(device)


def test_get_traj_features_null_condition_matches_reference(golden_dir, fake):
    """`cfg_random_null_om=True` (util.py:194-199): a dropped clip's FEATURES are zeroed, its mask still reaches the Adapter,
    so the result is the mask pyramid times the propagated biases -- against the reference's own output (golden G3b)."""
    import os
    import numpy as np
    from synfmc_amd.util import get_traj_features_v2
    g, gn = np.load(os.path.join(golden_dir, "g3_traj.npz")), np.load(os.path.join(golden_dir, "g3_traj_null.npz"))
    masks = [[torch.from_numpy(g["masks"][b, f]) for f in range(g["masks"].shape[1])] for b in range(g["masks"].shape[0])]
    infos = [[g["infos"][b, f] for f in range(g["infos"].shape[1])] for b in range(g["infos"].shape[0])]
    ad = _product_small_adapter(golden_dir)
    with torch.no_grad():
        kept = get_traj_features_v2(infos, masks, ad, True, 0.0, [False], "cpu", torch.float32)     # ratio 0: never dropped
        null = get_traj_features_v2(infos, masks, ad, True, 1.0, [False], "cpu", torch.float32)     # ratio 1: always dropped
    for i in range(4):
        assert rel_inf(kept[i], torch.from_numpy(g[f"feat_{i}"])) < 1e-4
        assert i > 0 or float(np.abs(gn[f"feat_{i}"]).max()) > 0      # (level 3 is one pixel whose nearest mask sample is 0)
        assert rel_inf(null[i], torch.from_numpy(gn[f"feat_{i}"])) < 1e-4


def test_traj_feature_batch_mismatch_raises(stack, fake):
    """OMC features must cover the whole batch or exactly the conditioned CFG half; anything else raised in the reference
    (`hidden_states + traj_features[idx]`) and must not be silently added to the trailing clips here."""
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    clip = stack["clip"]
    x3 = clip["latents"].repeat(3, 1, 1, 1, 1)
    pf3 = [p.repeat(3, 1, 1, 1, 1) for p in stack["pose_feats"]]
    with pytest.raises(ValueError, match="conditioned CFG half"):
        pu(x3, stack["t"], clip["text"].repeat(3, 1, 1), pose_embedding_features=pf3, traj_features=stack["traj"])
    bad = [t[:, :, :, :-1] for t in stack["traj"]]
    with pytest.raises(ValueError, match="does not match"):
        pu(clip["latents"], stack["t"], clip["text"], pose_embedding_features=stack["pose_feats"], traj_features=bad)


# ---- f4: checkpoint-directory round trip and the VAE / CLIP seam of the pipeline -------------------------------------------
SD15_2D_CONFIG = {   # the keys of SD-1.5's `unet/config.json` (diffusers UNet2DConditionModel), at the test widths
    "_class_name": "UNet2DConditionModel", "_diffusers_version": "0.6.0", "act_fn": "silu", "attention_head_dim": 8,
    "block_out_channels": list(W4), "center_input_sample": False, "cross_attention_dim": 32,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "downsample_padding": 1, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 4, "layers_per_block": 2,
    "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32, "out_channels": 4, "sample_size": 16,
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"]}


def test_from_pretrained_2d_checkpoint_directory_round_trip(tmp_path, stack, fake):
    """`UNet3DConditionModelCamObjCond.from_pretrained_2d(path, subfolder, unet_additional_kwargs)` (unet.py:762-826) on a
    checkpoint directory laid out like SD-1.5's: `unet/config.json` with the 2-D block types + `diffusion_pytorch_model.bin`
    holding the 2-D (non motion-module) weights.  Block types are overridden, every 2-D key loads, exactly the motion-module
    keys stay missing; after loading the motion-module weights on top the model reproduces the oracle."""
    import json
    import os
    from synfmc_amd.models.unet import UNet3DConditionModelCamObjCond
    from synfmc_amd.modified_modules import patch_unet_for_omc
    ou = stack["ou"]
    sd = ou.state_dict()
    sd_2d = {k: v for k, v in sd.items() if "motion_modules." not in k and ".processor." not in k}
    assert 0 < len(sd_2d) < len(sd)
    unet_dir = tmp_path / "sd15" / "unet"
    os.makedirs(unet_dir)
    (unet_dir / "config.json").write_text(json.dumps(SD15_2D_CONFIG))
    torch.save(sd_2d, unet_dir / "diffusion_pytorch_model.bin")
    extra = {k: v for k, v in CM.unet_kwargs(W4, 32).items() if k.startswith("use_motion") or k.startswith("motion_module")}
    unet = UNet3DConditionModelCamObjCond.from_pretrained_2d(str(tmp_path / "sd15"), subfolder="unet",
                                                             unet_additional_kwargs=extra)
    assert unet.config.block_out_channels == tuple(W4) or list(unet.config.block_out_channels) == list(W4)
    got = unet.state_dict()
    for k, v in sd_2d.items():
        assert torch.equal(got[k], v), k
    missing = [k for k in got if k not in sd_2d]
    assert missing and all("motion_modules." in k for k in missing)
    # the second stage of the reference's loading: processors installed, motion-module / adapter checkpoint on top
    unet.set_all_attn_processor(**CM.processor_kwargs(W4))
    patch_unet_for_omc(unet)
    unet.load_state_dict({k: v for k, v in sd.items() if k not in sd_2d}, strict=False)
    clip = stack["clip"]
    with torch.no_grad():
        out = unet.eval()(clip["latents"], stack["t"], clip["text"], pose_embedding_features=stack["pose_feats"],
                          traj_features=stack["traj"]).sample
    assert rel_inf(out, stack["ref"]) < 1e-3
    with pytest.raises(RuntimeError, match="does not exist"):
        UNet3DConditionModelCamObjCond.from_pretrained_2d(str(tmp_path / "nowhere"), subfolder="unet")


def test_pipeline_with_tokenizer_text_encoder_and_vae(stack, fake):
    """The seam either side of the loop (pipeline_animation_cm_om.py:465-560): prompt -> tokenizer -> text encoder (cond and
    the empty negative prompt for the unconditional half), final latents -> `1/0.18215` -> per-frame `vae.decode` ->
    `(x/2 + 0.5).clamp(0, 1)` video `[B, C, F, H, W]`, through duck-typed stand-ins (CLIP / the VAE themselves are third-party
    models outside the hot path)."""
    import types
    from synfmc_amd.pipelines.pipeline_animation_cm_om import CameraObjCtrlPipeline
    from synfmc_amd.schedulers import DDIMScheduler
    clip = stack["clip"]
    uncond = torch.randn(1, 77, 32, generator=torch.Generator().manual_seed(5))
    seen = []

    class Tok:
        model_max_length = 77

        def __call__(self, texts, **kw):
            seen.append(list(texts))
            return types.SimpleNamespace(input_ids=torch.tensor([[1 if t else 0] * 77 for t in texts]))

    class TextEnc:
        def __call__(self, ids):
            return (torch.cat([clip["text"] if int(r[0]) else uncond for r in ids], 0),)

    class Vae:
        def decode(self, z):
            assert z.shape[0] == 1 and z.ndim == 4            # one frame at a time, like the reference (:471-473)
            return types.SimpleNamespace(sample=z[:, :3] * 0.18215 * 0.01)

    kw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
              clip_sample=False)
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    pipe = CameraObjCtrlPipeline(Vae(), TextEnc(), Tok(), pu, DDIMScheduler(**kw), pe)
    common = dict(traj_features=stack["traj"], height=128, width=128, num_inference_steps=2, guidance_scale=2.0,
                  latents=clip["latents"], use_graph=False)
    video = pipe("a prompt", stack["pose_emb"], 16, **common).videos
    assert seen == [["a prompt"], [""]]
    lat = pipe(None, stack["pose_emb"], 16, output_type="latent", prompt_embeds=torch.cat([uncond, clip["text"]]), **common).videos
    assert tuple(video.shape) == (1, 3, 16, 16, 16) and video.dtype == torch.float32
    want = (lat[:, :3] * 0.01 / 2 + 0.5).clamp(0, 1)
    assert rel_inf(video, want) < 1e-5


def test_ragged_and_empty_object_lists(stack, fake):
    """Frames with different numbers of objects (and frames with none): `stack_object_inputs` pads with empty masks that never
    win the rasteriser; the result equals the oracle's per-frame loops (fmc/util.py:158-201 iterate over whatever each frame holds)."""
    from synfmc_amd.util import get_traj_features_v2
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    clip = stack["clip"]
    infos = [[fi[: (f % 4)] for f, fi in enumerate(bi)] for bi in clip["infos"]]          # 0, 1, 2, 3, 0, 1, ... objects per frame
    masks = [[fm[: (f % 4)] for f, fm in enumerate(bm)] for bm in clip["masks"]]
    assert masks[0][0].shape[0] == 0 and masks[0][3].shape[0] == 3
    with torch.no_grad():
        want = OC.get_traj_features(infos, masks, stack["oa"])
        got = get_traj_features_v2(infos, masks, pa, False, 0.0, [False], "cpu", torch.float32)
    for g, w in zip(got, want):
        assert g.shape == w.shape and rel_inf(g, w) < 1e-4
    assert float(got[0][:, :, 0].abs().max()) == 0.0           # zero mask -> zero features on an object-free frame
    assert float(got[0][:, :, 3].abs().max()) > 0.0


def test_pipeline_without_classifier_free_guidance(stack, fake):
    """guidance_scale <= 1: batch is not doubled, no unconditional text, OMC features cover the whole batch (reference
    pipeline_animation_cm_om.py:662-676 with `do_classifier_free_guidance == False`)."""
    from synfmc_amd.pipelines.pipeline_animation_cm_om import CameraObjCtrlPipeline
    from synfmc_amd.schedulers import DDIMScheduler
    kw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1,
              clip_sample=False)
    clip = stack["clip"]
    ref = OP.denoise(stack["ou"], OD.DDIMScheduler(**kw), stack["oe"], clip["text"], stack["pose_emb"], clip["latents"],
                     num_inference_steps=3, guidance_scale=1.0, traj_features=stack["traj"])
    pu, pe, pa = CM.build_product(stack["ou"], stack["oe"], stack["oa"], W4, cross_dim=32, device="cpu")
    pipe = CameraObjCtrlPipeline(None, None, None, pu, DDIMScheduler(**kw), pe)
    out = pipe(None, stack["pose_emb"], 16, traj_features=stack["traj"], height=128, width=128, num_inference_steps=3,
               guidance_scale=1.0, latents=clip["latents"], output_type="latent", prompt_embeds=clip["text"],
               use_graph=False).videos
    assert rel_inf(out, ref) < 2e-3


def test_vae_and_clip_state_dict_keys():
    """f4: `synfmc_amd.models.clip_text.CLIPTextModel` carries exactly `transformers.CLIPTextModel`'s keys (and accepts the `text_model.`-prefixed
    form of older checkpoints); `synfmc_amd.models.vae.AutoencoderKL` carries the keys of diffusers' AutoencoderKL as the restated
    oracle lists them (encoder, quant_conv, post_quant_conv, decoder); `load_decoder_state_dict` reads only the decoder half."""
    import transformers
    from oracle import vae_restated as OV
    from synfmc_amd.models.clip_text import CLIPTextConfig, CLIPTextModel
    from synfmc_amd.models.vae import AutoencoderKL
    kw = dict(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, max_position_embeddings=77)
    with torch.device("meta"):
        ref = transformers.CLIPTextModel(transformers.CLIPTextConfig(**kw, bos_token_id=98, eos_token_id=99, pad_token_id=0))
        mine = CLIPTextModel(CLIPTextConfig(**kw))
    strip = lambda k: k[len("text_model."):] if k.startswith("text_model.") else k
    want = {strip(k): tuple(v.shape) for k, v in ref.state_dict().items() if not k.endswith("position_ids")}
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == want
    real = CLIPTextModel(CLIPTextConfig(**kw))
    real.load_state_dict({"text_model." + k: v for k, v in real.state_dict().items()}, strict=True)      # transformers-4.x style keys
    widths = (64, 64, 128, 128)
    ov = OV.AutoencoderKLFull(widths)
    vae = AutoencoderKL(block_out_channels=widths)
    assert {k: tuple(v.shape) for k, v in vae.state_dict().items()} == {k: tuple(v.shape) for k, v in ov.state_dict().items()}
    vae.load_state_dict(ov.state_dict(), strict=True)                       # a full checkpoint loads key for key
    dec_only = {k: v for k, v in OV.AutoencoderKLDecoderOnly(widths).state_dict().items()}
    assert all(k.startswith(("decoder.", "post_quant_conv.")) for k in dec_only)
    vae.load_decoder_state_dict(dec_only, strict=True)                      # a decoder-only checkpoint leaves the encoder half as it is
    full = dict(dec_only)
    full["encoder.conv_in.weight"] = torch.zeros(1)                         # encoder keys of a full checkpoint are ignored by this loader
    full["quant_conv.weight"] = torch.zeros(1)
    vae.load_decoder_state_dict(full, strict=True)
    for k in ("encoder.down_blocks.0.downsamplers.0.conv.weight", "encoder.mid_block.attentions.0.to_q.weight", "quant_conv.bias"):
        assert k in vae.state_dict()
    for k in ("decoder.mid_block.attentions.0.to_q.weight", "decoder.up_blocks.2.resnets.0.conv_shortcut.weight",
              "decoder.up_blocks.0.upsamplers.0.conv.weight", "decoder.conv_norm_out.weight", "post_quant_conv.bias"):
        assert k in vae.state_dict()
