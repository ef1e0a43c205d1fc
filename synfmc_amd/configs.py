"""Constructor arguments of the shipped FMC configurations (`configs/obj.yaml`, `cam.yaml`, `lora.yaml` of the reference: U-Net /
processor layout / camera encoder / OMC adapter keyword arguments, as `train_cam_obj_ctrl.py:231-329` passes them) at any width ladder, and
the synthetic workload of SURVEY.md section 8d (seeded latents, text, smooth camera trajectory, per-object poses and Gaussian circle masks).
Used by `bench.py`, `__graft_entry__.smoke()` and the tests (`tests/common_models.py` re-exports these and adds the oracle-side builders)."""
import copy

import numpy as np
import torch

# SD-1.5 `unet/config.json` as `UNet3DConditionModel.from_pretrained_2d` reads it (fmc/models/unet.py:762-826) with the 3-D block names
SD15_UNET_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
    mid_block_type="UNetMidBlock3DCrossAttn",
    up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
    mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768,
    attention_head_dim=8,
)

MMK = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
           temporal_position_encoding=True, temporal_position_encoding_max_len=32, temporal_attention_dim_div=1,
           zero_initialize=False)


def unet_kwargs(widths=(64, 128, 256, 256), cross_dim=64, motion=True):
    cfg = dict(SD15_UNET_CONFIG)
    cfg.update(block_out_channels=tuple(widths), cross_attention_dim=cross_dim, sample_size=16)
    cfg.update(use_motion_module=motion, motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False,
               motion_module_type="Vanilla", motion_module_kwargs=copy.deepcopy(MMK))
    return cfg


def processor_kwargs(widths, lora=True, temporal=True, motion_lora=False):
    """configs/obj.yaml / cam.yaml processor layout.  `temporal=False`: the Domain-LoRA-only model of BASELINE
    configs[1] (LoRA on attn1 / attn2, plain temporal attention).  `motion_lora=True`: `LORAPoseAdaptorAttnProcessor`
    (Camera Adapter + LoRA, rank C/4) on the temporal attention."""
    return dict(add_spatial=False, spatial_attn_names="attn1", add_temporal=temporal, temporal_attn_names="0",
                add_spatial_lora=lora, add_motion_lora=motion_lora,
                lora_kwargs={"lora_rank": 2, "lora_scale": 1.0},
                motion_lora_kwargs={"lora_rank": 4 if motion_lora else -1, "lora_scale": 1.0},
                pose_feature_dimensions=list(widths), query_condition=True, key_value_condition=True, scale=1.0)


def encoder_kwargs(widths, max_len=16):
    return dict(downscale_factor=8, channels=list(widths), nums_rb=2, cin=384, ksize=1, sk=True, use_conv=False,
                compression_factor=1, temporal_attention_nhead=8, attention_block_types=["Temporal_Self"],
                temporal_position_encoding=True, temporal_position_encoding_max_len=max_len)


def adapter_kwargs(widths):
    return dict(channels=list(widths), nums_rb=2, cin=832, sk=True, use_conv=False, use_pre_zero_conv=True,
                use_post_zero_conv=True)



def gaussian_circle_mask(H: int, W: int, center, radius: float) -> np.ndarray:
    """Analytic part of the reference's sphere mask (fmc/data/dataset.py:5365-5380) on the host, for the synthetic clips: a filled disc of
    integer centre / radius times a Gaussian of sigma = radius / 2 around the float centre, normalised by its maximum over the image.
    (The device form is `fmc_gaussian_circle_mask_fwd`, `synfmc_amd.data.dataset.gaussian_circle_masks`.)"""
    yy, xx = np.ogrid[:H, :W]
    dist = np.sqrt((xx - center[0]) ** 2 + (yy - center[1]) ** 2)
    g = np.exp(-0.5 * (dist / (radius / 2.0)) ** 2)
    g = g / g.max()
    disc = ((xx - int(center[0])) ** 2 + (yy - int(center[1])) ** 2) <= int(radius) ** 2
    return (disc * g).astype(np.float64)


def synthetic_clip(B=1, Fr=16, H=128, W=128, n_obj=3, cross_dim=64, seed=100):
    """SURVEY.md section 8d synthetic inputs: latents, text, smooth relative camera trajectory, intrinsics,
    per-object relative poses and Gaussian circle masks."""
    g = torch.Generator().manual_seed(seed)
    rng = np.random.default_rng(seed)
    latents = torch.randn(B, 4, Fr, H // 8, W // 8, generator=g)
    text = torch.randn(B, 77, cross_dim, generator=g)
    c2w = np.zeros((B, Fr, 3, 4), dtype=np.float32)
    for b in range(B):
        ang, t = np.zeros(3), np.zeros(3)
        for f in range(Fr):
            if f:
                ang += rng.normal(0, 0.02, 3)
                t += rng.normal(0, 0.03, 3)
            cx, cy, cz = np.cos(ang)
            sx, sy, sz = np.sin(ang)
            Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
            Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
            c2w[b, f, :, :3] = Rz @ Ry @ Rx
            c2w[b, f, :, 3] = t
    K = torch.tensor([float(W), float(W), W / 2.0, H / 2.0]).view(1, 1, 4).repeat(B, Fr, 1)
    infos, masks = [], []
    for b in range(B):
        ctr = rng.uniform([W * 0.25, H * 0.25], [W * 0.75, H * 0.75], size=(n_obj, 2))
        rad = rng.uniform(min(H, W) * 0.12, min(H, W) * 0.3, size=n_obj)
        fi, fm = [], []
        for f in range(Fr):
            ctr = ctr + rng.normal(0, 1.5, size=ctr.shape)
            fm.append(torch.from_numpy(np.stack(
                [gaussian_circle_mask(H, W, ctr[o], rad[o])[None] for o in range(n_obj)])).float())
            fi.append(rng.normal(0, 0.5, size=(n_obj, 12)))
        infos.append(fi)
        masks.append(fm)
    return dict(latents=latents, text=text, c2w=torch.from_numpy(c2w), K=K, infos=infos, masks=masks)



def union_masks(clip, thr=0.5):
    """`[B, F, H, W]` boolean union of the objects' masks (the loss mask of train_cam_obj_ctrl.py:880-899)."""
    return torch.stack([torch.stack([(m[:, 0] > thr).any(dim=0) for m in clip["masks"][b]]) for b in range(len(clip["masks"]))])
