import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from einops import rearrange
from oracle import conditioning as OC
from tests import common_models as CM
from synfmc_amd.pipelines.pipeline_animation_cm_om import CameraObjCtrlPipeline
from synfmc_amd.schedulers import DDIMScheduler
W4 = (64, 128, 256, 256)
kw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
ou, oe, oa = CM.build_oracle(W4)
pu, pe, pa = CM.build_product(ou, oe, oa, W4, dtype=torch.bfloat16)
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
clips = {"A": CM.synthetic_clip(B=1, Fr=16, H=128, W=128), "B": CM.synthetic_clip(B=1, Fr=16, H=128, W=128, seed=777)}
prep = {}
for k, clip in clips.items():
    g = torch.Generator().manual_seed(5 if k == "A" else 6)
    text2 = torch.cat([torch.randn(1, 77, 64, generator=g), clip["text"]])
    with torch.no_grad():
        pose_emb = rearrange(OC.to_plucker_embedding(clip["c2w"], clip["K"], (128, 128)), "b f c h w -> b c f h w")
        traj = OC.get_traj_features(clip["infos"], clip["masks"], oa)
    prep[k] = (text2.cuda(), pose_emb.cuda().bfloat16(), [t.cuda() for t in traj], clip["latents"].cuda())
for use_graph in (False, True):
    pipe = CameraObjCtrlPipeline(None, None, None, pu, DDIMScheduler(**kw), pe)
    outs = []
    seq = "AABBAB"
    for k in seq:
        text2, pose_emb, traj, lat = prep[k]
        outs.append(pipe(None, pose_emb, 16, traj_features=traj, height=128, width=128, num_inference_steps=3, guidance_scale=2.0,
                         latents=lat, output_type="latent", prompt_embeds=text2, use_graph=use_graph).videos.clone())
    print("graph" if use_graph else "eager", "A0-A1", rel(outs[0], outs[1]), "A1-A4", rel(outs[1], outs[4]), "B2-B3", rel(outs[2], outs[3]),
          "B3-B5", rel(outs[3], outs[5]), "A-B", rel(outs[0], outs[2]))
    # encoder determinism
    f1 = pe(prep["A"][1]); f2 = pe(prep["A"][1])
    print("  encoder repeat", max(rel(a, b) for a, b in zip(f1, f2)))
    x = torch.cat([prep["A"][3]] * 2).bfloat16()
