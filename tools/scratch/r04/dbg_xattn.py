import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch, torch.nn.functional as F
from synfmc_amd import hip_ops as K
torch.manual_seed(0)
B,Fr,hw,C,S,H,d=1,1,80,640,77,8,80
dt=torch.bfloat16
h=(torch.randn(B*Fr,hw,C)*1.0).to(dt).cuda()
g=torch.ones(C).cuda(); bt=torch.zeros(16,C).cuda()
wq=(torch.randn(C,C)*C**-0.5).to(dt).cuda(); wo=torch.eye(C).to(dt).cuda()
kv=(torch.randn(B,S,2*C)).to(dt).cuda()
out=K.xattn_block640(h,g,bt,1e-5,K.pack_w_frag80(wq),kv,K.pack_w_frag80(wo),None,d**-0.5,Fr)
o=(out.float()-h.float()).cpu()
x=F.layer_norm(h.float().cpu(),(C,)).bfloat16().float()
q=F.linear(x,wq.float().cpu()).bfloat16().float().reshape(B,Fr*hw,H,d).permute(0,2,1,3)
k=kv.float().cpu()[...,:C].reshape(B,S,H,d).permute(0,2,1,3); v=kv.float().cpu()[...,C:].reshape(B,S,H,d).permute(0,2,1,3)
sc=q@k.transpose(-1,-2)*d**-0.5
p=torch.softmax(sc,-1)
oref=(p@v).permute(0,2,1,3).reshape(B*Fr,hw,C)
err=(o-oref).abs()
print("max err",err.max().item(),"ref max",oref.abs().max().item())
e=err.view(hw,H,d)
print("per head max:",[round(e[:,hh].max().item(),3) for hh in range(H)])
print("per channel-block (head 0):",[round(e[:,0,16*b:16*b+16].max().item(),3) for b in range(5)])
print("per row-block:",[round(e[16*m:16*m+16].max().item(),3) for m in range(5)])
# alternatives: uniform attention? no-tail? 
p2=torch.softmax(sc[..., :64],-1)
for name,alt in [("first 64 keys only",(p2@v[:,:,:64]).permute(0,2,1,3).reshape(B*Fr,hw,C)),
                 ("no d-tail in scores",(torch.softmax((q[...,:64]@k[...,:64].transpose(-1,-2))*d**-0.5,-1)@v).permute(0,2,1,3).reshape(B*Fr,hw,C)),
                 ("mean of v",v.mean(2,keepdim=True).expand(-1,-1,hw,-1).permute(0,2,1,3).reshape(B*Fr,hw,C))]:
    print(name,(o-alt).abs().max().item())
