import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from synfmc_amd import hip_ops as K, _lib
B,S,C=2,77,640
kv=torch.randn(B,S,2*C,device="cuda",dtype=torch.bfloat16)
frag=torch.empty(B*8*12800,dtype=torch.bfloat16,device="cuda")
_lib.check(_lib.load().fmc_xattn_pack_kv(kv.data_ptr(),frag.data_ptr(),B,S,kv.stride(0),K._stream()),"pack")
torch.cuda.synchronize()
f=frag.view(B,8,12800).float().cpu(); kvc=kv.float().cpu()
kp=torch.zeros(B,80,2*C); kp[:,:S]=kvc
bad=0
for b in range(B):
  for h in range(8):
    k=kp[b,:,h*80:(h+1)*80]; v=kp[b,:,640+h*80:640+(h+1)*80]
    for kb in range(5):
      for lane in range(64):
        l15,kq=lane&15,lane>>4
        key=16*kb+l15
        exp_a=torch.cat([k[key,4*kq:4*kq+4],k[key,16+4*kq:16+4*kq+4]])
        exp_b=torch.cat([k[key,32+4*kq:32+4*kq+4],k[key,48+4*kq:48+4*kq+4]])
        exp_t=k[key,64+4*kq:64+4*kq+4]
        base=kb*1280
        if not torch.equal(f[b,h,base+lane*8:base+lane*8+8],exp_a): bad+=1
        if not torch.equal(f[b,h,base+512+lane*8:base+512+lane*8+8],exp_b): bad+=1
        if not torch.equal(f[b,h,base+1024+lane*4:base+1024+lane*4+4],exp_t): bad+=1
    for cb in range(5):
      for kb in range(5):
        for lane in range(64):
          l15,kq=lane&15,lane>>4
          exp=v[16*kb+4*kq:16*kb+4*kq+4,16*cb+l15]
          o=6400+((cb*5+kb)*64+lane)*4
          if not torch.equal(f[b,h,o:o+4],exp): bad+=1
print("bad fragments:",bad)
