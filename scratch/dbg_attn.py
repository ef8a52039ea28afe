import sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
D=128; dev="cuda"; bf=torch.bfloat16
def run(q,k,v):
    o,_=ops.attention(q.to(dev),k.to(dev),v.to(dev),1.0); return o.float().cpu()
Nq,Nk=32,64
# A: Q=0 -> uniform P; V[key,d]=d  -> O[q,d]=d
q=torch.zeros(1,Nq,1,D,dtype=bf); k=torch.zeros(1,Nk,1,D,dtype=bf)
v=torch.arange(D).float().view(1,1,1,D).expand(1,Nk,1,D).contiguous().to(bf)
o=run(q,k,v); print("A: O[0,:]", o[0,0,0,:].tolist()); print("A rows equal:", (o[0,:,0]==o[0,0,0]).all().item())
# A2: V[key,d]=key -> O = mean(key)=31.5
v=torch.arange(Nk).float().view(1,Nk,1,1).expand(1,Nk,1,D).contiguous().to(bf)
o=run(q,k,v); print("A2: O[0,:8]", o[0,0,0,:8].tolist())
# B: one-hot attention: Q[i]=30*e_i, K[key]=e_key ; V[key,d]=key
q=torch.zeros(1,Nq,1,D); 
for i in range(Nq): q[0,i,0,i]=30.0
k=torch.zeros(1,Nk,1,D)
for j in range(Nk): k[0,j,0,j]=1.0
o=run(q.to(bf),k.to(bf),v); print("B: attended key per q:", o[0,:,0,0].tolist())
# B2: shift: Q[i]=30*e_{i+32}
q=torch.zeros(1,Nq,1,D)
for i in range(Nq): q[0,i,0,i+32]=30.0
o=run(q.to(bf),k.to(bf),v); print("B2: attended key per q (expect i+32):", o[0,:,0,0].tolist())
# C: V[key,d] = key*128+d small: one-hot -> O[i,d] = V[i,d]
v=(torch.arange(Nk).view(Nk,1)*1.0+torch.arange(D).view(1,D)/256.0).view(1,Nk,1,D).to(bf)
q=torch.zeros(1,Nq,1,D)
for i in range(Nq): q[0,i,0,i]=30.0
o=run(q.to(bf),k.to(bf),v); print("C row5:", o[0,5,0,:16].tolist(), " expect", v[0,5,0,:16].float().tolist())
