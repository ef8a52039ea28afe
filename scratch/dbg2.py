import sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
torch.set_printoptions(linewidth=200, precision=4, sci_mode=False)
D=128; dev="cuda"; bf=torch.bfloat16
def run(q,k,v):
    o,_=ops.attention(q.to(dev),k.to(dev),v.to(dev),1.0); return o.float().cpu()
Nq,Nk=32,64
q=torch.zeros(1,Nq,1,D,dtype=bf); k=torch.zeros(1,Nk,1,D,dtype=bf)
res=[]
for k0 in range(Nk):
    v=torch.zeros(1,Nk,1,D); v[0,k0]=1.0
    o=run(q,k,v.to(bf))
    res.append(round(o[0,0,0,0].item()*64,2))
print("D: 64*O for one-hot V row k0:", res)
# also per-d check for k0=5
v=torch.zeros(1,Nk,1,D); v[0,5]=1.0
o=run(q,k,v.to(bf)); print("k0=5 per d *64:", (o[0,0,0,:]*64).tolist()[:40])
print("k0=5 per q row, d=0 *64:", (o[0,:,0,0]*64).tolist())
