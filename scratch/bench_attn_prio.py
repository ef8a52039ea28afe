import sys, os, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev="cuda"; bf=torch.bfloat16
N=46800; H=32; D=128
qkv=torch.randn(1,N,3,H,D,device=dev,dtype=bf); o=torch.empty(1,N,H,D,device=dev,dtype=bf)
def t(n=3):
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): ops.attention(qkv[:,:,0],qkv[:,:,1],qkv[:,:,2],D**-0.5,out=o)
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for rnd in range(3):
    for prio in ("0","1","2"):
        os.environ["LCV_ATTN_PRIO"]=prio
        t(1); ms=t(3); print(f"round {rnd} prio {prio}: {ms:.2f} ms {4*N*N*H*D/ms/1e9:.0f} TF/s", flush=True)
