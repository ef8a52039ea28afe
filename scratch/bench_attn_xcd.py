import sys, os, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev="cuda"; bf=torch.bfloat16
for (B,N) in ((1,46800),(2,46800),(1,20280)):
    H=32; D=128
    qkv=torch.randn(B,N,3,H,D,device=dev,dtype=bf); o=torch.empty(B,N,H,D,device=dev,dtype=bf)
    def t(n=3):
        torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
        for _ in range(n): ops.attention(qkv[:,:,0],qkv[:,:,1],qkv[:,:,2],D**-0.5,out=o)
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
    for rnd in range(2):
        for x in ("0","1"):
            os.environ["LCV_ATTN_XCD"]=x
            t(1); ms=t(3); print(f"B={B} N={N} round {rnd} xcd {x}: {ms:.2f} ms {4*B*N*N*H*D/ms/1e9:.0f} TF/s", flush=True)
