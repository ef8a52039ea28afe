import os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev="cuda"; bf=torch.bfloat16
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
for (M,N,K,name) in ((46800,12288,4096,"qkv"),(46800,4096,4096,"proj"),(46800,4096,11008,"w2")):
    a=torch.randn(M,K,device=dev,dtype=bf); w=torch.randn(N,K,device=dev,dtype=bf)*0.02; b=torch.randn(N,device=dev,dtype=bf)
    os.environ["LCV_GEMM_TILE"]="9"; ref=ops.gemm_nt(a,w,b)
    os.environ["LCV_GEMM_TILE"]="4"; out=ops.gemm_nt(a,w,b)
    print(name, "equal to 8-phase:", torch.equal(out, ref), "max diff", (out.float()-ref.float()).abs().max().item(), flush=True)
    for tile in ("9","4","9","4"):
        os.environ["LCV_GEMM_TILE"]=tile
        ms=timeit(lambda: ops.gemm_nt(a,w,b), n=5, warm=2)
        print(f"  gemm[{tile}] {name}: {ms:.2f} ms  {2*M*N*K/ms/1e9:.1f} TF/s", flush=True)
    del a,w
