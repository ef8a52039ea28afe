import sys, os, torch, time
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
from lcv_hip.lib import LCV_EPI_SWIGLU
dev="cuda"; bf=torch.bfloat16
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
which = sys.argv[1] if len(sys.argv)>1 else "all"
if which in ("all","attn"):
    for N in (12480, 46800):
        H=32; D=128
        qkv=torch.randn(1,N,3,H,D,device=dev,dtype=bf)
        o=torch.empty(1,N,H,D,device=dev,dtype=bf)
        ms=timeit(lambda: ops.attention(qkv[:,:,0],qkv[:,:,1],qkv[:,:,2],D**-0.5,out=o), n=3, warm=1)
        fl=4*N*N*H*D
        print(f"attn N={N}: {ms:.2f} ms  {fl/ms/1e9:.1f} TF/s", flush=True)
if which in ("attn_unit",):
    import math
    for N in (46800,):
        H=32; D=128
        qkv=torch.randn(1,N,3,H,D,device=dev,dtype=torch.float32)
        c=D**-0.5*math.log2(math.e)
        q_pre=(qkv[:,:,0]*c).to(bf).contiguous(); qkv=qkv.to(bf)
        o=torch.empty(1,N,H,D,device=dev,dtype=bf); o2=torch.empty_like(o)
        ms=timeit(lambda: ops.attention(qkv[:,:,0],qkv[:,:,1],qkv[:,:,2],D**-0.5,out=o), n=3, warm=1)
        fl=4*N*N*H*D
        print(f"attn general N={N}: {ms:.2f} ms  {fl/ms/1e9:.1f} TF/s", flush=True)
        ms=timeit(lambda: ops.attention(q_pre,qkv[:,:,1],qkv[:,:,2],math.log(2.0),out=o2), n=3, warm=1)
        print(f"attn unit    N={N}: {ms:.2f} ms  {fl/ms/1e9:.1f} TF/s", flush=True)
        d=(o.float()-o2.float()); print("rel_l2 unit vs general:", (d.norm()/o.float().norm()).item(), "max abs", d.abs().max().item())
if which in ("attn_bwd",):
    import math
    N=25200; H=32; D=128
    qkv=torch.randn(1,N,3,H,D,device=dev,dtype=torch.float32)
    c=D**-0.5*math.log2(math.e)
    q=(qkv[:,:,0]*c).to(bf).contiguous(); k=qkv[:,:,1].to(bf).contiguous(); v=qkv[:,:,2].to(bf).contiguous()
    o,lse=ops.attention(q,k,v,math.log(2.0),need_lse=True)
    do=torch.randn(1,N,H,D,device=dev,dtype=bf)
    dq=torch.empty_like(q); dk=torch.empty_like(k); dv=torch.empty_like(v)
    outs={}
    for var in ("0","1","2","3","0","3"):
        os.environ["LCV_ATTN_BWD_VAR"]=var
        ms=timeit(lambda: ops.attention_bwd(q,k,v,o,do,lse,dq,dk,dv,math.log(2.0)), n=3, warm=1)
        outs[var]=dq.clone()
        print(f"attn_bwd var={var} N={N}: {ms:.2f} ms  (fwd-equivalent {10*N*N*H*D/ms/1e9:.1f} TF/s algorithmic)", flush=True)
    d=(outs["0"].float()-outs["3"].float()); print("dq rel_l2 new vs old:", (d.norm()/outs["0"].float().norm()).item())
if which in ("gemm_group",):
    for (M,N,K,name) in ((46800,12288,4096,"qkv"),(46800,4096,4096,"proj"),(46800,22016,4096,"w13"),(46800,4096,11008,"w2")):
        a=torch.randn(M,K,device=dev,dtype=bf); w=torch.randn(N,K,device=dev,dtype=bf)*0.02; b=torch.randn(N,device=dev,dtype=bf)
        for gm in ("4","8","4","8","6","8","16"):
            os.environ["LCV_GEMM_GROUP_M"]=gm
            ms=timeit(lambda: ops.gemm_nt(a,w,b), n=5, warm=2)
            print(f"gemm group_m={gm} {name}: {ms:.2f} ms  {2*M*N*K/ms/1e9:.1f} TF/s", flush=True)
        del a,w
if which in ("all","gemm"):
    for (M,N,K,name) in ((46800,12288,4096,"qkv"),(46800,4096,4096,"proj"),(46800,22016,4096,"w13"),(46800,4096,11008,"w2")):
        a=torch.randn(M,K,device=dev,dtype=bf); w=torch.randn(N,K,device=dev,dtype=bf)*0.02; b=torch.randn(N,device=dev,dtype=bf)
        for tile in ("8","9"):
            os.environ["LCV_GEMM_TILE"]=tile
            ms=timeit(lambda: ops.gemm_nt(a,w,b), n=5, warm=2)
            print(f"gemm[{tile}] {name} M={M} N={N} K={K}: {ms:.2f} ms  {2*M*N*K/ms/1e9:.1f} TF/s", flush=True)
        ms=timeit(lambda: torch.nn.functional.linear(a,w,b), n=5, warm=2)
        print(f"   torch(hipblaslt) : {ms:.2f} ms  {2*M*N*K/ms/1e9:.1f} TF/s", flush=True)
        del a,w
if which in ("all","elem"):
    N=46800; C=4096; T=13
    x=torch.randn(1,N,C,device=dev,dtype=bf); mod=torch.randn(1,T,6*C,device=dev)
    ms=timeit(lambda: ops.adaln_modulate(x,mod,0,1,T)); print(f"adaln: {ms:.3f} ms {2*N*C*2/ms/1e6:.0f} GB/s")
    y=torch.randn_like(x)
    ms=timeit(lambda: ops.gate_residual(x,y,mod,2,T)); print(f"gate_residual: {ms:.3f} ms {3*N*C*2/ms/1e6:.0f} GB/s")
    qkv=torch.randn(1,N,3,32,128,device=dev,dtype=bf); w=torch.ones(128,device=dev,dtype=bf); cs=torch.randn(N,64,2,device=dev)
    ms=timeit(lambda: ops.qknorm_rope(qkv[:,:,0],qkv[:,:,1],None,qkv[:,:,0],qkv[:,:,1],None,w,w,cs)); print(f"qknorm_rope: {ms:.3f} ms {4*N*C*2/ms/1e6:.0f} GB/s")
if which in ("eval",):
    # on-device evaluation kernels at the frame sizes of the two resolutions (HBM-bound: 4 B gen + 1 B uint8 gt per element)
    import ctypes
    from lcv_hip.lib import call
    for (N, H, W) in ((14, 480, 832), (36, 720, 1280)):
        gen = torch.rand(N, H, W, 3, device=dev); gt = torch.randint(0, 256, (N, H, W, 3), device=dev, dtype=torch.uint8)
        for win in (11, 7):
            a, b = ctypes.c_int64(0), ctypes.c_int64(0)
            call("lcv_frame_metric_partials", H, W, 3, win, ctypes.byref(a), ctypes.byref(b))
            p1 = torch.empty(N, a.value, device=dev); p2 = torch.empty(N, b.value, device=dev)
            taps = ops.gaussian_window11() if win == 11 else __import__("numpy").full(7, 1 / 7, dtype="float32")
            st = torch.cuda.current_stream().cuda_stream
            f1 = lambda: call("lcv_frame_sqerr", gen.data_ptr(), gt.data_ptr(), 1, p1.data_ptr(), N, H * W * 3, st)
            f2 = lambda: call("lcv_frame_ssim", gen.data_ptr(), gt.data_ptr(), 1, p2.data_ptr(), N, H, W, 3, taps.ctypes.data, win,
                              1.0, 1, 1e-4, 9e-4, st)
            by = N * H * W * 3 * 5
            m1, m2 = timeit(f1, n=20, warm=3), timeit(f2, n=20, warm=3)
            print(f"eval {N}x{H}x{W} win {win}: sqerr {m1*1e3:.1f} us {by/m1/1e6:.0f} GB/s | ssim {m2*1e3:.1f} us {by/m2/1e6:.0f} GB/s", flush=True)
        full = lambda: ops.frame_metrics(gen, gt)
        print(f"  frame_metrics end to end (2 launches + partial sums + D2H): {timeit(full, n=10, warm=2)*1e3:.1f} us", flush=True)
if which in ("gemm_tail",):
    # the reference's 480p operating shapes (generation: M = 2 x 6 240 rows; TTA: M = 6 240): thin last rounds on 256 CUs,
    # with and without the split-K tail, interleaved in one process
    for (M,N,K,name) in ((12480,4096,4096,"proj@12480"),(12480,12288,4096,"qkv@12480"),(12480,4096,11008,"w2@12480"),
                         (6240,4096,4096,"proj@6240"),(6240,12288,4096,"qkv@6240"),(6240,11008,4096,"w1@6240"),(6240,4096,11008,"w2@6240")):
        a=torch.randn(M,K,device=dev,dtype=bf); w=torch.randn(N,K,device=dev,dtype=bf)*0.02; b=torch.randn(N,device=dev,dtype=bf)
        res={}
        for rep in range(3):
            for split in ("0","1"):
                os.environ["LCV_GEMM_SPLITK_TAIL"]=split
                res.setdefault(split,[]).append(timeit(lambda: ops.gemm_nt(a,w,b), n=10, warm=2))
        m0,m1=min(res["0"]),min(res["1"])
        tiles=((M+255)//256)*((N+255)//256)
        print(f"{name}: {tiles} tiles ({tiles/256:.2f} rounds)  unsplit {m0*1e3:.0f} us {2*M*N*K/m0/1e9:.0f} TF/s | split-K tail {m1*1e3:.0f} us {2*M*N*K/m1/1e9:.0f} TF/s  ({(m0/m1-1)*100:+.1f} %)", flush=True)
        del a,w
if which in ("skinny",):
    # LoRA parameter-gradient contraction (dA: x [M, 4096]; dB: dy [M, 12288]) at the K3-TTA token count; HBM-bound on x
    for (M, K) in ((25200, 4096), (25200, 12288), (6240, 4096)):
        x = torch.randn(M, K, device=dev, dtype=bf); g = torch.randn(M, 64, device=dev, dtype=bf)
        ms = timeit(lambda: ops.tn_skinny(g, x, 8), n=20, warm=3)
        print(f"tn_skinny M={M} K={K} R=8: {ms*1e3:.0f} us  {M*K*2/ms/1e6:.0f} GB/s", flush=True)
if which in ("attn_skipmax",):
    # unit-score attention with (VAR 7) and without (VAR 3) the every-second-tile max skip, interleaved in one process;
    # inputs shaped like the DiT's: q, k RMS-normalised rows (|row| = sqrt(128)), q pre-scaled into log2 units
    import math
    N=46800; H=32; D=128
    g=torch.Generator(device=dev).manual_seed(0)
    def rmsn(t): return t*torch.rsqrt(t.float().pow(2).mean(-1,keepdim=True)+1e-6)
    q=rmsn(torch.randn(1,N,H,D,device=dev,generator=g)); k=rmsn(torch.randn(1,N,H,D,device=dev,generator=g)).to(bf)
    v=torch.randn(1,N,H,D,device=dev,generator=g).to(bf)
    q_pre=(q*(D**-0.5*math.log2(math.e))).to(bf)
    o3=torch.empty(1,N,H,D,device=dev,dtype=bf); o7=torch.empty_like(o3)
    fl=4*N*N*H*D
    res={"3":[],"7":[]}
    for rep in range(3):
        for var,o in (("3",o3),("7",o7)):
            os.environ["LCV_ATTN_VAR"]=var
            res[var].append(timeit(lambda: ops.attention(q_pre,k,v,math.log(2.0),out=o), n=3, warm=1))
    for var in ("3","7"):
        print(f"attn unit VAR={var}: {min(res[var]):.2f} ms  {fl/min(res[var])/1e9:.1f} TF/s   all: {[round(x,2) for x in res[var]]}", flush=True)
    d=(o3.float()-o7.float()); print("rel_l2 skipmax vs plain:", (d.norm()/o3.float().norm()).item(), "max abs", d.abs().max().item(), "finite", bool(torch.isfinite(o7.float()).all()))
