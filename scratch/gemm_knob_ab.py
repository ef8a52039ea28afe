"""In-process A/B of an environment knob of the default GEMM at the four K3 projection shapes (+ two K3-TTA shapes).
usage: gemm_knob_ab.py KNOB v1,v2,...   Prints TF/s per value and checks the outputs are bit-identical."""
import os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
dev = "cuda"; bf = torch.bfloat16
knob, vals = sys.argv[1], sys.argv[2].split(",")
def timeit(fn, n=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K, name) in ((93600, 12288, 4096, "qkv"), (93600, 4096, 4096, "proj"), (93600, 22016, 4096, "w13"), (93600, 4096, 11008, "w2"),
                        (25200, 12288, 4096, "tta qkv"), (25200, 4096, 11008, "tta w2")):
    a = (torch.randn(M, K, device=dev, generator=g)).to(bf); w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(bf); b = torch.randn(N, device=dev, generator=g).to(bf)
    res = {v: [] for v in vals}; outs = {}
    for rep in range(2):
        for v in vals:
            os.environ[knob] = v
            res[v].append(timeit(lambda: ops.gemm_nt(a, w, b)))
            outs[v] = ops.gemm_nt(a, w, b)
    same = all(torch.equal(outs[vals[0]], outs[v]) for v in vals)
    print(f"{name:8s} " + "  ".join(f"{knob}={v}: {2 * M * N * K / min(res[v]) / 1e9:6.0f} TF/s" for v in vals) + f"   bit-identical {same}", flush=True)
    del a, w
