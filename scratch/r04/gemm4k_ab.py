"""Round 4: the four-wave K64 kernel (csrc/gemm4k.h, LCV_GEMM_TILE=k) - bit-identity against the one-barrier kernel (tile 6) on
ragged shapes with every epilogue, then same-process A/B timing against the 8-phase default (9), gemm4w (4) and hipBLASLt (torch)
on the four K3 projection shapes at M = 93 600.  argv[1] = "check" | "time" | "all"."""
import os, sys, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from lcv_hip import ops
import lcv_hip.lib as L
from lcv_hip.lib import LCV_EPI_GATE_RESIDUAL, LCV_EPI_SWIGLU
dev = "cuda"; bf = torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def rn(*s, seed=0, scale=1.0, dtype=bf):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(dtype).to(dev)


def tile(t):
    L.set_knob("LCV_GEMM_TILE", t)


if what in ("check", "all"):
    bad = 0
    cases = [(300, 512, 256, 0, "plain"), (256 * 3 + 19, 256 * 2 + 40, 512, 0, "plain"), (2500, 512, 192, 128, "plain"),
             (256 * 37 + 19, 256 * 11 + 40, 512, 0, "plain"), (4096 + 70, 2048 + 30, 4096, 64, "plain"),
             (4096, 1024, 1024, 0, "gate"), (4096 + 33, 1024, 1024, 64, "gate"), (4096, 2048, 1024, 0, "swiglu"), (4096 + 7, 2048, 1024, 0, "swiglu_train"),
             (12480, 4096, 4096, 0, "gate"), (2048, 1024, 1024, 0, "f32out")]
    for (M, N, K, K2, kind) in cases:
        a = rn(M, K, seed=1); w = rn(N, K, seed=2, scale=0.05); b = rn(N, seed=3)
        kw = {}
        if K2:
            kw.update(a2=rn(M, K2, seed=4), w2=rn(N, K2, seed=5, scale=0.05))
        if kind == "gate":
            T = 4
            kw.update(epilogue=LCV_EPI_GATE_RESIDUAL, resid=rn(M, N, seed=6), mod=rn(1, T, 3 * N, seed=7, dtype=torch.float32), gate_idx=2,
                      rows_per_frame=(M + T - 1) // T)
        if kind.startswith("swiglu"):
            kw.update(epilogue=LCV_EPI_SWIGLU)
            if kind == "swiglu_train":
                kw.update(swiglu_aux=torch.empty(M, N, device=dev, dtype=bf))
        if kind == "f32out":
            kw.update(out_f32=True)
        try:
            tile("6"); ref = ops.gemm_nt(a, w, b, **kw)
            ref_pre = kw["swiglu_aux"].clone() if "swiglu_aux" in kw else None
            tile("k")
            for rep in range(3):
                if "swiglu_aux" in kw:
                    kw["swiglu_aux"].zero_()
                got = ops.gemm_nt(a, w, b, **kw)
                torch.cuda.synchronize()
                ok = torch.equal(got, ref) and (ref_pre is None or torch.equal(kw["swiglu_aux"], ref_pre))
                if not ok:
                    d = (got.float() - ref.float()).abs()
                    print(f"MISMATCH {M}x{N}x{K}+{K2} {kind} rep {rep}: max abs {d.max().item():.4g}, differing {(d > 0).float().mean().item():.4f}", flush=True)
                    bad += 1
                    break
            else:
                print(f"ok {M}x{N}x{K}+{K2} {kind}", flush=True)
        except TypeError as ex:
            print("skip", kind, ex, flush=True)
    print("check done, mismatches:", bad, flush=True)
    if bad:
        sys.exit(1)

if what in ("time", "all"):
    def timeit(fn, n=8, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    M = 93600
    for (N, K, name) in ((4096, 4096, "proj"), (12288, 4096, "qkv"), (22016, 4096, "w13"), (4096, 11008, "w2")):
        a = rn(M, K, seed=11); w = rn(N, K, seed=12, scale=0.02); b = rn(N, seed=13)
        fl = 2 * M * N * K
        row = []
        for rnd in range(2):
            for t in ("9", "k", "4"):
                tile(t)
                ms = timeit(lambda: ops.gemm_nt(a, w, b))
                row.append(f"[{t}] {ms:.3f} ms {fl / ms / 1e9:.0f}")
            ms = timeit(lambda: torch.nn.functional.linear(a, w, b))
            row.append(f"[hipblaslt] {ms:.3f} ms {fl / ms / 1e9:.0f}")
        print(f"{name} M={M} N={N} K={K}: " + " | ".join(row), flush=True)
        del a, w
        torch.cuda.empty_cache()
