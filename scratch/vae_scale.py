import sys, time, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
dev="cuda"
vae=AutoencoderKLWan(device=dev).init_synthetic_()
h,w=(90,160) if len(sys.argv)<2 or sys.argv[1]=="720p" else (60,104)
z=torch.randn(1,16,13,h,w,device=dev)
for i in range(2):
    torch.cuda.synchronize(); t0=time.time()
    v=vae.decode(z.to(torch.bfloat16))[0]
    torch.cuda.synchronize(); print("decode", v.shape, f"{time.time()-t0:.2f}s", "peak GB", torch.cuda.max_memory_allocated()/2**30, "finite", torch.isfinite(v).all().item(), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "encode":          # the clip just decoded, back through the encoder
    x = v.to(torch.bfloat16).clamp(-1, 1)
    for i in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        lat = vae.encode(x)
        torch.cuda.synchronize(); print("encode", tuple(x.shape), f"{time.time()-t0:.2f}s", flush=True)
