"""GPU: the RCCL ("nccl" backend) branches of the sequence-parallel code on a WORLD-SIZE-1 group.

Every other multi-rank test in this suite runs over gloo with host staging (two ranks cannot share one GPU over RCCL), so
before this file the production branches — `dist._coalescing_manager` around the K / V `all_gather_into_tensor` calls,
`reduce_scatter_tensor` into views of the padded gradient buffer, device-resident all-reduce of adapter gradients, the row-wise
gather of the prediction, `bench.py`'s `init_process_group("nccl")` — had never executed.  One rank exercises the same torch API
calls on device tensors; with one rank every collective is an identity, so the sequence-parallel forward must equal the plain
forward BIT FOR BIT.  (the exchange itself over xGMI stays unmeasured until an 8-GPU node runs it.)
"""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, port, out_path):
    for p in (str(ROOT), str(ROOT / "longcat-video-tta_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    notes = []
    try:
        from longcat_video.parallel import sequence_parallel as sp
        assert dist.get_backend() == "nccl"
        T, rows, per_row, H, D = 3, 5, 6, 2, 128
        ctx = sp.SPContext(T, rows * per_row, rows_per_frame=rows)
        assert not ctx._host_staged and ctx.counts == [T * rows] and ctx.token_offset == 0
        notes.append(f"grouped={ctx.grouped}")
        assert ctx.grouped, "torch 2.10 + RCCL: the grouped K/V exchange is expected to be available"
        N = T * rows * per_row
        g = torch.Generator(device="cuda").manual_seed(3)
        k = torch.randn(2, N, H, D, generator=g, device="cuda").to(torch.bfloat16)      # a CFG pair: 2 x B = 4 collectives, ONE launch
        v = torch.randn(2, N, H, D, generator=g, device="cuda").to(torch.bfloat16)
        kg, vg = ctx.all_gather_kv(k, v)
        assert kg.is_cuda and torch.equal(kg, k) and torch.equal(vg, v)
        d = ctx.padded_zeros(kg)
        d += 1.5
        dl = ctx.reduce_scatter_kv(d)                                                    # reduce_scatter_tensor into out[b] views
        assert dl.shape == kg.shape and torch.all(dl == 1.5)
        p = torch.nn.Parameter(torch.ones(7, device="cuda", dtype=torch.bfloat16))
        p.grad = torch.full_like(p, 2.0)
        ctx.all_reduce_grads([p])
        assert torch.all(p.grad == 2.0) and p.grad.dtype == torch.bfloat16
        pred = torch.randn(1, 16, T * rows, 2, 12, generator=g, device="cuda")
        assert torch.equal(ctx.gather_frames(pred), pred)
        full = ctx.gather_frames_autograd(pred.clone().requires_grad_(True))
        assert full.requires_grad
        # the ungrouped path on the same backend (what a torch without the coalescing manager would run)
        os.environ["LCV_SP_COALESCE"] = "0"
        sp.SPContext._GROUPED.clear()
        ctx2 = sp.SPContext(T, rows * per_row, rows_per_frame=rows)
        assert not ctx2.grouped
        kg2, vg2 = ctx2.all_gather_kv(k, v)
        assert torch.equal(kg2, k) and torch.equal(vg2, v)
        os.environ["LCV_SP_COALESCE"] = "1"
        sp.SPContext._GROUPED.clear()

        # the model: sequence-parallel forward on the one-rank RCCL group == the plain forward, bitwise (CFG pair, one pinned frame)
        from oracle import dit_oracle as orc
        from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
        BF16 = torch.bfloat16
        cfg = orc.small_config(hidden_size=256, depth=2, num_heads=2, caption_channels=64)
        m = LongCatVideoTransformer3DModel(device="cuda", dtype=BF16, hidden_size=256, depth=2, num_heads=2,
                                           caption_channels=64, adaln_tembed_dim=64).eval()
        m.load_state_dict(orc.make_params(cfg, seed=9, std=0.05), strict=False)
        gc = torch.Generator().manual_seed(1)
        hs = torch.randn(2, 16, 3, 10, 12, generator=gc).to(BF16).cuda()
        y = torch.randn(2, 1, 16, 64, generator=gc).to(BF16).cuda()
        mask = torch.zeros(2, 16, dtype=torch.int64); mask[0, :10] = 1; mask[1, :4] = 1
        ts = torch.tensor([[0.0, 640.0, 640.0]] * 2).to(BF16).cuda()
        with torch.no_grad():
            ref = m(hs, ts, y, mask.cuda(), num_cond_latents=1)
            m.enable_sequence_parallel(None)
            got = m(hs, ts, y, mask.cuda(), num_cond_latents=1)
            m.disable_sequence_parallel()
        assert torch.equal(got, ref), (got - ref).abs().max().item()
        # what bench.py does with the group: a barrier, a MAX all-reduce of the elapsed time, the rank census
        dist.barrier()
        t = torch.tensor([1.25], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, device="cuda"); dist.all_reduce(ones)
        assert t.item() == 1.25 and int(ones.item()) == 1
        notes.append("ok")
    finally:
        Path(out_path).write_text(" ".join(notes))
        dist.destroy_process_group()


def test_rccl_branches_on_a_one_rank_group(tmp_path):
    out = tmp_path / "notes.txt"
    mp.spawn(_worker, args=(30700 + os.getpid() % 200, str(out)), nprocs=1, join=True)
    notes = out.read_text()
    print("RCCL one-rank group:", notes)
    assert notes.endswith("ok")
