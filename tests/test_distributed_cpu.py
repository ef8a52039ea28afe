"""CPU, world_size 2 over gloo: the N>1 code paths (DP result gather/merge + checkpoint shards, and the sequence-parallel
K/V all-gather bookkeeping) with real process groups."""
import json
import os
import sys
import tempfile
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from longcat_video.parallel import data_parallel as dp
        from longcat_video.parallel import sequence_parallel as sp
        # ---- DP: shard 7 videos, fake per-video results, per-rank checkpoint shard, gather on rank 0
        mine = dp.shard_indices(7, rank, world)
        rows = [{"idx": i, "success": True, "seed": dp.seed_for_video(42, i), "rank": rank} for i in mine]
        dp.write_checkpoint(tmp, (mine[-1] + world) if mine else rank, rows, rank=rank)
        merged = dp.gather_results(rows)
        if rank == 0:
            assert [r["idx"] for r in merged] == list(range(7))
            assert all(r["seed"] == 42 + r["idx"] for r in merged)
            dp.write_checkpoint(tmp, dp.contiguous_next_idx(merged), merged)
        else:
            assert merged is None
        # ---- the runners' form: a file rendezvous with a bounded wait, no collective (round 3)
        dp.begin_job(tmp, rank)
        dist.barrier()
        merged = dp.gather_results(rows, output_dir=tmp, wait_s=60.0, poll_s=0.05)
        if rank == 0:
            assert [r["idx"] for r in merged] == list(range(7)) and dp.exit_code_after_merge() == 0
        else:
            assert merged is None
        dist.barrier()
        # a wedged peer: rank 1 never delivers (its GPU hung; it did not exit).  Rank 0 waits the bounded time, takes rank 1's
        # rows from the per-video checkpoint it wrote after its LAST FINISHED video, still merges, and reports a non-zero code
        dp.begin_job(tmp, rank)
        dist.barrier()
        if rank == 1:
            dp.write_checkpoint(tmp, mine[0] + world, rows[:1], rank=rank)        # one video done, then it hangs
            # ... and a deposit of an EARLIER job lies in the directory (a peer that died before begin_job could clear it): it
            # carries another job's nonce and is recognised by content, whatever its file time says
            Path(dp._deposit_path(tmp, 1)).write_text(json.dumps({"finished": True, "rank": 1, "job": "an-earlier-job", "results": rows}))
        dist.barrier()
        if rank == 0:
            import time
            t0 = time.time()
            merged = dp.gather_results(rows, output_dir=tmp, wait_s=0.5, poll_s=0.05)
            assert 0.4 < time.time() - t0 < 10.0
            assert dp.LAST_MERGE_MISSING == [1] and dp.exit_code_after_merge() == 3
            assert [r["idx"] for r in merged] == sorted([r["idx"] for r in rows] + [mine_of_1[0] for mine_of_1 in [dp.shard_indices(7, 1, world)]])
        dist.barrier()
        if rank == 1:
            dp.write_checkpoint(tmp, (mine[-1] + world) if mine else rank, rows, rank=rank)   # restore for the asserts after the spawn
        # ---- SP: frame-axis shards (uneven: 5 frames over 2 ranks -> 3 + 2), all-gather of K/V rows
        T, S, H, D = 5, 6, 2, 8
        counts = sp.frame_shards(T, world)
        assert counts == [3, 2] and sum(counts) == T
        t0 = sum(counts[:rank])
        g = torch.Generator().manual_seed(0)
        full_k = torch.randn(1, T * S, H, D, generator=g)
        full_v = torch.randn(1, T * S, H, D, generator=g)
        lo, hi = t0 * S, (t0 + counts[rank]) * S
        k_all, v_all = sp.all_gather_kv(full_k[:, lo:hi].contiguous(), full_v[:, lo:hi].contiguous(), counts, S)
        assert torch.equal(k_all, full_k) and torch.equal(v_all, full_v)
        assert sp.token_offset(rank, counts, S) == lo
        # the gradient of an all-gather is a reduce-scatter: every rank contributes a full-length dK
        dk_full = torch.full((1, T * S, H, D), float(rank + 1))
        dk_local = sp.reduce_scatter_kv_grad(dk_full, counts, S)
        assert dk_local.shape[1] == counts[rank] * S and torch.all(dk_local == 3.0)
        # the context object: gathered K/V are VIEWS of one padded buffer (pads at the end), the gradient buffer is allocated
        # in the reduce-scatter's layout, CFG batch of two
        ctx = sp.SPContext(T, S)
        assert ctx.counts == [3, 2] and sp.pads_at_end(ctx.counts) and ctx.token_offset == lo
        k2 = torch.stack([full_k[0], full_k[0] * 2.0])
        kg, vg = ctx.all_gather_kv(k2[:, lo:hi].contiguous(), k2[:, lo:hi].contiguous())
        assert torch.equal(kg, k2) and kg.shape == (2, T * S, H, D) and kg._base is not None and kg._base.shape[1] == 2 * 3 * S
        dg = ctx.padded_zeros(kg)
        assert dg.shape == kg.shape and dg._lcv_padded.shape[1] == 2 * 3 * S and not dg.any()
        dg += float(rank + 1)
        dl = ctx.reduce_scatter_kv(dg)
        assert dl.shape == (2, counts[rank] * S, H, D) and torch.all(dl == 3.0)
        # round 3: the shard unit is a TOKEN ROW.  5 frames x 3 rows of 2 tokens = 15 rows -> 8 + 7 (frames would give 3 + 2):
        # shards end inside a frame, conditioning frames pinned in the sequence split over the ranks by rows
        rctx = sp.SPContext(T, S, rows_per_frame=3)
        assert rctx.counts == [8, 7] and rctx.S == 2 and rctx.grid == (5, 3, 2) and rctx.num_units == 15
        assert rctx.token_offset == (0 if rank == 0 else 16) and sp.pads_at_end(rctx.counts)
        assert rctx.local_units_of_leading_frames(2) == (6 if rank == 0 else 0)      # 2 cond frames = rows 0..5: all on rank 0
        assert rctx.local_units_of_leading_frames(3) == (8 if rank == 0 else 1)      # rows 0..8: 8 on rank 0, 1 on rank 1
        lo_r, hi_r = rctx.token_offset, rctx.token_offset + rctx.counts[rank] * rctx.S
        kg, vg = rctx.all_gather_kv(k2[:, lo_r:hi_r].contiguous(), (k2 * 3.0)[:, lo_r:hi_r].contiguous())
        assert torch.equal(kg, k2) and torch.equal(vg, k2 * 3.0) and kg._base.shape[1] == 2 * 8 * 2   # one padded buffer per tensor
        dgr = rctx.padded_zeros(kg); dgr += float(rank + 1)
        dlr = rctx.reduce_scatter_kv(dgr)
        assert dlr.shape == (2, rctx.counts[rank] * 2, H, D) and torch.all(dlr == 3.0)
        # the prediction comes back by rows of height 2: [B, C, rows_local, 2, W] -> [B, C, 15, 2, W]
        pred = torch.arange(15 * 2 * 4, dtype=torch.float32).view(1, 1, 15, 2, 4)
        got = rctx.gather_frames(pred[:, :, rctx.t0:rctx.t1].contiguous())
        assert torch.equal(got, pred)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_frame_shards_put_the_pads_at_the_end_when_they_can():
    sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
    from longcat_video.parallel import sequence_parallel as sp
    assert sp.frame_shards(31, 8) == [4, 4, 4, 4, 4, 4, 4, 3] and sp.pads_at_end(sp.frame_shards(31, 8))     # K5 on 8 GPUs
    assert sp.frame_shards(32, 8) == [4] * 8 and sp.frame_shards(5, 2) == [3, 2] and sp.frame_shards(3, 2) == [2, 1]
    assert sp.frame_shards(13, 8) == [2, 2, 2, 2, 2, 1, 1, 1] and not sp.pads_at_end(sp.frame_shards(13, 8))  # no empty rank
    for T in range(1, 40):
        for W in (1, 2, 3, 4, 8):
            c = sp.frame_shards(T, W)
            assert sum(c) == T and len(c) == W and all(x >= 0 for x in c) and c == sorted(c, reverse=True)


def test_token_row_shards_balance_the_headline_shapes():
    """Round 3: whole latent frames cannot balance 49x720p on 8 ranks (13 frames -> 2,2,2,2,2,1,1,1: 0.81, and
    gaps inside the padded sequence); token rows can, and always keep the pads at the end (the single-collective path)."""
    sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
    from longcat_video.parallel import sequence_parallel as sp
    cases = {"K3 49x720p": (13, 45, 80), "K5 121x480p": (31, 30, 52), "K2 49x480p": (13, 30, 52), "K3' 49x90x160": (49, 45, 80)}
    for name, (T, rows, per_row) in cases.items():
        for W in (2, 4, 8):
            c = sp.row_shards(T * rows, W)
            assert sum(c) == T * rows and sp.pads_at_end(c), (name, W, c)
            tokens = [x * per_row for x in c]
            eff = sum(tokens) / (W * max(tokens))
            assert eff > 0.98, (name, W, c, eff)
            assert max(tokens) - min(tokens) <= (W - 1) * per_row           # at most W - 1 rows of imbalance, all on the last rank
    c = sp.row_shards(13 * 45, 8)
    assert c == [74] * 7 + [67] and [x * 80 for x in c] == [5920] * 7 + [5360]      # K3 on 8 ranks: 5 850 +- 80 tokens is the ceiling share
    f = sp.frame_shards(13, 8)
    assert sum(f) / (8 * max(f)) < 0.82 and not sp.pads_at_end(f)                   # what rounds 1-2 did


def test_world_size_2_gloo():
    port = 29500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
        ck = json.loads((Path(tmp) / "checkpoint.json").read_text())
        assert ck["next_idx"] == 7 and len(ck["results"]) == 7
        assert (Path(tmp) / "checkpoint.rank1.json").exists()


def _worker_8(rank, world, port):
    """8 ranks over gloo at the two sequence-parallel geometries of BASELINE.json (K3: 13 frames x 45 rows of 80 tokens; K5: 31
    frames x 30 rows of 52 tokens), with 4 conditioning frames pinned in front (H x D shrunk to 1 x 4: the bookkeeping is what
    is under test): shard sizes, global offsets, K/V gather of a CFG pair, the reduce-scatter adjoint, the conditioning rows
    each rank holds, the row-wise gather of the prediction."""
    sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from longcat_video.parallel import sequence_parallel as sp
        for name, (T, rows, per_row, counts) in {"K3": (13, 45, 80, [74] * 7 + [67]), "K5": (31, 30, 52, [117] * 7 + [111])}.items():
            ctx = sp.SPContext(T, rows * per_row, rows_per_frame=rows)
            assert ctx.counts == counts and ctx.S == per_row and ctx.num_units == T * rows and ctx.grid == (T, rows, per_row)
            assert not ctx.grouped                                         # gloo: one launch per tensor
            assert ctx.token_offset == sum(counts[:rank]) * per_row and sp.pads_at_end(ctx.counts)
            N = T * rows * per_row
            g = torch.Generator().manual_seed(7)
            full = torch.randn(2, N, 1, 4, generator=g)                    # the CFG pair, identical on every rank
            lo, hi = ctx.token_offset, ctx.token_offset + counts[rank] * per_row
            k, v = ctx.all_gather_kv(full[:, lo:hi].contiguous(), (full * 2.0)[:, lo:hi].contiguous())
            assert k.shape == (2, N, 1, 4) and torch.equal(k, full) and torch.equal(v, full * 2.0)
            assert k._base.shape[1] == world * counts[0] * per_row         # ONE padded buffer, pads at the end
            d = ctx.padded_zeros(k)
            d += float(rank + 1)
            dl = ctx.reduce_scatter_kv(d)
            assert dl.shape == (2, counts[rank] * per_row, 1, 4) and torch.all(dl == float(sum(range(1, world + 1))))
            # 4 conditioning frames = the first 4 * rows units: a prefix of the sequence, split over the first ranks by rows
            held = ctx.local_units_of_leading_frames(4)
            tot = torch.tensor([held]); dist.all_reduce(tot)
            assert tot.item() == 4 * rows
            first = 4 * rows
            assert held == max(0, min(ctx.t1, first) - ctx.t0) and (held == counts[rank] or ctx.t1 > first or held == 0 or ctx.t1 == first)
            # the prediction returns by token rows of height 2
            pred = torch.arange(T * rows * 2 * 3, dtype=torch.float32).view(1, 1, T * rows, 2, 3)
            assert torch.equal(ctx.gather_frames(pred[:, :, ctx.t0:ctx.t1].contiguous()), pred)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world_size_8_gloo_k3_and_k5_geometries():
    mp.spawn(_worker_8, args=(8, 31500 + os.getpid() % 2000), nprocs=8, join=True)


def test_bench_launches_its_own_ranks_for_more_than_one_gpu():
    """`python bench.py --gpus N` without a torch.distributed environment (the form the driver uses) starts N ranks under
    torch.distributed.run as a child process; --dry-launch prints that command.  The parent never initialises the GPU."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--dry-launch"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    cmd = r.stdout.strip().split()
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(str(ROOT / "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "2", "--warmup", "1"]          # the ranks get the caller's flags, minus --dry-launch
    # a mismatched environment is an error, not a silent single-GPU run
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
