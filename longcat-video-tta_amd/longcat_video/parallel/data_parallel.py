"""Data parallelism over independent test videos: one process per GPU, rank r owns the videos
`idx = r (mod W)`, no collective on the data path; results are merged once at the end into the reference's
`checkpoint.json` / `summary.json` shapes (`delta_experiment/scripts/common.py:2040-2059`).

The reference iterates `eval_videos` sequentially with per-video adapter reset and per-video generation seed
`args.seed + idx` (`lora_experiment/scripts/run_lora_tta.py:974, 1127, 1215`); the only cross-video state is the global
torch RNG stream used for sigma / eps, so each video is re-seeded with `seed + idx` here (declared deviation)."""
import json
import os
from typing import Any, Dict, List, Optional, Sequence

import torch


def shard_indices(n_items: int, rank: int, world_size: int, start: int = 0) -> List[int]:
    """Indices owned by `rank`: start + rank, start + rank + W, ... (< n_items)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(start + rank, n_items, world_size))


def seed_for_video(base_seed: int, idx: int) -> int:
    return int(base_seed) + int(idx)


def merge_results(per_rank: Sequence[Sequence[Dict[str, Any]]]) -> List[Dict[str, Any]]:
    """Merge per-rank result rows (each carrying its global `idx`) into reference order."""
    rows = [r for rank_rows in per_rank for r in rank_rows]
    rows.sort(key=lambda r: r["idx"])
    seen = set()
    for r in rows:
        if r["idx"] in seen:
            raise ValueError(f"video idx {r['idx']} reported by two ranks")
        seen.add(r["idx"])
    return rows


_HOST_GROUP = None


def host_group():
    """A gloo (host-side) process group with a day-long timeout for the end-of-job result merge.  Ranks finish their video
    shards at very different times (early stopping, skipped videos): a rank parked in an RCCL collective while the others
    still work trips the collective watchdog and aborts the job before `summary.json` exists; a host-side gather just waits.
    Collective: every rank must call it once, at start-up (tta/runner_common.setup_distributed does)."""
    global _HOST_GROUP
    import datetime
    import torch.distributed as dist
    if _HOST_GROUP is None and dist.is_initialized() and dist.get_world_size() > 1:
        _HOST_GROUP = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=24))
    return _HOST_GROUP


def gather_results(local_rows: List[Dict[str, Any]], group=None) -> Optional[List[Dict[str, Any]]]:
    """All ranks call; rank 0 gets the merged list (one `gather_object` over the HOST group: the only collective of the DP
    path, and never on the device's collective queue)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return merge_results([local_rows])
    if group is None:
        group = host_group()
    world = dist.get_world_size(group)
    out = [None] * world if dist.get_rank(group) == 0 else None
    dist.gather_object(local_rows, out, dst=0, group=group)
    return merge_results(out) if out is not None else None


def write_checkpoint(output_dir: str, next_idx: int, results: List[Dict[str, Any]], rank: Optional[int] = None):
    """`checkpoint.json` = {"next_idx", "results"} (common.py:2055-2059); per-rank shards are `checkpoint.rank{r}.json`."""
    name = "checkpoint.json" if rank is None else f"checkpoint.rank{rank}.json"
    tmp = os.path.join(output_dir, name + ".tmp")
    with open(tmp, "w") as f:
        json.dump({"next_idx": next_idx, "results": results}, f, indent=2)
    os.replace(tmp, os.path.join(output_dir, name))


def load_checkpoint(output_dir: str, rank: Optional[int] = None) -> Optional[Dict[str, Any]]:
    name = "checkpoint.json" if rank is None else f"checkpoint.rank{rank}.json"
    p = os.path.join(output_dir, name)
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f)


def contiguous_next_idx(results: Sequence[Dict[str, Any]], start: int = 0) -> int:
    """Resume point of a merged run: the first index not yet present (the reference's `next_idx` semantics)."""
    have = {r["idx"] for r in results}
    i = start
    while i in have:
        i += 1
    return i
