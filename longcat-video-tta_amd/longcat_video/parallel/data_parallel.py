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


LAST_MERGE_MISSING: List[int] = []      # ranks whose final rows rank 0 did not get in time (filled by gather_results)
_JOB_NONCE = {}                         # output directory -> the identity of THIS job (rank 0 draws it, every rank holds it)


def _deposit_path(output_dir: str, rank: int) -> str:
    return os.path.join(output_dir, f"results.rank{rank}.json")


def begin_job(output_dir: str, rank: int) -> None:
    """Call once per rank right after the output directory exists (collective at world size > 1: every rank calls it, at
    start-up, when all of them are alive): clears this rank's end-of-job deposit of an earlier run in the same directory (a
    resumed job) and agrees on a job nonce — rank 0 draws it, the host group broadcasts it.  Deposits carry the nonce, so
    rank 0 recognises a leftover of an earlier job by CONTENT; file times are never compared (on a shared file system they
    come from another host's clock)."""
    import uuid
    import torch.distributed as dist
    try:
        os.remove(_deposit_path(output_dir, rank))
    except FileNotFoundError:
        pass
    box = [uuid.uuid4().hex]
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast_object_list(box, src=0, group=host_group() if dist.get_backend() != "gloo" else None)
    _JOB_NONCE[os.path.abspath(output_dir)] = box[0]


def merge_wait_seconds(per_video_s: float = 0.0, videos_per_rank: int = 1) -> float:
    """How long rank 0 waits for the other ranks' final rows: the expected time of a whole shard again (ranks start together
    and do equal shares, so a healthy peer is at most about one video behind), never under 15 minutes; LCV_DP_MERGE_WAIT_S
    overrides."""
    env = os.environ.get("LCV_DP_MERGE_WAIT_S")
    if env:
        return float(env)
    return max(900.0, 2.0 * float(per_video_s) * max(int(videos_per_rank), 1))


def gather_results(local_rows: List[Dict[str, Any]], group=None, output_dir: Optional[str] = None,
                   wait_s: Optional[float] = None, poll_s: float = 2.0) -> Optional[List[Dict[str, Any]]]:
    """All ranks call; rank 0 gets the merged list, the others None.

    With `output_dir` (what the runners pass) the merge is a FILE rendezvous with a bounded wait, no collective at all: every
    rank deposits its final rows as `results.rank{r}.json` (atomic rename) and leaves; rank 0 polls for the W deposits for at
    most `wait_s` seconds.  A peer that wedged (a hung GPU does not exit, so no socket closes and a host-side gather would
    sit there for its whole timeout) costs that bounded wait: rank 0 then takes the peer's rows from its per-video
    `checkpoint.rank{r}.json` - written after every video - records the rank in `LAST_MERGE_MISSING`, still returns the merged
    rows so that `summary.json` gets written, and the runner exits non-zero (`exit_code_after_merge`).
    Without `output_dir`: one `gather_object` over the host (gloo) group, as before."""
    import torch.distributed as dist
    del LAST_MERGE_MISSING[:]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return merge_results([local_rows])
    if output_dir is None:
        if group is None:
            group = host_group()
        world = dist.get_world_size(group)
        out = [None] * world if dist.get_rank(group) == 0 else None
        dist.gather_object(local_rows, out, dst=0, group=group)
        return merge_results(out) if out is not None else None
    import time
    rank, world = dist.get_rank(), dist.get_world_size()
    tmp = _deposit_path(output_dir, rank) + ".tmp"
    with open(tmp, "w") as f:
        json.dump({"finished": True, "rank": rank, "job": _JOB_NONCE.get(os.path.abspath(output_dir)), "results": local_rows}, f,
                  default=str)
    os.replace(tmp, _deposit_path(output_dir, rank))
    if rank != 0:
        return None
    job = _JOB_NONCE.get(os.path.abspath(output_dir))
    deadline = time.time() + (merge_wait_seconds() if wait_s is None else float(wait_s))
    per_rank: Dict[int, List[Dict[str, Any]]] = {0: local_rows}
    while True:
        for r in range(1, world):
            p = _deposit_path(output_dir, r)
            if r in per_rank or not os.path.exists(p):
                continue
            try:
                with open(p) as f:
                    dep = json.load(f)
                if dep.get("job") == job:                # else: a leftover of an earlier job in this directory
                    per_rank[r] = dep["results"]
            except (OSError, ValueError, KeyError):
                pass                                     # caught mid-replace on a network file system: next poll
        if len(per_rank) == world or time.time() >= deadline:
            break
        time.sleep(poll_s)
    for r in range(1, world):
        if r in per_rank:
            continue
        LAST_MERGE_MISSING.append(r)
        ck = load_checkpoint(output_dir, r)
        per_rank[r] = ck["results"] if ck else []
        print(f"  WARNING: rank {r} did not deliver its final results within the wait; merged its per-video checkpoint "
              f"({len(per_rank[r])} rows) instead")
    return merge_results([per_rank[r] for r in sorted(per_rank)])


def exit_code_after_merge() -> int:
    """0 after a complete merge; 3 when rank 0 had to fall back to a peer's checkpoint file (the summary exists, the job did
    not finish cleanly)."""
    return 3 if LAST_MERGE_MISSING else 0


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
