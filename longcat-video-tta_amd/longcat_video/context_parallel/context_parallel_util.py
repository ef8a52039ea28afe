"""Context-parallel bookkeeping kept for API compatibility (baseline_experiment/scripts/run_baseline.py:76-79 calls
`init_context_parallel(context_parallel_size=1, global_rank=..., world_size=...)`).  The MI355X build shards long
sequences on the FRAME axis with an RCCL K/V all-gather (longcat_video.parallel.sequence_parallel), not on H x W."""
_STATE = {"size": 1, "rank": 0, "world_size": 1, "group": None}


def init_context_parallel(context_parallel_size: int = 1, global_rank: int = 0, world_size: int = 1):
    if context_parallel_size < 1 or world_size % context_parallel_size:
        raise ValueError("context_parallel_size must divide world_size")
    _STATE.update(size=context_parallel_size, rank=global_rank % context_parallel_size, world_size=world_size)
    if context_parallel_size > 1:
        import torch.distributed as dist
        base = (global_rank // context_parallel_size) * context_parallel_size
        groups = [dist.new_group(list(range(b, b + context_parallel_size)))
                  for b in range(0, world_size, context_parallel_size)]
        _STATE["group"] = groups[base // context_parallel_size]


def get_cp_size() -> int:
    return _STATE["size"]


def get_cp_rank() -> int:
    return _STATE["rank"]


def get_cp_group():
    return _STATE["group"]
