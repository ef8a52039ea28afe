"""Frame-axis sequence parallelism: rank r holds a contiguous run of latent frames (N/W tokens);
every per-token op is local (the adaLN table is per frame, RoPE uses GLOBAL positions via `pos_off`), and self-attention
all-gathers K and V (post-norm, post-RoPE) so each rank runs local-Q x full-KV on the flash kernel.

Collectives (torch.distributed; backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests):
  forward   all_gather of K, V rows (uneven shards allowed: 31 frames over 8 ranks = 4,4,4,4,4,4,4,3)
  backward  reduce_scatter of dK, dV (the adjoint of the all-gather)
xGMI is point-to-point (7 links per GPU): RCCL's all-gather on a fully connected 8-GPU node is issued as direct peer
exchanges, so a per-layer message of 2*N*C*2 bytes (792 MB at K5) moves over all 7 links, not around a ring.
"""
from typing import List, Tuple

import torch


def frame_shards(num_frames: int, world_size: int) -> List[int]:
    """Frames per rank, as even as possible, larger shards first (sum == num_frames)."""
    base, rem = divmod(num_frames, world_size)
    return [base + (1 if r < rem else 0) for r in range(world_size)]


def token_offset(rank: int, counts: List[int], tokens_per_frame: int) -> int:
    return sum(counts[:rank]) * tokens_per_frame


def _gather_rows(x: torch.Tensor, counts: List[int], S: int, group=None) -> torch.Tensor:
    """All-gather along the token axis.  Shards may be uneven: each rank pads to the largest shard so the collective
    is one equal-sized all_gather (what RCCL / gloo implement natively), then the pads are dropped."""
    import torch.distributed as dist
    world = len(counts)
    n_max = max(counts) * S
    if x.shape[1] < n_max:
        pad = torch.zeros((x.shape[0], n_max - x.shape[1]) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        x = torch.cat([x, pad], dim=1)
    outs = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(outs, x.contiguous(), group=group)
    return torch.cat([o[:, : c * S] for o, c in zip(outs, counts)], dim=1)


def all_gather_kv(k_local: torch.Tensor, v_local: torch.Tensor, counts: List[int], tokens_per_frame: int,
                  group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """[B, n_local, H, D] shards -> full [B, N, H, D] K and V in frame order."""
    return (_gather_rows(k_local, counts, tokens_per_frame, group),
            _gather_rows(v_local, counts, tokens_per_frame, group))


def reduce_scatter_kv_grad(d_full: torch.Tensor, counts: List[int], tokens_per_frame: int, group=None) -> torch.Tensor:
    """Sum the full-length dK (or dV) over ranks and keep this rank's rows."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    d = d_full.contiguous().clone()
    dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)   # uneven shards: all-reduce + slice (reduce_scatter needs equal parts)
    lo = sum(counts[:rank]) * tokens_per_frame
    return d[:, lo:lo + counts[rank] * tokens_per_frame].contiguous()


class SPContext:
    """Frame-axis shard of one forward pass: which latent frames / tokens this rank owns and the K/V exchange.

    `group` is a torch.distributed process group ("nccl" = RCCL over xGMI in production).  With the gloo backend
    (CPU tests, or several ranks sharing one GPU) device tensors are staged through host memory for the collective."""

    def __init__(self, num_frames: int, tokens_per_frame: int, group=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.S = tokens_per_frame
        self.counts = frame_shards(num_frames, self.world)
        self.t0 = sum(self.counts[: self.rank])
        self.t1 = self.t0 + self.counts[self.rank]
        self.num_frames = num_frames
        self._host_staged = dist.get_backend(group) == "gloo"

    @property
    def token_offset(self) -> int:
        return self.t0 * self.S

    def _coll(self, fn, x: torch.Tensor) -> torch.Tensor:
        if self._host_staged and x.is_cuda:
            return fn(x.cpu()).to(x.device)
        return fn(x)

    def all_gather_kv(self, k_local: torch.Tensor, v_local: torch.Tensor):
        k = self._coll(lambda t: _gather_rows(t, self.counts, self.S, self.group), k_local)
        v = self._coll(lambda t: _gather_rows(t, self.counts, self.S, self.group), v_local)
        return k, v

    def reduce_scatter_kv(self, d_full: torch.Tensor) -> torch.Tensor:
        """Adjoint of `all_gather_kv` for one tensor: sum the full-length gradient over ranks, keep this rank's rows."""
        return self._coll(lambda t: reduce_scatter_kv_grad(t, self.counts, self.S, self.group), d_full)

    def gather_frames_autograd(self, x_local: torch.Tensor) -> torch.Tensor:
        """`gather_frames` with a backward: every rank evaluates the SAME loss on the gathered prediction, so the gradient
        w.r.t. the full tensor is identical everywhere and each rank simply keeps the slice of its own frames."""
        return _GatherFramesFn.apply(x_local, self)

    def all_reduce_grads(self, params) -> None:
        """Each rank back-propagates through its own token shard only: parameter gradients are partial sums."""
        import torch.distributed as dist
        for p_ in params:
            if p_.grad is not None:
                g = p_.grad
                r = self._coll(lambda t: (dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group), t)[1], g.contiguous())
                p_.grad = r.to(g.dtype)

    def gather_frames(self, x_local: torch.Tensor) -> torch.Tensor:
        """[B, C, T_local, H, W] -> [B, C, T, H, W] on every rank."""
        B, C, Tl, H, W = x_local.shape
        rows = x_local.permute(0, 2, 1, 3, 4).reshape(B, Tl, C * H * W).contiguous()
        full = self._coll(lambda t: _gather_rows(t, self.counts, 1, self.group), rows)
        return full.view(B, self.num_frames, C, H, W).permute(0, 2, 1, 3, 4).contiguous()


class _GatherFramesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, sp):
        ctx.sp = sp
        return sp.gather_frames(x_local)

    @staticmethod
    def backward(ctx, dfull):
        sp = ctx.sp
        return dfull[:, :, sp.t0:sp.t1].contiguous(), None
