"""Multi-GPU glue: one process per GPU, independent baseband segments per rank (no data-path
collective), and the single exchange step of the design -- a gather of the decoded TS bytes on
rank 0 (RCCL over xGMI with the "nccl" backend; the same code runs under "gloo" on CPU tensors,
which is how tests/ cover it without GPUs)."""
import torch
import torch.distributed as dist


def split_superframes(n_superframes, world):
    """Contiguous, near-equal runs of whole superframes per rank: [(first, count)] * world."""
    base, rem = divmod(n_superframes, world)
    out, first = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((first, cnt))
        first += cnt
    return out


def gather_ts(ts, nbytes, cap, dst=0, group=None):
    """Gather variable-length decoded TS byte strings on `dst`.

    ts: uint8 tensor of at least `cap` elements on this rank (only the first nbytes are valid).
    Fixed-stride padded buffers + one count vector, so it is a single gather of `cap` bytes per rank.
    Returns (list of per-rank uint8 tensors trimmed to their counts) on dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = ts.device
    cnt = torch.tensor([int(nbytes)], dtype=torch.int64, device=dev)
    if rank == dst:
        cnts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(cnt, cnts, dst=dst, group=group)
        dist.gather(ts[:cap], bufs, dst=dst, group=group)
        return [b[:int(c.item())] for b, c in zip(bufs, cnts)]
    dist.gather(cnt, None, dst=dst, group=group)
    dist.gather(ts[:cap], None, dst=dst, group=group)
    return None
