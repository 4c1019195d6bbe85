"""Multi-GPU batch mode: independent scan pairs are sharded across ranks by pair id (one process
per GPU); the only communication is the gather of the 64-byte result records
(loamx_reg_result) — RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.
There is no data-path collective: a registration never needs another rank's data.
"""
import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 64


def shard_range(total_pairs, world_size, rank):
    """Contiguous block of pair ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total_pairs, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_results(local_records, total_pairs=None):
    """all-gathers per-rank result records (uint8 tensor of n_local * 64 bytes, on the rank's
    device) into pair-id order. Ranks may own different numbers of pairs."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_records
    world, rank = dist.get_world_size(), dist.get_rank()
    assert local_records.dtype == torch.uint8 and local_records.numel() % RECORD_BYTES == 0
    n_local = local_records.numel() // RECORD_BYTES
    if total_pairs is None:
        counts = torch.tensor([n_local], dtype=torch.int64, device=local_records.device)
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts)
        sizes = [int(c.item()) for c in all_counts]
    else:
        sizes = [shard_range(total_pairs, world, r)[1] - shard_range(total_pairs, world, r)[0] for r in range(world)]
    assert sizes[rank] == n_local
    if len(set(sizes)) == 1:
        out = torch.empty(world * n_local * RECORD_BYTES, dtype=torch.uint8, device=local_records.device)
        dist.all_gather_into_tensor(out, local_records.contiguous())
        return out
    pad = max(sizes) * RECORD_BYTES
    padded = torch.zeros(pad, dtype=torch.uint8, device=local_records.device)
    padded[: local_records.numel()] = local_records
    parts = [torch.empty(pad, dtype=torch.uint8, device=local_records.device) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([p[: s * RECORD_BYTES] for p, s in zip(parts, sizes)])


def records_to_numpy(records, dtype):
    return records.cpu().numpy().view(dtype)
