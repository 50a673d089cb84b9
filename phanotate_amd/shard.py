"""Per-contig sharding across the GPUs of one node.

Contigs are independent units (phanotate.py:40,56), so the multi-GPU path is: every rank takes a
subset of contigs, runs the whole path on its own GPU, and rank 0 re-assembles the per-contig
results in input order.  No collective touches the data path; torch.distributed (RCCL on GPUs, gloo
in the CPU tests) is only used to bring the small result lists back.
"""
import heapq


def partition(lengths, world):
    """Greedy longest-first assignment of contigs to `world` ranks, balanced by total bases.

    Deterministic: ties broken by contig index, then by rank.  Returns a list of index lists."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    heap = [(0, r) for r in range(world)]
    heapq.heapify(heap)
    out = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + int(lengths[i]), r))
    for r in range(world):
        out[r].sort()
    return out


def run_sharded(seqs, annotate, rank=0, world=1, dist=None):
    """annotate(list of sequences) -> list of per-contig results.  Returns the full list in input
    order on rank 0 (None elsewhere).  `dist` is torch.distributed (initialised) when world > 1."""
    if world == 1:
        return annotate(seqs)
    parts = partition([len(s) for s in seqs], world)
    mine = parts[rank]
    local = annotate([seqs[i] for i in mine])
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(list(zip(mine, local)), gathered, dst=0)
    if rank != 0:
        return None
    out = [None] * len(seqs)
    for part in gathered:
        for i, r in part:
            out[i] = r
    return out
