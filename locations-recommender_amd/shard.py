"""Multi-GPU sharding of the two hot paths, kept free of any GPU call so that the
N > 1 logic is testable with gloo on CPUs.

KNN: every rank holds the full candidate set and owns a share of the QUERIES -- independent units,
no collective in the data path.  The candidate set itself is assembled at set-up from per-rank
shards with one all-gather per array (RCCL over xGMI when the tensors live on GPUs).
SG:  whole graphs are independent units (one per rank)."""
import numpy as np
import torch
import torch.distributed as dist


def person_shard(n_persons, rank, world):
    """[first, first + rows) of the persons rank `rank` generates/uploads."""
    per = (n_persons + world - 1) // world
    first = min(rank * per, n_persons)
    return first, min(per, n_persons - first)


def query_batch_of(step, rank, world, n_batches):
    """Index of the query batch rank `rank` processes at step `step`: ranks never overlap within
    a step and together sweep the batches round-robin."""
    return (step * world + rank) % n_batches


def all_gather_ragged(local, device, world):
    """All-gather of a 1-D numpy array whose length differs per rank (padded to the longest)."""
    t_local = torch.from_numpy(np.ascontiguousarray(local))
    n = torch.tensor([t_local.numel()], device=device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    pad = max(max(sizes), 1)
    t = torch.zeros(pad, device=device, dtype=t_local.dtype)
    t[:t_local.numel()] = t_local.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o[:sizes[r]].cpu().numpy() for r, o in enumerate(out)])


def gather_knn_dataset(shard, device, world):
    """Rebuild the full CSR input from per-rank shards (one all-gather per array)."""
    full = {"p_dim": shard["p_dim"], "c_dim": shard["c_dim"]}
    full["person_ids"] = all_gather_ragged(shard["person_ids"], device, world)
    for fam in ("p", "c"):
        nnz = np.diff(shard[f"{fam}_rowptr"]).astype(np.int64)
        nnz_all = all_gather_ragged(nnz, device, world)
        full[f"{fam}_rowptr"] = np.concatenate([[0], np.cumsum(nnz_all)]).astype(np.int64)
        full[f"{fam}_idx"] = all_gather_ragged(shard[f"{fam}_idx"].astype(np.int32), device, world)
        full[f"{fam}_val"] = all_gather_ragged(shard[f"{fam}_val"], device, world)
    return full
