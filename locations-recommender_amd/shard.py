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


def _spread_stride(n_batches):
    """A stride near 0.38 * n_batches that is coprime with n_batches (so that multiplying by it is a
    bijection modulo n_batches)."""
    import math
    s = max(1, int(round(n_batches * 0.381966)))
    while math.gcd(s, n_batches) != 1:
        s += 1
    return s


def query_batch_of(step, rank, world, n_batches):
    """Index of the query batch rank `rank` processes at step `step`: ranks never overlap within
    a step (world <= n_batches) and together visit every batch once per n_batches / world steps.
    Rows are sorted by length, so batch b is the b-th length quantile of the persons: consecutive
    work items are spread over the quantiles (a golden-ratio stride) instead of walking them in
    order, and a short run of steps samples short and long queries alike."""
    return ((step * world + rank) * _spread_stride(n_batches)) % n_batches


def all_gather_ragged(local, device, world, keep_on_device=False):
    """All-gather of a 1-D array whose length differs per rank (padded to the longest).  `local` is a
    numpy array or a tensor; with keep_on_device the result is a tensor on `device` (the RCCL path:
    the gathered shards go straight into locrec_knn_create_from_device), else a numpy array."""
    t_local = local if torch.is_tensor(local) else torch.from_numpy(np.ascontiguousarray(local))
    n = torch.tensor([t_local.numel()], device=device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    pad = max(max(sizes), 1)
    t = torch.zeros(pad, device=device, dtype=t_local.dtype)
    t[:t_local.numel()] = t_local.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    if keep_on_device:
        return torch.cat([o[:sizes[r]] for r, o in enumerate(out)]).contiguous()
    return np.concatenate([o[:sizes[r]].cpu().numpy() for r, o in enumerate(out)])


def gather_knn_dataset(shard, device, world, keep_on_device=False):
    """Rebuild the full CSR input from per-rank shards (one all-gather per array).  keep_on_device (a CUDA
    `device`, RCCL): the result stays in HBM as tensors, ready for KnnIndex.from_device - no host hop."""
    full = {"p_dim": shard["p_dim"], "c_dim": shard["c_dim"]}
    full["person_ids"] = all_gather_ragged(shard["person_ids"], device, world, keep_on_device)
    for fam in ("p", "c"):
        nnz = np.diff(shard[f"{fam}_rowptr"]).astype(np.int64)
        nnz_all = all_gather_ragged(nnz, device, world, keep_on_device)
        if keep_on_device:
            full[f"{fam}_rowptr"] = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(nnz_all, 0)])
        else:
            full[f"{fam}_rowptr"] = np.concatenate([[0], np.cumsum(nnz_all)]).astype(np.int64)
        full[f"{fam}_idx"] = all_gather_ragged(shard[f"{fam}_idx"].astype(np.int32), device, world, keep_on_device)
        full[f"{fam}_val"] = all_gather_ragged(shard[f"{fam}_val"], device, world, keep_on_device)
    return full


class ShardedSgRecommender:
    """StochasticRecommender (StochasticRecommender.scala:66-106) over a graph whose rows of P are
    sharded across the ranks of a process group (BASELINE.json configs[4]).

    Every rank passes the SAME edge list; the library keeps this rank's rows.  Per iteration the only
    exchange is one all-reduce(sum) of the live entries of sigma = P^T x (float64[live]); with the
    "nccl" backend it is RCCL on torch's current stream, the stream the kernels are enqueued on, so
    the iteration never leaves the device except for isConverged's 8-byte read-back.

    The all-reduce adds the shards' partial sums in RCCL's order, not in the single-GPU kernel's
    fixed order: probabilities agree with the unsharded path to rounding (tests: rtol 1e-9), not
    bit for bit, and an epsilon within rounding of a sweep's |dx|^2 may stop one sweep apart."""

    def __init__(self, source_ids, target_ids, balanced_weights, rank=None, world=None, group=None,
                 always_reduce=False, exchange="all_reduce"):
        """exchange = "all_reduce": rows of P (sources) sharded, sigma summed over the ranks (the wording
        of BASELINE.json configs[4]).  exchange = "all_gather": rows of P^T (targets) sharded, every
        rank owns the live rows l = rank (mod world) and holds all of their inbound edges; the ranks
        all-gather their owned entries of sigma - half the bytes, no cross-rank summation, so the
        result is bit-identical to the single-GPU one (build-order item 2)."""
        from .stochastic import SgGraph
        if exchange not in ("all_reduce", "all_gather"):
            raise ValueError(exchange)
        self.exchange = exchange
        self.group = group
        self.always_reduce = always_reduce  # run the collective even in a group of one (rehearsal)
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.graph = SgGraph(source_ids, target_ids, balanced_weights, self.rank, self.world,
                             by_target=exchange == "all_gather")
        self.device = torch.device("cuda", torch.cuda.current_device())
        live = self.graph.live_count()
        self.sigma = torch.zeros(max(1, live), dtype=torch.float64, device=self.device)
        if exchange == "all_gather":
            # owned entries l = rank + world * j, padded to equal chunks; position of l in the gathered buffer
            self.chunk = max(1, (live + self.world - 1) // self.world)
            own = self.rank + self.world * torch.arange(self.chunk, device=self.device)
            self.own_idx = own.clamp(max=max(0, live - 1))
            ls = torch.arange(max(1, live), device=self.device)
            self.inv_idx = (ls % self.world) * self.chunk + ls // self.world
            self.gathered = torch.zeros(self.world * self.chunk, dtype=torch.float64, device=self.device)
        self._collective = self.world > 1 or (always_reduce and dist.is_initialized())
        self._host_collective = self._collective and dist.get_backend(group) != "nccl"

    def _all_reduce_sigma(self):
        if self.exchange == "all_gather":
            return self._all_gather_sigma()
        if not self._collective:
            return
        if self._host_collective:  # gloo rehearsal: through host memory
            h = self.sigma.cpu()
            dist.all_reduce(h, group=self.group)
            self.sigma.copy_(h)
        else:
            dist.all_reduce(self.sigma, group=self.group)

    def _all_gather_sigma(self):
        packed = self.sigma.index_select(0, self.own_idx)
        if not self._collective:
            self.gathered[:self.chunk] = packed
        elif self._host_collective:  # gloo rehearsal: through host memory
            parts = [torch.empty(self.chunk, dtype=torch.float64) for _ in range(self.world)]
            dist.all_gather(parts, packed.cpu(), group=self.group)
            self.gathered.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(self.gathered, packed, group=self.group)
        torch.index_select(self.gathered, 0, self.inv_idx, out=self.sigma)

    def _one_sweep(self, alpha):
        self.graph.shard_sigma(self.sigma.data_ptr())
        self._all_reduce_sigma()
        self.graph.shard_apply(self.sigma.data_ptr(), alpha)

    def sweeps(self, vertex_id, alpha, n_sweeps):
        """n_sweeps applications of calcNextX with no convergence read-back (fixed work, for timing)."""
        g = self.graph
        g.set_stream(torch.cuda.current_stream().cuda_stream)
        g.shard_begin(vertex_id)
        for _ in range(n_sweeps):
            self._one_sweep(alpha)
        g.shard_finish(n_sweeps, False)

    def recommend(self, vertex_id, alpha, epsilon, max_iterations):
        """-> (ids, probabilities, iterations, converged), the same on every rank."""
        g = self.graph
        g.set_stream(torch.cuda.current_stream().cuda_stream)
        g.shard_begin(vertex_id)
        eps2 = float(epsilon) * float(epsilon)
        it, converged = 0, False
        while it < max_iterations:          # step(), :92-106
            self._one_sweep(alpha)
            if g.shard_d2() <= eps2:        # isConverged, :130-141 (identical on every rank: same sigma)
                converged = True
                break
            it += 1
        g.shard_finish(it, converged)
        return g.fetch()

    def close(self):
        self.graph.close()


def merge_local_topk(lists, k):
    """Merge per-shard (ids, similarities) lists into the k nearest by (similarity desc, person_id
    asc) - the order findSimilarPersons' orderBy(desc).limit(k) takes with the project's tie rule
    (SURVEY.md H1)."""
    ids = np.concatenate([np.asarray(a, np.int64) for a, _ in lists]) if lists else np.empty(0, np.int64)
    sims = np.concatenate([np.asarray(b, np.float64) for _, b in lists]) if lists else np.empty(0, np.float64)
    order = np.lexsort((ids, -sims))[:k]
    return ids[order], sims[order]


class ShardedKnnRequest:
    """One KnnRecommender request with the candidate scan split over the ranks of a process group
    ("KNN single request, latency mode").  Every rank holds the whole index; rank r
    scans candidate shard r, the local lists (K * 16 B) are all-gathered and merged identically on
    every rank, and makeRecommendations0 runs on the merged list."""

    def __init__(self, index, rank=None, world=None, group=None):
        self.index, self.group = index, group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world

    def find_similar(self, person_id, pw, cw, k):
        ids, sims = self.index.query_shard(person_id, pw, cw, k, self.rank, self.world)
        if self.world == 1:
            return ids, sims
        keff = int(min(k, max(1, self.index.n - 1)))
        # one int64 message per rank: count, ids, the similarities' bit patterns (exact either way)
        buf = torch.zeros(2 * keff + 1, dtype=torch.int64)
        buf[0] = len(ids)
        buf[1:1 + len(ids)] = torch.from_numpy(np.ascontiguousarray(ids, np.int64))
        buf[1 + keff:1 + keff + len(ids)] = torch.from_numpy(np.ascontiguousarray(sims, np.float64).view(np.int64))
        if dist.get_backend(self.group) == "nccl":
            buf = buf.cuda()
        got = [torch.empty_like(buf) for _ in range(self.world)]
        dist.all_gather(got, buf, group=self.group)
        lists = []
        for t in got:
            t = t.cpu().numpy()
            c = int(t[0])
            lists.append((t[1:1 + c].copy(), t[1 + keff:1 + keff + c].copy().view(np.float64)))
        return merge_local_topk(lists, keff)

    def recommend(self, person_id, pw, cw, k):
        ids, sims = self.find_similar(person_id, pw, cw, k)
        return self.index.recommend_neighbours(ids, sims)
