"""Rehearsal of the row-sharded SG path's RCCL call on a one-GPU box: a process group of ONE rank
with backend "nccl" (= RCCL), so that dist.all_reduce really runs on the device tensor and on
torch's current stream, between the library's kernels.  Checks the result against the unsharded
handle.  Run: python tools/rccl_one_rank.py"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pkg = graft.load_package()
from locations_recommender_amd import shard, synth  # noqa: E402

g = synth.sg_dataset(seed=0x5EED0003)
src, dst, w = g["source_id"], g["target_id"], g["balanced_weight"]
v = int(g["first_person"])
rec = shard.ShardedSgRecommender(src, dst, w, 0, 1, always_reduce=True)
whole = pkg.SgGraph(src, dst, w)
for n in (100, 100, 100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec.sweeps(v, 0.15, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} sweeps with a 1-rank RCCL all-reduce of {rec.graph.live_count() * 8} B each: {dt / n * 1e6:.1f} us/sweep", flush=True)
_, ps, _, _ = rec.graph.fetch()
whole.sweeps_async(v, 0.15, 100)
_, pw, _, _ = whole.fetch()
print("max rel diff vs unsharded:", float(np.max(np.abs(ps - pw) / pw)), flush=True)
assert np.allclose(ps, pw, rtol=1e-9, atol=0)
ids, probs, it, conv = rec.recommend(v, 0.15, 1e-6, 1000)
wi, wp, wit, wconv = whole.recommend(v, 0.15, 1e-6, 1000)
print("recommend:", it, conv, "unsharded:", wit, wconv, flush=True)
assert (it, conv) == (wit, wconv) and np.array_equal(ids, wi) and np.allclose(probs, wp, rtol=1e-9, atol=0)
# the all-gather form: dist.all_gather_into_tensor through RCCL, bit-identical to the unsharded handle
rec2 = shard.ShardedSgRecommender(src, dst, w, 0, 1, always_reduce=True, exchange="all_gather")
torch.cuda.synchronize()
t0 = time.perf_counter()
rec2.sweeps(v, 0.15, 100)
torch.cuda.synchronize()
print(f"all-gather form, 100 sweeps with a 1-rank RCCL all-gather each: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us/sweep", flush=True)
_, pg, _, _ = rec2.graph.fetch()
assert np.array_equal(pg, pw), "all-gather form is not bit-identical"
rec2.close()
# the KNN set-up of bench.py at N > 1: every array all-gathered through RCCL, the gathered tensors stay in
# HBM and go into locrec_knn_create_from_device (here with one rank; the gather code path is the same)
dev = torch.device("cuda", 0)
first, rows = shard.person_shard(60_000, 0, 1)
part = synth.knn_dataset(60_000, 20_000, 0x5EED0002, first_row=first, rows=rows)
full = shard.gather_knn_dataset(part, dev, 1, keep_on_device=True)
assert all(torch.is_tensor(full[k]) and full[k].is_cuda for k in ("person_ids", "p_rowptr", "p_idx", "p_val", "c_idx"))
torch.cuda.synchronize()
ixd = pkg.KnnIndex.from_device(full["person_ids"], full["p_rowptr"], full["p_idx"], full["p_val"], full["p_dim"],
                               full["c_rowptr"], full["c_idx"], full["c_val"], full["c_dim"])
ixh = pkg.KnnIndex(part["person_ids"], part["p_rowptr"], part["p_idx"], part["p_val"], part["p_dim"],
                   part["c_rowptr"], part["c_idx"], part["c_val"], part["c_dim"])
a = ixd.query_batch(part["person_ids"][:2000], 0.5, 0.5, 50)
b = ixh.query_batch(part["person_ids"][:2000], 0.5, 0.5, 50)
assert all(np.array_equal(x, y) for x, y in zip(a, b)), "device-gathered index differs from the host-built one"
print("KNN device gather -> create_from_device: identical to the host-built index", flush=True)
ixd.close()
ixh.close()
print("RCCL_ONE_RANK_OK", flush=True)
rec.close()
whole.close()
dist.destroy_process_group()
