"""Worker of tests/test_gpu_sg.py::test_sharded_two_ranks_process_group: one rank of a row-sharded
SG request and of a candidate-sharded KNN request.  The ranks share the test box's single GPU, so the process group is gloo (RCCL refuses
two ranks on one device) and ShardedSgRecommender stages sigma through host memory; everything
else - shard construction from rank/world, the per-iteration protocol, step()'s loop - is the code
the RCCL path runs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    graft.load_package()
    from locations_recommender_amd import shard, synth
    g = synth.sg_dataset(n_persons=4000, n_places=400, seed=31)
    src, dst, w = g["source_id"], g["target_id"], g["balanced_weight"]
    rec = shard.ShardedSgRecommender(src, dst, w)
    assert (rec.rank, rec.world) == (rank, world)
    vertex = int(g["first_person"]) + 11
    ids, probs, it, conv = rec.recommend(vertex, 0.15, 1e-4, 500)
    # every rank must hold the same answer
    mine = torch.from_numpy(probs.copy())
    ref = mine.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(mine, ref), "ranks disagree"
    if rank == 0:
        import oracle_binding as ob
        oi, op, oit, oconv = ob.sg_recommend(src, dst, w, vertex, 0.15, 1e-4, 500)
        assert np.array_equal(ids, oi) and (it, conv) == (oit, oconv), (it, conv, oit, oconv)
        np.testing.assert_allclose(probs, op, rtol=1e-6, atol=0)
        rec.sweeps(vertex, 0.15, 7)
    else:
        rec.sweeps(vertex, 0.15, 7)
    _, p7, it7, _ = rec.graph.fetch()
    if rank == 0:
        import oracle_binding as ob
        _, o7, _, _ = ob.sg_recommend(src, dst, w, vertex, 0.15, 0.0, 7)
        np.testing.assert_allclose(p7, o7, rtol=1e-6, atol=0)
        print("SHARDED_OK", it, conv, flush=True)
    rec.close()
    # the all-gather form (rows of P^T sharded): bit-identical to the single-GPU handle
    rec = shard.ShardedSgRecommender(src, dst, w, exchange="all_gather")
    ids2, probs2, it2, conv2 = rec.recommend(vertex, 0.15, 1e-4, 500)
    pkg0 = graft.load_package()
    whole = pkg0.SgGraph(src, dst, w)
    wi, wp, wit, wconv = whole.recommend(vertex, 0.15, 1e-4, 500)
    assert np.array_equal(ids2, wi) and (it2, conv2) == (wit, wconv) and np.array_equal(probs2, wp), "all-gather form"
    whole.close()
    rec.close()
    if rank == 0:
        print("SHARDED_AG_OK", flush=True)

    # ---- KNN: one request, candidate scan split over the ranks ----
    d = synth.knn_dataset(30_000, 2_000, seed=78)
    d["r_rowptr"], d["r_place"] = d["p_rowptr"].copy(), d["p_idx"].astype(np.int64)
    d["r_rating"] = (1 + d["p_idx"] % 5).astype(np.int64)
    pkg = graft.load_package()
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    req = shard.ShardedKnnRequest(ix)
    for row in (5, 29_999):
        pid = int(d["person_ids"][row])
        ids, sims = req.find_similar(pid, 0.5, 0.5, 50)
        places, est = req.recommend(pid, 0.5, 0.5, 50)
        uid, usim = ix.query(pid, 0.5, 0.5, 50)
        uplaces, uest = ix.recommend(pid, 0.5, 0.5, 50)
        assert np.array_equal(ids, uid) and np.array_equal(sims, usim), "sharded request differs from the unsharded one"
        assert np.array_equal(places, uplaces) and np.array_equal(est, uest)
        if rank == 0:
            import oracle_binding as ob
            oi, osim, oc = ob.knn_similar_batch(d, np.array([row]), 0.5, 0.5, 50)
            assert np.array_equal(ids, oi[0][:oc[0]]) and np.array_equal(sims, osim[0][:oc[0]])
    ix.close()
    if rank == 0:
        print("SHARDED_KNN_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
