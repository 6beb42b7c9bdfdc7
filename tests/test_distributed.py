"""world_size-2 gloo tests of the multi-GPU sharding logic (no GPU: the N > 1 path of bench.py
differs from N = 1 only in what these functions decide)."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from locations_recommender_amd import shard, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 3001  # not divisible by the world size on purpose
    first, rows = shard.person_shard(n, rank, world)
    part = synth.knn_dataset(n, 700, seed=0x5EED0002, first_row=first, rows=rows)
    full = shard.gather_knn_dataset(part, "cpu", world)
    ref = synth.knn_dataset(n, 700, seed=0x5EED0002)
    for k in ("person_ids", "p_rowptr", "p_idx", "p_val", "c_rowptr", "c_idx", "c_val"):
        assert np.array_equal(full[k], ref[k]), k
    batches = [shard.query_batch_of(s, rank, world, 7) for s in range(14)]
    np.save(os.path.join(out_dir, f"batches_{rank}.npy"), np.array(batches))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_query_sharding(tmp_path):
    import socket
    with socket.socket() as sock:  # a free port for the rendezvous
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    b = [np.load(tmp_path / f"batches_{r}.npy") for r in range(world)]
    assert np.all(b[0] != b[1]), "two ranks took the same query batch in one step"
    assert set(np.concatenate(b).tolist()) == set(range(7)), "some batch is never processed"


def test_person_shards_tile_the_range():
    import __graft_entry__ as g
    g.load_package()
    from locations_recommender_amd import shard
    for n, world in ((10, 3), (1_000_000, 8), (7, 8), (64, 2)):
        spans = [shard.person_shard(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(rows for _, rows in spans) == n
        for (f0, r0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + r0 == f1


def test_merge_of_local_lists_keeps_the_tie_rule():
    import __graft_entry__ as g
    g.load_package()
    from locations_recommender_amd import shard
    lists = [(np.array([9, 4]), np.array([0.9, 0.5])), (np.array([], np.int64), np.array([])),
             (np.array([7, 2, 3]), np.array([0.9, 0.5, 0.1]))]
    ids, sims = shard.merge_local_topk(lists, 4)
    assert ids.tolist() == [7, 9, 2, 4] and sims.tolist() == [0.9, 0.9, 0.5, 0.5]
    ids, sims = shard.merge_local_topk(lists, 10)
    assert ids.tolist() == [7, 9, 2, 4, 3]
    ids, sims = shard.merge_local_topk([], 3)
    assert len(ids) == 0 and len(sims) == 0


def test_query_batches_are_spread_and_disjoint():
    import __graft_entry__ as g
    g.load_package()
    from locations_recommender_amd import shard
    for nb, world in ((61, 1), (61, 8), (64, 8), (7, 2), (1, 1), (610, 4)):
        seen = []
        for step in range((nb + world - 1) // world):
            now = [shard.query_batch_of(step, r, world, nb) for r in range(world)]
            assert len(set(now)) == min(world, nb) or nb < world, (nb, world, step)
            seen += now
        assert set(seen) == set(range(nb)), (nb, world)
    # six consecutive steps of one rank reach into every third of the length spectrum
    six = [shard.query_batch_of(s, 0, 1, 61) for s in range(6)]
    assert min(six) < 20 and max(six) > 40 and any(20 <= b <= 40 for b in six)
