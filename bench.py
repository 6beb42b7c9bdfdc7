#!/usr/bin/env python3
"""Benchmark of the two hot paths on MI355X (contract: see the task statement / DESIGN.md).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Headline (BASELINE.json metric, config "1M persons x 100k places, K=50, KNN cosine + top-K on 1
MI355X"): KNN person-pair cosines per second.  One step = one batch of --batch queries against
ALL persons (cosine over place + category vectors, combine, per-query top-K), with the index
resident in HBM and the results left in HBM.  Ranks hold the full candidate set (built from
per-rank shards exchanged with one RCCL all-gather at set-up) and shard the QUERIES: no
collective in the timed region, weak scaling.
The same JSON line carries the SG figure (config "~5M edges, 100 power iterations") under "sg".
"""
import argparse
import json
import os
import sys
import time

# the cpu_baseline legs use OpenMP (libgomp is initialised when torch is imported): with the policy
# left unset, short parallel regions separated by barriers - the SG sweeps - ran slower on 8 threads
# than on one; an explicit policy restores the scaling.  Nothing on the GPU path uses OpenMP.
os.environ.setdefault("OMP_WAIT_POLICY", "active")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue: 256 CUs x 4 SIMDs x 2.4 GHz SIMD-cycles per second.  How many of them one wave64 instruction takes depends
# on its class (tools/ubench_valu.hip, profiles/r03_ubench_valu.log: plain 32-bit ops ~3, packed-16 / SDWA / dot2 /
# alignbit ~4.6, fp64 ~5.9), so the peak in instructions per second is derived from the measured CLASS MIX of the kernel
# (rocprofv3 class counters, profiles/r03_valu_classes.txt) - not asserted here.
SIMD_GCYCLES = 256 * 4 * 2.4
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc.json")


# the sources of the kernels the PMC record is about (knn_scan_ht, knn_scan1*, sg_sweep) and everything they include
PMC_SOURCES = ("common.h", "knn_index.h", "knn_device.h", "knn_ht.h", "knn_rowscan.h", "knn_single.h", "knn_merge.h", "knn_side.h",
               "knn_aggregate.h", "knn.hip", "sg.hip")


def kernel_source_hash():
    """Hash of the measured kernels' sources: a PMC record taken on other sources is stale and not reported."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "locations-recommender_amd", "csrc")
    for name in PMC_SOURCES:
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_record(kernel, **workload):
    """Counters of one launch from the committed rocprofv3 --pmc passes (profiles/r03_pmc.json:
    separate passes for SQ_INSTS_VALU, the VALU class counters, FETCH_SIZE, WRITE_SIZE; tools/gpu/r3_profile.sh), or None
    when they were taken on other kernel sources or another workload than this run."""
    try:
        with open(PMC_FILE) as f:
            rec = json.load(f)
        if rec.get("source_hash") != kernel_source_hash():
            return None
        t = rec[kernel]
    except (OSError, KeyError, ValueError):
        return None
    if any(t["workload"].get(k) != v for k, v in workload.items()):
        return None
    out = dict(t)
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE is in KiB and counts the 128-B requests of wide
    # streaming reads as 64 B on gfx950 (x2); WRITE_SIZE is exact
    if "fetch_kib" in t and "write_kib" in t:
        out["hbm_bytes"] = (2 * t["fetch_kib"] + t["write_kib"]) * 1024
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--persons", type=int, default=1_000_000)
    ap.add_argument("--places", type=int, default=100_000)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--batch", type=int, default=16_384, help="queries per step per GPU")
    ap.add_argument("--sg-sweeps", type=int, default=100)
    ap.add_argument("--no-aggregate", action="store_true",
                    help="KNN step = neighbours only (skip makeRecommendations0's aggregation kernel)")
    ap.add_argument("--no-sg", action="store_true")
    ap.add_argument("--sg-graphs", type=int, default=8,
                    help="graphs per GPU of the batched SG leg (cfg5: 64 graphs over 8 GPUs); 0 = skip")
    ap.add_argument("--sg-sharded", choices=["auto", "on", "off"], default="auto",
                    help="also time ONE graph row-sharded over the ranks (auto: when --gpus > 1)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (and the oracle sample check)")
    ap.add_argument("--no-formats", action="store_true", help="skip the PACK32 / GENERIC secondary legs")
    ap.add_argument("--cpu-queries", type=int, default=0, help="0 = 2 per core")
    return ap.parse_args()


def build_knn_input(args, rank, world, device):
    from locations_recommender_amd import shard, synth
    seed = 0x5EED0002
    if world == 1:
        return synth.knn_dataset(args.persons, args.places, seed)
    # each rank generates its shard of persons; one RCCL all-gather per array rebuilds the full set
    first, rows = shard.person_shard(args.persons, rank, world)
    part = synth.knn_dataset(args.persons, args.places, seed, first_row=first, rows=rows)
    # RCCL: the gathered arrays stay in HBM and go straight into locrec_knn_create_from_device
    return shard.gather_knn_dataset(part, device, world, keep_on_device=device.type == "cuda")


def cpu_baseline_knn(d, args):
    """The oracle (a scalar port of the reference's per-request scan), OpenMP over queries, on a
    bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    avail = len(os.sched_getaffinity(0))
    # the port allocates and sorts a candidate list per query, as the reference does; beyond a few dozen
    # threads that stops scaling on a many-core host (measured: 35 M pairs/s at 32 threads, 17 M at 256),
    # so the thread count is probed and the best one is used and reported as "cores"
    best, best_rate = 1, 0.0
    for th in sorted({min(8, avail), min(32, avail), min(64, avail)}):
        rows = np.linspace(0, args.persons - 1, th).astype(np.int64)
        t0 = time.perf_counter()
        ob.knn_similar_batch(d, rows, 0.5, 0.5, args.k, nthreads=th)
        rate = th * (args.persons - 1) / (time.perf_counter() - t0)
        if rate > best_rate:
            best, best_rate = th, rate
    nq = args.cpu_queries or 8 * best
    rows = np.linspace(0, args.persons - 1, nq).astype(np.int64)
    t0 = time.perf_counter()
    ob.knn_similar_batch(d, rows, 0.5, 0.5, args.k, nthreads=best)
    dt = time.perf_counter() - t0
    return {"value": nq * (args.persons - 1) / dt, "unit": "person-pair cosines/s", "cores": best, "kind": "port",
            "note": "the port scores and then fully sorts the ~1M candidates of every query, as the reference's "
                    "orderBy does: a stated baseline, not a tuned CPU kernel",
            "sample": f"{nq} queries x {args.persons} candidates (oracle/locrec_oracle.c, OpenMP over queries; "
                      f"best of 8/32/64 threads on {avail} available), {dt:.1f} s"}


def check_last_step(ix, d, first, nq, k, ids, sims, counts, rec, nsample=16):
    """16 sampled queries of the last timed step against the oracle: neighbour ids and similarities
    bit-exact, estimated ratings within 1e-6 (tests/test_gpu_configs.py does the same in -m gpu)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    sample = np.unique(np.linspace(0, nq - 1, nsample).astype(np.int64))
    qids = ix.row_person_ids(first, nq)
    qrows = (qids[sample] - int(d["person_ids"][0])).astype(np.int64)  # synthetic person ids are contiguous
    assert np.array_equal(d["person_ids"][qrows], qids[sample])
    oi, os_, oc = ob.knn_similar_batch(d, qrows, 0.5, 0.5, k, nthreads=min(16, len(os.sched_getaffinity(0))))
    for j, q in enumerate(sample):
        c = int(counts[q])
        assert c == int(oc[j]) and np.array_equal(ids[q, :c], oi[j, :c]), f"neighbours of query {q} differ from the oracle"
        assert np.array_equal(sims[q, :c], os_[j, :c]), f"similarities of query {q} differ from the oracle bit-wise"
    nrec = 0
    if rec is not None:
        off, places, est = rec
        for q in sample[::4]:
            op, oe = ob.knn_recommend(d, int(qids[q]), 0.5, 0.5, k)
            assert np.array_equal(places[off[q]:off[q + 1]], op), f"recommended places of query {q} differ"
            assert np.allclose(est[off[q]:off[q + 1]], oe, rtol=1e-6, atol=0), f"ratings of query {q} differ"
            nrec += 1
    return {"queries": int(len(sample)), "with_ratings": nrec, "ids_and_similarities": "bit-exact", "ratings_rtol": 1e-6}


def secondary_format_leg(pkg, d, args, env, label):
    """The same step on an index forced into another stored format (PACK32: u32 dots; GENERIC: the
    reference's own fp64 values and int32 indices) - the headline depends on PACK16 being legal."""
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        t0 = time.perf_counter()
        ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                          d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
        create_s = time.perf_counter() - t0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    n = ix.info()["n"]
    batch = min(args.batch, n)
    nb = max(1, n // batch)
    from locations_recommender_amd import shard
    ix.recommend_range_async(shard.query_batch_of(0, 0, 1, nb) * batch, batch, 0.5, 0.5, args.k)
    ix.synchronize()
    steps = 2
    t0 = time.perf_counter()
    for i in range(steps):
        ix.recommend_range_async(shard.query_batch_of(1 + i, 0, 1, nb) * batch, batch, 0.5, 0.5, args.k)
    ix.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"format": label, "mode": ix.info()["mode"], "kernel": ix.scan_kernel_name(), "ms_per_step": dt * 1e3,
           "value": batch * (n - 1) / dt, "unit": "person-pair cosines/s", "create_s": create_s,
           "bytes_per_pair": ix.info()["batch_scan_bytes"] / n}
    ix.close()
    return out


def cpu_baseline_sg(g, v, args):
    """The oracle's CSR sweep (same arithmetic order as the reference's scalar loop), OpenMP over
    target rows, on the same graph and request."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    avail = len(os.sched_getaffinity(0))
    # 10k rows carry all 5M edges: more threads than that parallelism feeds only add barrier cost, so
    # pick the thread count that is fastest on a 5-sweep probe and report THAT count as "cores"
    best, best_rate = 1, 0.0
    for th in sorted({1, min(8, avail), min(32, avail), min(96, avail)}):
        _, _, dt = ob.sg_sweeps_csr(g["source_id"], g["target_id"], g["balanced_weight"], v, 0.15, 5, nthreads=th)
        if 5 / dt > best_rate:
            best, best_rate = th, 5 / dt
    sweeps = min(args.sg_sweeps, 200)
    _, _, dt = ob.sg_sweeps_csr(g["source_id"], g["target_id"], g["balanced_weight"], v, 0.15, sweeps, nthreads=best)
    return {"value": sweeps / dt, "unit": "iterations/s", "cores": best, "kind": "port",
            "sample": f"{sweeps} sweeps of the same graph (oracle/locrec_oracle.c oracle_sg_sweeps_csr, OpenMP over rows; "
                      f"best of 1/8/32/96 threads on {avail} available), {dt:.2f} s"}


def sg_batched(args, pkg, rank, world, barrier, max_over_ranks):
    """cfg5's per-GPU share: --sg-graphs independent graphs resident together, each iterated on its own HIP
    stream so that the sweeps of different graphs overlap.  (With fp64 weights streamed 8 graphs sweep 428 MB per
    round - beyond the 256 MiB Infinity Cache: round 2's HBM-honest figure; in the dictionary form they sweep 175 MB
    and fit it: `swept_bytes_fit_infinity_cache` says which case a run was.)"""
    from locations_recommender_amd import synth
    n = args.sg_graphs
    graphs, targets, streams = [], [], []
    sweep_bytes = 0
    for i in range(n):
        gi = rank * n + i
        g = synth.sg_dataset(seed=0x5EED0500 + gi)
        h = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
        st = torch.cuda.Stream()
        h.set_stream(st.cuda_stream)
        graphs.append(h)
        targets.append(int(g["first_person"]))
        streams.append(st)
        sweep_bytes += h.info()["device_sweep_bytes"]

    def run_streams():
        for h, v in zip(graphs, targets):
            h.sweeps_async(v, 0.15, args.sg_sweeps)
        for h in graphs:
            h.synchronize()

    reps = max(1, args.steps)

    def rate(run):
        run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        barrier()
        sdt = max_over_ranks(time.perf_counter() - t0)
        return world * n * reps * args.sg_sweeps / sdt

    # three ways to iterate the same n resident graphs: every graph on its own stream (two launches per graph
    # and round); ONE group (locrec_sg_group_*: two launches per round for all graphs); two groups of n / 2 on
    # two streams, so that one half's combine kernel overlaps the other half's sweep
    forms = {"per_graph_streams": rate(run_streams)}
    group_all = pkg.SgGroup(graphs)

    def run_group():
        group_all.sweeps_async(targets, 0.15, args.sg_sweeps)
        group_all.synchronize()

    forms["one_group"] = rate(run_group)
    group_all.close()
    if n >= 2:
        halves = [pkg.SgGroup(graphs[:n // 2]), pkg.SgGroup(graphs[n // 2:])]
        tv = [targets[:n // 2], targets[n // 2:]]

        def run_halves():
            for grp, t in zip(halves, tv):
                grp.sweeps_async(t, 0.15, args.sg_sweeps)
            for grp in halves:
                grp.synchronize()

        forms["two_groups"] = rate(run_halves)
        for grp in halves:
            grp.close()
    for h in graphs:
        h.close()
    best = max(forms, key=forms.get)
    its = forms[best]
    gbs = its / world * (sweep_bytes / n) / 1e9
    return {"metric": "SG SpMV graph-iterations/s, independent graphs batched", "value": its, "unit": "iterations/s",
            "graphs_per_gpu": n, "scaling": "weak", "resident_bytes_per_gpu": sweep_bytes,
            "swept_bytes_fit_infinity_cache": bool(sweep_bytes < (256 << 20)),
            "achieved_GBps_per_gpu": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "form": best,
            "graph_iterations_per_s_by_form": forms,
            "note": "whole-leg rate x the bytes the device layout moves per sweep (locrec_sg_device_bytes); "
                    "includes every kernel of the iteration and the launch gaps"}


def sg_row_sharded(args, pkg, whole, v, rank, world, barrier, max_over_ranks):
    """One graph (rank 0's) sharded over all ranks, in both forms of SURVEY.md 8e: rows of P sharded
    with an all-reduce(sum) of the live entries of sigma per sweep, and rows of P^T sharded with an
    all-gather of the owned entries (RCCL on the kernels' stream either way)."""
    from locations_recommender_amd import shard, synth
    g = synth.sg_dataset(seed=0x5EED0003)
    out = {"metric": "SG SpMV iterations/s, one graph sharded over the ranks", "unit": "iterations/s",
           "scaling": "strong", "shards": world}
    for exchange in ("all_reduce", "all_gather"):
        rec = shard.ShardedSgRecommender(g["source_id"], g["target_id"], g["balanced_weight"], rank, world,
                                         exchange=exchange)
        live = rec.graph.live_count()
        rec.sweeps(v, 0.15, args.sg_sweeps)
        torch.cuda.synchronize()
        reps = max(1, args.steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            rec.sweeps(v, 0.15, args.sg_sweeps)
        torch.cuda.synchronize()
        barrier()
        sdt = max_over_ranks(time.perf_counter() - t0)
        checked = None
        if rank == 0:  # same request on the unsharded handle of the same graph
            _, ps, _, _ = rec.graph.fetch()
            whole.sweeps_async(v, 0.15, args.sg_sweeps)
            _, pw, _, _ = whole.fetch()
            checked = bool(np.array_equal(ps, pw)) if exchange == "all_gather" else \
                bool(ps.shape == pw.shape and np.allclose(ps, pw, rtol=1e-9, atol=0))
        rec.close()
        out[exchange] = {"value": reps * args.sg_sweeps / sdt, "ms_per_iteration": sdt / (reps * args.sg_sweeps) * 1e3,
                         "exchange_bytes_per_sweep_per_rank": live * 8 if exchange == "all_reduce" else (live + world - 1) // world * 8,
                         "matches_unsharded": checked,
                         "match": "bit-identical" if exchange == "all_gather" else "1e-9 relative"}
    out["value"] = max(out["all_reduce"]["value"], out["all_gather"]["value"])
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot paths have no CPU fallback")
    # one process per GPU.  LOCREC_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer
    # GPUs than ranks (ranks then share devices and the collectives run on host tensors).
    backend = os.environ.get("LOCREC_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], device=coll_device, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    pkg = graft.load_package()
    from locations_recommender_amd import _lib as L
    L.check(L.lib().locrec_set_device(dev_index))

    # ---------------- KNN (headline) ----------------
    d = build_knn_input(args, rank, world, coll_device)
    # placeRatings (KnnRecommender.scala:13): one row per (person, visited place), rating 1..5 derived
    # from the place index - the same on every rank, no exchange needed
    on_device = torch.is_tensor(d["p_idx"])
    d["r_rowptr"] = d["p_rowptr"]
    d["r_place"] = d["p_idx"].to(torch.int64) if on_device else d["p_idx"].astype(np.int64)
    d["r_rating"] = 1 + d["r_place"] % 5
    if on_device:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    make = pkg.KnnIndex.from_device if on_device else pkg.KnnIndex
    ix = make(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
              d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    create_s = time.perf_counter() - t0  # locrec_knn_create[_from_device]: the index is built by kernels (csrc/knn_build.hip)
    info = ix.info()
    n = info["n"]
    batch = min(args.batch, n)
    nbatches = max(1, n // batch)

    from locations_recommender_amd import shard

    def step(i):
        b = shard.query_batch_of(i, rank, world, nbatches)  # ranks never overlap: no collective
        # the whole path for the batch: cosine scan + combine + top-K (a2-a4) AND the similarity-
        # weighted rating aggregation (a5); everything stays in HBM
        if args.no_aggregate:
            ix.topk_range_async(b * batch, batch, 0.5, 0.5, args.k)
        else:
            ix.recommend_range_async(b * batch, batch, 0.5, 0.5, args.k)

    for i in range(args.warmup):
        step(i)
    ix.synchronize()
    ix.profile_enable(True)
    allocs0 = pkg._lib.device_allocations()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    ix.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    allocs_timed = pkg._lib.device_allocations() - allocs0   # (reported: a steady-state step allocates nothing)
    scan_ms, launches = ix.profile_read()
    replayed = ix.replayed_intervals()  # statistics only: handled inside the kernel, nothing is redone later
    plan_name, plan_qt = ix.scan_kernel_name(), ix.query_tile()
    # read the last step back once (outside the timed region): results exist and are complete
    last_ids, last_sims, last_counts = ix.fetch_topk(batch, args.k)
    assert int(last_counts.min()) == min(args.k, n - 1), "a query of the last step has fewer neighbours than K"
    rec_rows, last_rec = None, None
    if not args.no_aggregate:
        last_rec = ix.fetch_recommend(batch)
        roff, _, rest = last_rec
        # a weighted mean of ratings 1..5 (up to rounding of sum(r*s) / sum(s))
        assert np.all(np.diff(roff) > 0), "a query of the last step has no recommendation rows"
        assert rest.min() >= 1.0 - 1e-9 and rest.max() <= 5.0 + 1e-9, (rest.min(), rest.max())
        rec_rows = int(roff[-1])
    # ... and CORRECT: a sample of the last timed step's queries against the oracle (the checker,
    # outside the timed region; rank 0 at N = 1 only, with the cpu_baseline leg)
    # (every CPU leg - this check, the two cpu_baseline figures - runs at the very END of the program: dozens of OpenMP
    # threads on the box's CPU share left the host stalling for tens of milliseconds in the legs that followed them,
    # e.g. one 56 ms step among 21 ms ones in knn_wide_rows, 57 instead of 10.5 ms in knn_large_k_batched)
    oracle_checked = None
    # the timed launches were final: reading results back must not have launched another scan
    _, extra_launches = ix.profile_read()
    ix.profile_enable(False)
    assert launches == args.steps and extra_launches == 0, (launches, extra_launches)
    dt = max_over_ranks(dt)
    pairs = world * args.steps * batch * (n - 1)
    knn_value = pairs / dt
    scan_avg_s = scan_ms / max(1, launches) * 1e-3
    # The batched scan holds a tile of queries in LDS and reads each candidate row once per TILE, so
    # it is bound by vector-instruction issue, not by HBM (PMC: 1.75 TB/s of real traffic at r01).
    # roofline = wave64 VALU instructions per launch (PMC SQ_INSTS_VALU of the same workload and the
    # same sources) / launch duration against the chip's integer issue rate.  The effective
    # bandwidth under SURVEY 8d's per-query streaming model is reported beside it, never as frac.
    algo_bytes = batch * info["batch_scan_bytes"]
    pmc = pmc_record("knn_scan", persons=n, places=args.places, batch=batch, k=args.k)
    insts = pmc.get("insts_valu") if pmc else None
    mix = pmc.get("valu_mix") if pmc else None
    ach = insts / scan_avg_s / 1e9 if insts else None
    traffic = pmc.get("hbm_bytes") if pmc else None
    # peak = the issue rate of THIS kernel's instruction mix: SIMD-cycles per second / measured cycles per instruction
    # of the mix (each class of the SQ's class counters priced by the microbenchmark); frac = the share of all
    # SIMD-cycles of the launch that its VALU instructions occupy.  Beside it the two bounds VERDICT r02 asked for.
    cpi = mix["mix_cycles_per_instruction"] if mix else None
    peak = SIMD_GCYCLES / cpi if cpi else None
    roofline = {"bound": "valu", "kernel": plan_name, "achieved": ach, "peak": peak,
                "unit": "G wave-instr/s", "frac": ach / peak if ach and peak else None,
                "frac_at_4cycle": ach / (SIMD_GCYCLES / 4) if ach else None,
                "frac_at_2cycle": ach / (SIMD_GCYCLES / 2) if ach else None,
                "mix_cycles_per_instruction": cpi,
                "valu_classes": {"instructions": mix["counts"], "cycles_per_instruction": mix["cycles_per_instruction"]} if mix else None,
                "traffic": traffic, "hbm_frac": traffic / scan_avg_s / 1e9 / HBM_PEAK_GBS if traffic else None,
                "query_tile": plan_qt, "avg_launch_ms": scan_avg_s * 1e3,
                "valu_instructions_per_launch": insts,
                "valu_instructions_per_pair": insts / (batch * n) if insts else None,
                "effective_GBps_streaming_model": algo_bytes / scan_avg_s / 1e9,
                "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_pair": info["batch_scan_bytes"] / n,
                "note": "VALU-issue bound: peak = 2457.6 G SIMD-cycles/s / the measured cycles per instruction of the kernel's "
                        "own class mix (profiles/r03_valu_classes.txt: rocprofv3 class counters x tools/ubench_valu.hip issue "
                        "costs); instruction counts and HBM traffic from --pmc passes of this workload on these sources "
                        "(profiles/r03_pmc.json, null when stale); effective_GBps is the per-query streaming model of "
                        "SURVEY 8d and exceeds HBM peak because a tile of 16 queries shares each read"}

    # ---------------- KNN, the reference's own operator: one person per call ----------------
    # (latency figure beside the batched headline; at N > 1 also with the candidate scan split over
    # the ranks, SURVEY.md 8e "latency mode")
    knn_request = None
    try:
        pid = int(ix.row_person_ids(n // 2, 1)[0])
        ix.recommend(pid, 0.5, 0.5, args.k)
        reqs = 20
        lat = []
        for _ in range(reqs):
            t0 = time.perf_counter()
            ix.recommend(pid, 0.5, 0.5, args.k)
            lat.append(time.perf_counter() - t0)
        one = float(np.median(lat))
        ix.query(pid, 0.5, 0.5, args.k)
        t0 = time.perf_counter()
        for _ in range(reqs):
            ix.query(pid, 0.5, 0.5, args.k)
        one_q = (time.perf_counter() - t0) / reqs
        # the single-request scan reads every candidate row exactly once: the per-query streaming model
        # of SURVEY 8d taken literally, so its bandwidth is a plain (not "effective") HBM figure
        ix.profile_enable(True)
        for _ in range(reqs):
            ix.query(pid, 0.5, 0.5, args.k)
        s1_ms, s1_n = ix.profile_read()
        ix.profile_enable(False)
        s1 = s1_ms / max(1, s1_n) * 1e-3
        knn_request = {"metric": "KnnRecommender.makeRecommendations latency, host buffers in and out",
                       "ms_per_request": one * 1e3, "pairs_per_s": (n - 1) / one,
                       "ms_min_max": [min(lat) * 1e3, max(lat) * 1e3], "find_similar_persons_ms": one_q * 1e3,
                       "scan_roofline": {"bound": "hbm",
                                         "kernel": "knn_scan1_direct8 (byte tables; knn_scan1 when they do not fit)",
                                         "avg_launch_ms": s1 * 1e3,
                                         "achieved": info["scan_bytes"] / s1 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": info["scan_bytes"] / s1 / 1e9 / HBM_PEAK_GBS,
                                         "traffic": (pmc_record("knn_scan1", persons=n, places=args.places, k=args.k) or {}).get("hbm_bytes"),
                                         "note": "every candidate row read once per request: real, not effective, bandwidth"}}
        # the SHIPPED parameter (bin/knn_recommender.sh:35: --k-nearest 2000000 = every person with a
        # positive similarity is a neighbour): stream scan -> device radix sort of all candidates /
        # place-major aggregation
        big_k = 2_000_000
        ix.recommend(pid, 0.5, 0.5, big_k)
        lat_big = []
        for _ in range(5):
            t0 = time.perf_counter()
            bp, _be = ix.recommend(pid, 0.5, 0.5, big_k)
            lat_big.append(time.perf_counter() - t0)
        ix.query(pid, 0.5, 0.5, big_k)
        lat_bigq = []
        for _ in range(5):
            t0 = time.perf_counter()
            bi, _bs = ix.query(pid, 0.5, 0.5, big_k)
            lat_bigq.append(time.perf_counter() - t0)
        knn_request["large_k"] = {"k_nearest": big_k, "recommend_ms": float(np.median(lat_big)) * 1e3,
                                  "recommendation_rows": int(len(bp)),
                                  "find_similar_persons_ms": float(np.median(lat_bigq)) * 1e3,
                                  "neighbours_returned": int(len(bi)),
                                  "note": "host buffers in and out; find_similar_persons returns every "
                                          "positive-similarity person (16 B each over PCIe)"}
        # The request as the UNCHANGED main issues it (KnnRecommenderMain.scala:53-67): name the region pair's three
        # Parquet sets, construct a recommender, ask it, forget it.  The constructor takes the device index from the
        # library's process-wide handle cache, keyed by the files (name + size + mtime; here three stand-in files):
        # cold = key + miss + locrec_knn_create from the host arrays + the request; warm = key + hit + the request.
        if rank == 0:
            import tempfile
            from locations_recommender_amd import _cache
            with tempfile.TemporaryDirectory() as td:
                paths = []
                for name in ("place_rating_vectors_region0", "category_rating_vectors_region0", "place_ratings_region0"):
                    paths.append(os.path.join(td, name))
                    with open(paths[-1], "wb") as f:
                        f.write(name.encode())

                def through_constructor():
                    key = _cache.files_key(paths)
                    h = pkg.KnnIndex.through_cache(key, lambda: pkg.KnnIndex(
                        d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"], d["c_rowptr"], d["c_idx"],
                        d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"]))
                    with h.lock:
                        out = h.recommend(pid, 0.5, 0.5, args.k)
                    h.close()   # drops the reference; the reference's main never even does that
                    return out

                if not on_device:
                    st0 = L.cache_stats()
                    t0 = time.perf_counter()
                    cp, ce = through_constructor()
                    cold = time.perf_counter() - t0
                    through_constructor()
                    allocs1 = L.device_allocations()
                    warm = []
                    for _ in range(50):
                        t0 = time.perf_counter()
                        wp, we = through_constructor()
                        warm.append(time.perf_counter() - t0)
                    st1 = L.cache_stats()
                    up, ue = ix.recommend(pid, 0.5, 0.5, args.k)
                    knn_request["through_constructor"] = {
                        "cold_ms": cold * 1e3, "warm_ms": float(np.median(warm)) * 1e3,
                        "warm_over_bare_request": float(np.median(warm)) / one,
                        "creates": st1["misses"] - st0["misses"], "cache_hits": st1["hits"] - st0["hits"],
                        "device_allocations_while_warm": L.device_allocations() - allocs1,
                        "matches_bare_request": bool(np.array_equal(wp, up) and np.array_equal(we, ue) and np.array_equal(cp, up)),
                        "note": "KnnRecommenderMain.makeRecommendations' per-request body through the handle cache "
                                "(include/locrec.h); the second index of the same data is destroyed again below"}
                    _cache.clear()
        if world > 1:
            req = shard.ShardedKnnRequest(ix, rank, world)
            req.recommend(pid, 0.5, 0.5, args.k)
            barrier()
            t0 = time.perf_counter()
            for _ in range(reqs):
                places, est = req.recommend(pid, 0.5, 0.5, args.k)
            sharded = max_over_ranks((time.perf_counter() - t0) / reqs)
            up, ue = ix.recommend(pid, 0.5, 0.5, args.k)
            knn_request["candidate_sharded"] = {"ms_per_request": sharded * 1e3, "shards": world,
                                                "matches_unsharded": bool(np.array_equal(places, up) and np.array_equal(est, ue))}
    except Exception as e:  # the headline line must still be printed
        knn_request = {"error": f"{type(e).__name__}: {e}"}

    # ---------------- SG (second figure of the metric) ----------------
    sg_out = None
    sg_cpu_inputs = None
    if not args.no_sg:
        from locations_recommender_amd import synth
        g = synth.sg_dataset(seed=0x5EED0003 + rank)  # one independent graph per rank (cfg5's natural form)
        t0 = time.perf_counter()
        sg = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
        sg_create_s = time.perf_counter() - t0   # locrec_sg_create from host arrays: vertex ranking, piece layout, upload
        sinfo = sg.info()
        v = int(g["first_person"])
        sg.sweeps_async(v, 0.15, args.sg_sweeps)
        sg.synchronize()
        # timed region WITHOUT per-launch events: at ~10 us per kernel the two event packets per
        # sweep cost as much as a kernel boundary each (18 -> 24 us per iteration when recorded)
        reps = max(1, args.steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            sg.sweeps_async(v, 0.15, args.sg_sweeps)
        sg.synchronize()
        barrier()
        sdt = max_over_ranks(time.perf_counter() - t0)
        # same requests again with HIP events around every sg_sweep launch: the kernel's duration
        sg.profile_enable(True)
        for _ in range(reps):
            sg.sweeps_async(v, 0.15, args.sg_sweeps)
        sg.synchronize()
        sweep_ms, slaunches = sg.profile_read()
        sg.profile_enable(False)
        its = world * reps * args.sg_sweeps / sdt
        sweep_avg_s = sweep_ms / max(1, slaunches) * 1e-3
        # algorithmic bytes = what the DEVICE layout moves per sweep (2-byte columns + fp64 weights of
        # every slot, descriptors, partials, x): the reference-width model of SURVEY 8d (12 B/edge,
        # sweep_bytes) is printed beside it and never priced against the peak
        dev_bytes = sinfo["device_sweep_bytes"]
        sg_ach = dev_bytes / sweep_avg_s / 1e9
        spmc = pmc_record("sg_sweep", edges=sinfo["edges"], vertices=sinfo["vertices"])
        sg_out = {"metric": "SG SpMV iterations/s", "value": its, "unit": "iterations/s",
                  "ms_per_iteration": sdt / (reps * args.sg_sweeps) * 1e3, "create_s": sg_create_s,
                  # one full sweep x -> x' INCLUDING the convergence sum over the whole iteration's
                  # time (every kernel and boundary of it), on the device layout's bytes
                  "iteration_GBps": dev_bytes * its / world / 1e9,
                  "iteration_frac_of_hbm_peak": dev_bytes * its / world / 1e9 / HBM_PEAK_GBS,
                  "iteration_frac_survey_widths": sinfo["sweep_bytes"] * its / world / 1e9 / HBM_PEAK_GBS,
                  "config": {"workload": f"stochastic graph E={sinfo['edges']} V={sinfo['vertices']}, "
                                         f"{args.sg_sweeps} sweeps per request, one graph per GPU",
                             # > 0: the sweep streams a uint16 index per edge into a table of this many distinct
                             # fp64 weights (sg_sweep_dict; bit-identical to the fp64 stream)
                             "weight_dictionary": sinfo["weight_dictionary"]},
                  "roofline": {"bound": "hbm", "kernel": "sg_sweep_dict" if sinfo["weight_dictionary"] else "sg_sweep",
                               "achieved": sg_ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": sg_ach / HBM_PEAK_GBS,
                               "frac_survey_widths": sinfo["sweep_bytes"] / sweep_avg_s / 1e9 / HBM_PEAK_GBS,
                               "traffic": spmc.get("hbm_bytes") if spmc else None,
                               "bytes_per_sweep": dev_bytes, "bytes_per_sweep_survey_widths": sinfo["sweep_bytes"],
                               "avg_launch_ms": sweep_avg_s * 1e3,
                               "note": "one graph fits the 256 MiB Infinity Cache (and so do the 8 of the batched leg in "
                                       "the dictionary form).  The sweep is a chain of latencies, not a stream (~5 us of "
                                       "its ~10 us remain with every load and store removed: "
                                       "profiles/r03_sg_dict_knockouts.log), so moving fewer bytes (the dictionary "
                                       "form) makes it faster AND lowers this fraction"}}
        # makeRecommendations at the shipped parameters (bin/stochastic_recommender.sh: epsilon 0.01, 20 iterations
        # at most): the whole call - set-up, iterations until isConverged, probabilities of all vertices back on
        # the host - for different persons; rank 0 only (a latency, not part of the aggregate)
        if rank == 0:
            try:
                rng_v = np.random.default_rng(7)
                persons = [int(g["first_person"]) + int(x) for x in rng_v.integers(0, 280_000, size=40)]
                sg.recommend(persons[0], 0.15, 0.01, 20)
                lat, its_seen = [], []
                for pv in persons:
                    t0 = time.perf_counter()
                    _ids, _pr, it_n, _conv = sg.recommend(pv, 0.15, 0.01, 20)
                    lat.append(time.perf_counter() - t0)
                    its_seen.append(int(it_n))
                sg_out["request"] = {"metric": "makeRecommendations latency (epsilon 0.01, maxIterations 20)",
                                     "median_ms": float(np.median(lat) * 1e3), "p90_ms": float(np.quantile(lat, 0.9) * 1e3),
                                     "iterations_median": float(np.median(its_seen)), "requests": len(lat),
                                     "includes": "request set-up, iterations with the host's looks at the convergence "
                                                 "word, all V probabilities copied to the host"}
            except Exception as e:  # the headline line must still be printed
                sg_out["request"] = {"error": f"{type(e).__name__}: {e}"}
        if args.sg_graphs > 0:
            sg_out["batched"] = sg_batched(args, pkg, rank, world, barrier, max_over_ranks)
        sg_cpu_inputs = (g, v)   # (the CPU baseline of this leg runs at the end of the program)
        # the same graph with the fp64 weights streamed (round 2's layout, LOCREC_SG_NO_DICT): what the dictionary form
        # buys, and the iteration priced on THAT layout's bytes - the figure VERDICT r02's "0.5 of peak = 13.3 us" is stated on
        if rank == 0 and sinfo["weight_dictionary"]:
            try:
                os.environ["LOCREC_SG_NO_DICT"] = "1"
                sg64 = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
                os.environ.pop("LOCREC_SG_NO_DICT", None)
                b64 = sg64.info()["device_sweep_bytes"]
                sg64.sweeps_async(v, 0.15, args.sg_sweeps)
                sg64.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    sg64.sweeps_async(v, 0.15, args.sg_sweeps)
                sg64.synchronize()
                t64 = (time.perf_counter() - t0) / (reps * args.sg_sweeps)
                sg64.close()
                t_it = sdt / (reps * args.sg_sweeps)
                sg_out["fp64_stream_form"] = {
                    "ms_per_iteration": t64 * 1e3, "bytes_per_sweep": b64,
                    "iteration_frac_of_hbm_peak": b64 / t64 / 1e9 / HBM_PEAK_GBS,
                    "dictionary_iteration_on_these_bytes": b64 / t_it / 1e9 / HBM_PEAK_GBS,
                    "note": "LOCREC_SG_NO_DICT=1: fp64 weights streamed (10 B per edge slot); the last figure prices the "
                            "dictionary form's iteration on this layout's bytes (same results bit for bit, fewer bytes moved)"}
            except Exception as e:  # the headline line must still be printed
                os.environ.pop("LOCREC_SG_NO_DICT", None)
                sg_out["fp64_stream_form"] = {"error": f"{type(e).__name__}: {e}"}
        # ---- one graph, rows of P sharded over the ranks, all-reduce of sigma per sweep (cfg "8xMI355X
        # row-sharded").  Strong scaling of a graph that fits one GPU's cache: reported beside the
        # independent-graphs figure above, never instead of it.
        if args.sg_sharded == "on" or (args.sg_sharded == "auto" and world > 1):
            try:
                sg_out["row_sharded"] = sg_row_sharded(args, pkg, sg, v, rank, world, barrier, max_over_ranks)
            except Exception as e:  # the headline line must still be printed
                sg_out["row_sharded"] = {"error": f"{type(e).__name__}: {e}"}
        sg.close()

    # ---------------- host-inclusive and large-K legs (rank 0) ----------------
    host_incl, large_k_batched = None, None
    if rank == 0 and not on_device:
        try:
            import ctypes as C
            first = shard.query_batch_of(args.warmup, rank, world, nbatches) * batch
            qids = ix.row_person_ids(first, batch)
            # findSimilarPersons for the batch, K ids + similarities per query on the HOST when the call returns
            ix.query_batch(qids[:256], 0.5, 0.5, args.k)
            t0 = time.perf_counter()
            ix.query_batch(qids, 0.5, 0.5, args.k)
            tq = time.perf_counter() - t0
            # makeRecommendationsBatch, rows on the host: one call with room for everything (the two-call sizing
            # protocol would run the batch twice)
            cap = int((rec_rows or batch * 512) * 1.25) + 1024
            off = np.zeros(batch + 1, np.int64)
            places, est = np.empty(cap, np.int64), np.empty(cap, np.float64)
            lat = []
            for _ in range(2):
                c = C.c_int64(cap)
                t0 = time.perf_counter()
                L.check(L.lib().locrec_knn_recommend_batch(ix._h, batch, L.ptr(qids, C.c_int64), 0.5, 0.5, args.k,
                                                           L.ptr(off, C.c_int64), L.ptr(places, C.c_int64),
                                                           L.ptr(est, C.c_double), C.byref(c)))
                lat.append(time.perf_counter() - t0)
                assert c.value <= cap
            tr = min(lat)
            host_incl = {"metric": "the batched step with results on the HOST (PCIe and host reordering inside the time)",
                         "queries": batch,
                         "find_similar_persons": {"ms": tq * 1e3, "value": batch * (n - 1) / tq, "unit": "person-pair cosines/s",
                                                  "bytes_to_host": int(batch * args.k * 16 + batch * 8)},
                         "make_recommendations": {"ms": tr * 1e3, "value": batch * (n - 1) / tr, "unit": "person-pair cosines/s",
                                                  "rows_to_host": int(off[-1]), "bytes_to_host": int(off[-1] * 16)},
                         "device_resident_ms_per_step": dt / args.steps * 1e3}
        except Exception as e:  # the headline line must still be printed
            host_incl = {"error": f"{type(e).__name__}: {e}"}
        try:
            big_k, nqb = 2_000_000, 64
            qrow0 = (n // 2 // 16) * 16
            qids = ix.row_person_ids(qrow0, nqb)
            ix.recommend_range_async(qrow0, nqb, 0.5, 0.5, big_k)   # (sizes the work buffers)
            ix.synchronize()
            tbs = []
            for _ in range(3):
                t0 = time.perf_counter()
                ix.recommend_range_async(qrow0, nqb, 0.5, 0.5, big_k)
                ix.synchronize()
                tbs.append(time.perf_counter() - t0)
            tb = float(np.median(tbs))
            boff, bplaces, best = ix.fetch_recommend(nqb)
            sp, se = ix.recommend(int(qids[5]), 0.5, 0.5, big_k)
            same = bool(np.array_equal(sp, bplaces[boff[5]:boff[6]]) and np.array_equal(se, best[boff[5]:boff[6]]))
            checked = None
            if not args.no_cpu:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_binding as ob
                op, oe = ob.knn_recommend(d, int(qids[17]), 0.5, 0.5, big_k)
                checked = bool(np.array_equal(op, bplaces[boff[17]:boff[18]]) and
                               np.allclose(oe, best[boff[17]:boff[18]], rtol=1e-6, atol=0))
            large_k_batched = {"metric": "makeRecommendations for a batch at the shipped --k-nearest 2000000 "
                                         "(every positive-similarity person is a neighbour)",
                               "k_nearest": big_k, "queries": nqb, "ms": tb * 1e3, "ms_per_query": tb / nqb * 1e3,
                               "value": nqb * (n - 1) / tb, "unit": "person-pair cosines/s",
                               "recommendation_rows": int(boff[-1]), "single_request_ms": (knn_request or {}).get("large_k", {}).get("recommend_ms"),
                               "equals_single_request_bitwise": same, "one_query_matches_oracle": checked,
                               "note": "results device-resident (fetch outside the time); tiles of 16 queries, dense fp64 "
                                       "query tables, no top-K, place-major aggregation shared by the tile"}
        except Exception as e:  # the headline line must still be printed
            large_k_batched = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    # the other two stored formats, so that the headline's dependence on PACK16 is visible
    formats = None
    if rank == 0 and world == 1 and not args.no_formats:
        formats = []
        for env, label in (({"LOCREC_KNN_NO_PACK16": "1"}, "PACK32 (u32 dots)"),
                           ({"LOCREC_KNN_FORCE_GENERIC": "1"}, "GENERIC (fp64 values, the reference's width)")):
            try:
                formats.append(secondary_format_leg(pkg, d, args, env, label))
            except Exception as e:  # the headline line must still be printed
                formats.append({"format": label, "error": f"{type(e).__name__}: {e}"})
    # the same step on an index with 0.1 % WIDE rows (a count of 300: does not fit the head / tail element's byte): the
    # per-row format fallback keeps the head / tail kernel and scores the wide rows from the plain CSR
    wide_leg = None
    if rank == 0 and world == 1 and not args.no_formats and not on_device:
        try:
            rng_w = np.random.default_rng(77)
            wide = np.sort(rng_w.choice(n, max(1, n // 1000), replace=False))
            dv = dict(d)
            v = d["p_val"].copy()
            for r in wide:
                v[d["p_rowptr"][r] + rng_w.integers(0, d["p_rowptr"][r + 1] - d["p_rowptr"][r])] = 300.0
            dv["p_val"] = v
            wix = pkg.KnnIndex(dv["person_ids"], dv["p_rowptr"], dv["p_idx"], dv["p_val"], dv["p_dim"], dv["c_rowptr"], dv["c_idx"],
                               dv["c_val"], dv["c_dim"], dv["r_rowptr"], dv["r_place"], dv["r_rating"])
            wix.recommend_range_async(shard.query_batch_of(0, 0, 1, nbatches) * batch, batch, 0.5, 0.5, args.k)
            wix.synchronize()
            wsteps = []
            for i in range(4):
                t0 = time.perf_counter()
                wix.recommend_range_async(shard.query_batch_of(1 + i, 0, 1, nbatches) * batch, batch, 0.5, 0.5, args.k)
                wix.synchronize()
                wsteps.append(time.perf_counter() - t0)
            wdt = float(np.median(wsteps))
            wide_leg = {"wide_rows": int(len(wide)), "kernel": wix.scan_kernel_name(), "ms_per_step": wdt * 1e3,
                        "ms_steps": [x * 1e3 for x in wsteps],
                        "value": batch * (n - 1) / wdt, "unit": "person-pair cosines/s",
                        "over_clean_step": wdt / (dt / args.steps),
                        "note": "0.1 % of the persons hold a count of 300; without the per-row fallback this index ran the "
                                "PACK32 row scan (knn_other_formats[0])"}
            wix.close()
        except Exception as e:  # the headline line must still be printed
            wide_leg = {"error": f"{type(e).__name__}: {e}"}
    # ---------------- the CPU legs, after every GPU leg ----------------
    if rank == 0 and world == 1 and not args.no_cpu:
        last_b = shard.query_batch_of(args.warmup + args.steps - 1, rank, world, nbatches)
        oracle_checked = check_last_step(ix, d, last_b * batch, batch, args.k, last_ids, last_sims, last_counts, last_rec)
        cpu = cpu_baseline_knn(d, args)
        if sg_out is not None and sg_cpu_inputs is not None:
            sg_out["cpu_baseline"] = cpu_baseline_sg(sg_cpu_inputs[0], sg_cpu_inputs[1], args)
    ix.close()
    import shutil
    spark = "unavailable on this host" if not (shutil.which("spark-submit") and shutil.which("java")) else \
        "present but not run: the reference jar is not part of this repository"

    if rank == 0:
        out = {
            "metric": "KNN person-pair cosines/s (cosine place+category, combine, top-K)",
            "value": knn_value, "unit": "person-pair cosines/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {2: "u16 dot (exact, packed) / f64 cosine", 1: "u32 dot (exact) / f64 cosine"}.get(info["mode"], "f64"),
            "data": "synthetic",
            "config": {"workload": f"KNN {n} persons x {args.places} places, K={args.k}, "
                                   f"{batch} queries/step/GPU vs all persons (BASELINE.json configs[1])",
                       "packed": info["packed"], "seed": "0x5EED0002",
                       "step": "scan + combine + top-K" + ("" if args.no_aggregate else " + rating aggregation"),
                       "recommendation_rows_last_step": rec_rows,
                       "scan_launches": launches, "device_allocations_in_timed_region": allocs_timed, "flush_intervals_replayed_in_kernel": replayed,
                       "create_s": create_s, "checked_against_oracle": oracle_checked},
            "roofline": roofline, "cpu_baseline": cpu, "spark": spark, "knn_request": knn_request,
            "knn_host_inclusive": host_incl, "knn_large_k_batched": large_k_batched,
            "knn_other_formats": formats, "knn_wide_rows": wide_leg, "sg": sg_out,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
