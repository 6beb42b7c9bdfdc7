"""The committed PMC record must belong to the committed kernel sources: bench.py reports
roofline.frac / traffic only from a record whose hash matches (a stale record is reported as null), so
shipping one that does not match would silently blank the roofline of the driver's bench run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_record_matches_the_kernel_sources():
    sys.path.insert(0, ROOT)
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc.json")))
    assert rec["source_hash"] == bench.kernel_source_hash(), \
        "kernel sources changed since the rocprofv3 --pmc passes: re-run tools/gpu/r3_profile.sh and copy its r03_* files"
    assert rec["hash_covers"] == list(bench.PMC_SOURCES)
    for key in ("knn_scan", "sg_sweep", "knn_scan1"):
        assert rec[key]["insts_valu"] > 0 and rec[key]["fetch_kib"] > 0
    mix = rec["knn_scan"]["valu_mix"]   # the class split the roofline's peak is derived from
    assert abs(sum(mix["counts"].values()) - rec["knn_scan"]["insts_valu"]) <= 0.02 * rec["knn_scan"]["insts_valu"]
    assert 2.0 <= mix["mix_cycles_per_instruction"] <= 6.0
    k = rec["knn_scan"]["workload"]
    assert (k["persons"], k["places"], k["batch"], k["k"]) == (1_000_000, 100_000, 16_384, 50)   # BASELINE.json configs[1]


def test_bench_reads_the_record():
    sys.path.insert(0, ROOT)
    import bench
    r = bench.pmc_record("knn_scan", persons=1_000_000, places=100_000, batch=16_384, k=50)
    assert r is not None and r["hbm_bytes"] > 0 and r["insts_valu"] > 0
    assert bench.pmc_record("knn_scan", persons=123, places=100_000, batch=16_384, k=50) is None   # another workload: not reused
