import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
from locations_recommender_amd import synth
d = synth.knn_dataset(1_000_000, 100_000, seed=0x5EED0002)
ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"], d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
pid = int(ix.row_person_ids(5000, 1)[0])
ix.recommend(pid, 0.5, 0.5, 50)
t0 = time.perf_counter()
for i in range(20):
    ix.recommend(int(ix.row_person_ids(5000 + 37 * i, 1)[0]), 0.5, 0.5, 50)
print("recommend avg ms", (time.perf_counter() - t0) / 20 * 1e3)
