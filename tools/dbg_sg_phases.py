import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["LOCREC_SG_DEBUG_PHASES"] = "1"
import __graft_entry__ as graft
pkg = graft.load_package()
from locations_recommender_amd import synth
g = synth.sg_dataset()
sg = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
v = int(g["first_person"])
for _ in range(3):
    sg.sweeps_async(v, 0.15, 100)
    sg.fetch()
