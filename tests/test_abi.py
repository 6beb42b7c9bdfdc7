"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/locrec.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_GPU = torch.cuda.is_available()


def header_functions():
    text = open(os.path.join(ROOT, "include", "locrec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(locrec_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    from locations_recommender_amd import _lib
    names = header_functions()
    assert len(names) >= 25
    handle = C.CDLL(pkg.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/locrec.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "python binding and header disagree"


def test_version_and_error_text(pkg):
    lib = pkg.lib()
    assert lib.locrec_version().decode().startswith("locrec")
    n = C.c_int32(-1)
    lib.locrec_device_count(C.byref(n))
    assert n.value >= 0


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU behaviour")
def test_device_allocation_counter_is_readable_without_a_gpu(pkg):
    """locrec_device_allocations only reads a counter: it works on any host (0 here, where nothing can be
    allocated) and rejects a NULL output."""
    from locations_recommender_amd import _lib
    n = _lib.device_allocations()
    assert n >= 0
    if not HAVE_GPU:
        assert n == 0
    assert pkg.lib().locrec_device_allocations(None) != 0


def test_no_cpu_fallback_without_gpu(pkg):
    """The product path must fail loudly when the device is missing."""
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=20, p_dim=30, seed=1)
    with pytest.raises(pkg.LocrecRuntimeError):
        pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                     d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
    with pytest.raises(pkg.LocrecRuntimeError):
        pkg.SgGraph(np.array([1, 2]), np.array([2, 1]), np.array([1.0, 1.0]))


def test_host_mirror_requires(pkg):
    """Constructor require()s of the two Scala classes are enforced before any device work."""
    import pandas as pd
    empty_v = pd.DataFrame({"person_id": [], "rating_vector": []})
    empty_r = pd.DataFrame({"person_id": [], "place_id": [], "rating": []})
    for pw, cw, k in [(0.0, 1.0, 3), (1.0, 0.0, 3), (0.5, 0.4, 3), (0.5, 0.5, 0), (0.5, 0.5, -1)]:
        with pytest.raises(pkg.IllegalArgumentException):
            pkg.KnnRecommender(empty_v, empty_v, empty_r, pw, cw, k)
    edges = pd.DataFrame({"source_id": [1], "target_id": [2], "balanced_weight": [1.0]})
    with pytest.raises(pkg.IllegalArgumentException):
        pkg.StochasticRecommender(edges, -0.1, 10)
    with pytest.raises(pkg.IllegalArgumentException):
        pkg.StochasticRecommender(edges, 0.1, -1)
