#!/usr/bin/env python3
"""Timings of the f-2 / f-4 producers on the device (device tensors in, device tensors out)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
import prep_cases  # noqa: E402

prep = pkg.prep


def timed(label, fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label}: {dt * 1e3:.2f} ms", flush=True)
    return out


rng = np.random.default_rng(1)
n = 25_000_000   # cfg2-sized: 1 M persons x 25 places
person = torch.as_tensor(2040 + rng.integers(0, 1_000_000, n)).cuda()
place = torch.as_tensor(40 + np.minimum(rng.geometric(0.0005, n) - 1, 99_999)).cuda()
pp, pe, pr = timed(f"calc_ratings, {n} visits (device tensors)", lambda: prep.calc_ratings(person, place, 100))
print("  rating rows", len(pp))
timed(f"calc_rating_vectors, {len(pp)} ratings", lambda: prep.calc_rating_vectors(pp, pe, pr))
hp, he = person.cpu().numpy(), place.cpu().numpy()
timed("calc_ratings, same visits from host arrays (PCIe both ways)", lambda: prep.calc_ratings(hp, he, 100), reps=1)

visits, places, visits_from = prep_cases.join_case(9, 100_000, 2_000_000, "moscow", lat_span=0.3)   # ~5 places within 100 m of a visit
dv = {k: torch.as_tensor(v).cuda() for k, v in visits.items()}
dp = {k: torch.as_tensor(v).cuda() for k, v in places.items()}
out = timed("calc_place_visits, 2 M visits x 100 k places in 3 regions (device tensors)",
            lambda: prep.calc_place_visits(dv, dp, visits_from))
print("  place visits", len(out["place_id"]), "; the cross join would test", len(visits["person_id"]) * len(places["id"]) // 3, "pairs")
