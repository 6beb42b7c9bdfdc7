"""Seeded inputs of the f-2 / f-4 producers, shared by the CPU (oracle vs host mirror) and the GPU
(device vs oracle) tests."""
import numpy as np


def visits_case(seed, n, persons=40, entities=30, base_person=2040, base_entity=40, negative_ids=False):
    """(person_id, entity_id) visit rows with many repeats, so counts tie often."""
    rng = np.random.default_rng(seed)
    p = base_person + rng.integers(0, persons, n)
    # a skewed entity choice: small ids are visited much more often (ties AND distinct counts occur)
    e = base_entity + np.minimum(rng.geometric(0.15, n) - 1, entities - 1)
    if negative_ids:
        p = p - base_person - persons // 2
    return p.astype(np.int64), e.astype(np.int64)


def ratings_case(seed, n, persons=25, entities=60):
    """(person_id, entity_id, rating) rows WITH duplicate (person, entity) pairs of different ratings."""
    rng = np.random.default_rng(seed)
    p = 1000 + rng.integers(0, persons, n)
    e = rng.integers(0, entities, n)
    r = rng.integers(1, 400, n)
    return p.astype(np.int64), e.astype(np.int64), r.astype(np.int64)


METERS_PER_DEGREE = 6371000.0 * np.pi / 180.0


def join_case(seed, n_places=300, n_visits=2000, where="moscow", lat_span=0.01):
    """Places scattered over three regions and visits placed 0 .. 250 m from some place (so both
    outcomes of `<= 100 m` are frequent), plus far-away visits, old visits and a visit in a region
    that has no places."""
    rng = np.random.default_rng(seed)
    centre = {"moscow": (55.75, 37.62), "equator": (0.0005, -78.5), "antimeridian": (-16.5, 179.9995),
              "antimeridian_west": (64.2, -179.9996), "north_pole": (89.9993, 12.0), "south_pole": (-89.9996, -140.0)}[where]
    lon_span = lat_span / max(np.cos(np.radians(centre[0])), 1e-3) if abs(centre[0]) < 89 else 360.0
    p_region = rng.integers(0, 3, n_places) * 7 - 5      # regions -5, 2, 9
    p_lat = np.clip(centre[0] + (rng.random(n_places) - 0.5) * lat_span, -90, 90)
    p_lon = centre[1] + (rng.random(n_places) - 0.5) * lon_span
    p_lon = (p_lon + 180.0) % 360.0 - 180.0
    places = {"id": 40 + np.arange(n_places, dtype=np.int64), "latitude": p_lat, "longitude": p_lon,
              "region_id": p_region.astype(np.int64), "category_id": rng.integers(0, 20, n_places).astype(np.int64)}
    near = rng.integers(0, n_places, n_visits)
    dist = rng.random(n_visits) * 250.0
    ang = rng.random(n_visits) * 2 * np.pi
    v_lat = p_lat[near] + dist * np.cos(ang) / METERS_PER_DEGREE
    coslat = np.maximum(np.cos(np.radians(p_lat[near])), 1e-6)
    v_lon = p_lon[near] + dist * np.sin(ang) / (METERS_PER_DEGREE * coslat)
    over = np.abs(v_lat) > 90                                  # walked over the pole: come down the other side
    v_lat = np.where(over, np.sign(v_lat) * 180.0 - v_lat, v_lat)
    v_lon = np.where(over, v_lon + 180.0, v_lon)
    v_lon = (v_lon + 180.0) % 360.0 - 180.0
    v_region = p_region[near].copy()
    flip = rng.random(n_visits) < 0.1
    v_region[flip] = rng.integers(0, 3, flip.sum()) * 7 - 5    # same spot, maybe another region
    v_region[rng.random(n_visits) < 0.02] = 1234               # a region without places
    ts = 1_600_000_000_000 + rng.integers(0, 90 * 86_400_000, n_visits)
    visits = {"person_id": 2040 + rng.integers(0, 50, n_visits).astype(np.int64), "timestamp": ts.astype(np.int64),
              "latitude": v_lat, "longitude": v_lon, "region_id": v_region.astype(np.int64)}
    visits_from = int(ts.max() - 60 * 86_400_000)               # lastDaysCount = 60 in a UTC session
    return visits, places, visits_from
