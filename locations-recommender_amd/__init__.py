"""MI355X-native hot paths of tashoyan/locations-recommender behind the reference's
operator surface: KnnRecommender and StochasticRecommender.

The directory name has a hyphen, so load it through __graft_entry__.load_package()
(module name `locations_recommender_amd`)."""
from ._lib import IllegalArgumentException, LocrecRuntimeError, LIB_PATH, lib  # noqa: F401
from .knn import KnnIndex, KnnRecommender, SparseVector  # noqa: F401
from .stochastic import ALPHA, SgGraph, SgGroup, StochasticRecommender  # noqa: F401
from .multi import KnnReplicas, SgSharded, set_devices  # noqa: F401
from . import prep  # noqa: F401,E402  (calc_ratings, calc_rating_vectors, build_with_balanced_weights, calc_place_visits)
