"""acav100m_amd -- MI355X-native implementation of ACAV100M's curation hot path.

Only the hot path of SURVEY.md section 8 lives here: the SGD k-means of clustering/code and the greedy
batch-MI subset selection of subset_selection/code, behind the reference's own class
interfaces (KMeans, EfficientBatchMI / get_measure), computed by hand-written HIP kernels in
libacav_hip.so (C ABI: include/acav_hip.h).
"""
from ._lib import AcavError, LIB_PATH, device_count, load_library  # noqa: F401
from .rng import Generator, default_generator, manual_seed  # noqa: F401

__all__ = ["AcavError", "LIB_PATH", "device_count", "load_library", "Generator", "default_generator",
           "manual_seed"]
