"""acav100m_amd -- MI355X-native implementation of ACAV100M's curation hot path.

Only the hot path of SURVEY.md section 8 lives here: the SGD k-means of clustering/code and the greedy
batch-MI subset selection of subset_selection/code, behind the reference's own class
interfaces (KMeans, EfficientBatchMI / get_measure), computed by hand-written HIP kernels in
libacav_hip.so (C ABI: include/acav_hip.h).
"""
import os as _os

# The reference trains every clustering of a batch stream per batch (ten in its real pipeline: 5 + 5 layers); here each is a
# persistent launch on its own stream, and launches only overlap across HARDWARE queues.  The HIP runtime maps a process's
# streams onto GPU_MAX_HW_QUEUES queues (default 4): with ten clusterings, the fifth launch waited for one of the first four
# to END -- the ten trained in three rounds (29.6 us per step of the ten instead of 14.9, tools/bench_train_real10.py).
# The runtime reads the variable when it initialises (the first device call), so importing this package before touching
# the GPU is enough; an explicit setting of the caller wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from ._lib import AcavError, LIB_PATH, device_count, load_library  # noqa: F401,E402
from .rng import Generator, default_generator, manual_seed  # noqa: F401,E402

__all__ = ["AcavError", "LIB_PATH", "device_count", "load_library", "Generator", "default_generator",
           "manual_seed"]
