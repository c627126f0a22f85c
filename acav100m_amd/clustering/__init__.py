from .sgd_clustering import KMeans  # noqa: F401
