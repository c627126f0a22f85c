"""Greedy MI subset selection (reference: subset_selection/code) -- hot-path pieces only."""
from .measures import get_measure  # noqa: F401
from .pairing import get_cluster_pairing  # noqa: F401
from .run_greedy import _run_greedy, run_greedy  # noqa: F401
