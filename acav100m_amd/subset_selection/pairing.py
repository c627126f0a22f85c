"""Clustering-pair selection -- subset_selection/code/pairing.py:5-41.

keys: the sorted (model_key, layer) tuples of dataloader.format_assignments (dataloader.py:43-53);
returns index pairs into the D columns of the assignment matrix.
"""
import itertools
from collections import OrderedDict


def _group_indices(keys, field):
    groups = OrderedDict()
    for idx, key in enumerate(keys):
        groups.setdefault(key[field], []).append(idx)
    return list(groups.values())


def get_combination(keys):
    """every unordered pair of clusterings, audio-audio included (pairing.py:16-20)"""
    return list(itertools.combinations(range(len(keys)), 2))


def get_bipartite(keys):
    """one clustering from every model_key group (pairing.py:23-30)"""
    return list(itertools.product(*_group_indices(keys, 0)))


def get_diagonal(keys):
    """clusterings that share a layer name (pairing.py:33-41)"""
    return _group_indices(keys, 1)


_PAIRINGS = {'diagonal': get_diagonal, 'bipartite': get_bipartite, 'combination': get_combination}


def get_cluster_pairing(keys, cluster_pairing):
    cluster_pairing = cluster_pairing.lower()
    assert cluster_pairing in _PAIRINGS, f"invalid cluster pairing type: {cluster_pairing}"
    return _PAIRINGS[cluster_pairing](keys)
