"""Measure registry: the names the reference's `get_measure` knows (subset_selection/code/measures/__init__.py:5-14).

'batch_mi' is the pipeline default (config.py:45); 'mi' and 'mem_mi' are the exact-greedy measures (SURVEY.md 8(f)
rank 2) and 'ami' their adjusted-MI variant (mi.py:212-259), 'contrastive' the baseline selector;
'nmi' / 'constant' are the reference's EfficientNMI / ConstantMeasure (mi.py:262-281), which its registry leaves out.
"""
from .batch import EfficientBatchMI
from .contrastive import Contrastive
from .mi import ConstantMeasure, EfficientAMI, EfficientMI, EfficientMemMI, EfficientNMI

_REGISTRY = {
    'batch_mi': EfficientBatchMI,
    'mi': EfficientMI,
    'mem_mi': EfficientMemMI,
    'ami': EfficientAMI,
    'nmi': EfficientNMI,            # classes of the reference (mi.py:262-281) that its own registry does not name
    'constant': ConstantMeasure,
    'contrastive': Contrastive,
}


def get_measure(measure_name):
    key = measure_name.lower()
    assert key in _REGISTRY, "no measure named {}".format(measure_name)
    return _REGISTRY[key]
