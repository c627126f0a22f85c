"""measures.get_measure(name) -- subset_selection/code/measures/__init__.py:5-14.

'batch_mi' is the pipeline default (config.py:45); 'mi' and 'mem_mi' are the reference's exact-greedy measures
(SURVEY.md 8(f) rank 2).  'ami' (adjusted MI, mi.py:212-260) is not built.
"""
from .batch import EfficientBatchMI
from .mi import EfficientMI, EfficientMemMI


def get_measure(measure_name):
    dt = {'mi': EfficientMI, 'mem_mi': EfficientMemMI, 'batch_mi': EfficientBatchMI}
    measure_name = measure_name.lower()
    assert measure_name in dt, "no measure named {}".format(measure_name)
    return dt[measure_name]
