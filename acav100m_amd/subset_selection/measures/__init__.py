"""measures.get_measure(name) -- subset_selection/code/measures/__init__.py:5-14.

Only the default measure of the pipeline ('batch_mi', config.py:45) is on the hot path; the
other names of the reference ('mi', 'ami', 'mem_mi') are listed in SURVEY.md 8(f) as "next".
"""
from .batch import EfficientBatchMI


def get_measure(measure_name):
    dt = {'batch_mi': EfficientBatchMI}
    measure_name = measure_name.lower()
    assert measure_name in dt, "no measure named {}".format(measure_name)
    return dt[measure_name]
