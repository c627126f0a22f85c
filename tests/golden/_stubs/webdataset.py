"""Stand-in for the two webdataset names the clustering loader touches (data/clustering.py:61-65,
run_clustering.py:155).  ResizedDataset semantics assumed: yield `length` samples per
epoch, restarting the source when it is exhausted; the golden shards hold a multiple of the batch size
so the wrap-around never triggers."""


from torch.utils.data import IterableDataset


class MultiDataset:
    pass


class ResizedDataset(IterableDataset):
    def __init__(self, dataset, length=None, nominal=None):
        self.dataset, self.length, self.nominal = dataset, length, nominal
        self.source = None

    def __len__(self):
        return self.nominal

    def __iter__(self):
        if self.source is None:
            self.source = iter(self.dataset)
        for _ in range(self.length):
            try:
                sample = next(self.source)
            except StopIteration:
                self.source = iter(self.dataset)
                sample = next(self.source)
            yield sample
