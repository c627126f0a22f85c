"""Stand-in for torch_scatter 2.0.5 (un-vendored reference dependency), used ONLY by
gen_golden.py to import the reference's KMeans in the build container."""


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    import torch
    if src.dim() == 1:
        if out is None:
            out = torch.zeros(dim_size, dtype=src.dtype)
        return out.scatter_add_(0, index, src)
    assert dim == 0
    if out is None:
        out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    # row-sequential accumulation, the order torch_scatter's CPU loop uses
    for i in range(src.shape[0]):
        out[index[i]] += src[i]
    return out
