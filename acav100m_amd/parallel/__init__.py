"""Multi-GPU plumbing for the hot path: one process per GPU, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" in the CPU tests).  Only the exchanges the path really has:

  k-means training   add(): one all-gather per SGD step of the rank-local rows and their labels
                     (replaces the reference's all_gather([batch]) + all_reduce([counts]) +
                     all_reduce([deltas]) -- sgd_clustering.py:97,115,126 -- with b*d*4 bytes
                     instead of K*d*4, and makes every rank apply the bit-identical update);
                     train_epoch_dp(): the same steps with the rows all-gathered in BULK ahead of the
                     (latency-bound) SGD chain -- no collective on the step path
                     train_epoch_view_parallel(): the CLI's multi-GPU mode -- the independent clusterings
                     (views / layers) are dealt out over the ranks, each trained at single-GPU speed with the
                     single-process arithmetic, owners broadcast their state once per epoch
  k-means assign     none: shards are strided rank::world (mps/distributed.py:439)
  MI selection       none: chunks are independent (chunk.py:21-53)
"""
from .collectives import gather_rows_and_labels, shard_slice, world  # noqa: F401
from .kmeans_dp import (average_state, broadcast_state, broadcast_states, distributed_add,  # noqa: F401
                        train_epoch_dp, train_epoch_view_parallel)
