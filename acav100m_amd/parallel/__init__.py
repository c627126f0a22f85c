"""Multi-GPU plumbing for the hot path: one process per GPU, torch.distributed (backend "nccl" is RCCL over xGMI on ROCm;
"gloo" in the CPU tests) and the library's own RCCL communicator (rccl_comm.py).  Only the exchanges the path really has:

  k-means training   `clustering.multi_gpu = rows` and bench.py --gpus N -- the reference's DDP (sgd_clustering.py:94-129):
                     every rank holds its own shards' rows, a step's global batch is batch_size rows of every rank.  The
                     SGD chain is latency-bound, so nothing is exchanged per step: the ROWS of 1 024 future steps go in
                     bulk to the one rank that runs a view's chain (acav_kmeans_train_dp_multi / train_epoch_dp; the views'
                     chains run on different ranks at the same time), which hands out its state afterwards.
                     `clustering.multi_gpu = views` (the CLI's default) -- train_epoch_view_parallel(): the independent
                     clusterings (views / layers) are dealt out over the ranks, each trained over all rows with the one-GPU
                     arithmetic and epoch count; owners broadcast their state once per epoch.
                     distributed_add(): one add() step with an all-gather of the local rows and labels (the per-step form,
                     kept for KMeans.add() under is_distributed)
  k-means assign     none: shards are strided rank::world (mps/distributed.py:439)
  MI selection       none: chunks are independent (chunk.py:21-53)
"""
from .collectives import gather_rows_and_labels, shard_slice, world  # noqa: F401
from .kmeans_dp import (average_state, broadcast_state, broadcast_states, distributed_add,  # noqa: F401
                        train_epoch_dp, train_epoch_view_parallel)
