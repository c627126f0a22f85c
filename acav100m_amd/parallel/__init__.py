"""Multi-GPU plumbing for the hot path: one process per GPU, torch.distributed (backend "nccl" is RCCL over xGMI on ROCm;
"gloo" in the CPU tests) and the library's own RCCL communicator (rccl_comm.py).  Only the exchanges the path really has:

  k-means training   The SGD chain of one clustering is latency-bound (a step needs the centres of the previous one), so
                     nothing is exchanged per step: the ROWS of 1 024 future steps go in bulk to the one rank that runs a
                     clustering's chain (the clusterings' chains run on different ranks at the same time), which hands out
                     its state afterwards.  Which rows form a step's global batch is a plan (row_plan.py):
                       `clustering.multi_gpu = views` (the CLI's default; bench.py --gpus N) -- the one-GPU run's batch
                         stream and epoch count, the clusterings dealt out over the ranks: the N-GPU run writes the files of
                         the one-GPU run.  CLI: every training rank reads all shards (train_epoch_view_parallel, no exchange
                         at all); bench.py: the rows stay partitioned and travel by plan_views.
                       `reference` -- the reference's own N-GPU run (per-rank batch int(batch_size / N) of a rotated stream
                         over ALL shards, ceil(epochs / N) epochs: data/clustering.py:25, mps/distributed.py:433-437,
                         run_clustering.py:146), as one process fed the same global batches.
                       `rows` -- a large-batch operating point (global batch N x batch_size over each rank's own shards,
                         ceil(epochs / N) epochs): N x N fewer SGD steps than `reference`; NOT the reference's run.
                     distributed_add(): one add() step with an all-gather of the local rows and labels (the per-step form,
                     kept for KMeans.add() under is_distributed)
  k-means assign     none: shards are strided rank::world (mps/distributed.py:439)
  MI selection       none: chunks are independent (chunk.py:21-53)
"""
from .collectives import gather_rows_and_labels, shard_slice, world  # noqa: F401
from .kmeans_dp import (average_state, broadcast_state, broadcast_states, distributed_add,  # noqa: F401
                        plan_warmup_labels, train_epoch_dp, train_epoch_plan, train_epoch_view_parallel)
from .row_plan import RowPlan, make_plan  # noqa: F401
