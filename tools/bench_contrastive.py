"""Contrastive baseline: ms per 128-clip training batch and clips/s of inference at the real widths
(2304-d visual, 128-d audio -> 128-d), for the record in DESIGN.md.  argv: [batches] [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.subset_selection.measures.contrastive import Contrastive

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rs = np.random.RandomState(0)
n = nb * B
visual = torch.from_numpy(rs.randn(n, 2304).astype(np.float32)).cuda()
audio = torch.from_numpy(rs.randn(n, 128).astype(np.float32)).cuda()
off = np.arange(0, n + 1, B, dtype=np.int64)
acav100m_amd.manual_seed(0)
m = Contrastive(1, "cuda:0", 2e-4, 1)
m.train_batches(visual, audio, off[:3], 1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
m.train_batches(visual, audio, off, 1e-4)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("training: %.3f ms per %d-clip batch (%d batches, %.0f clips/s)" % (dt / nb * 1e3, B, nb, n / dt))
m.infer_scores(visual[:1024], audio[:1024])
t0 = time.perf_counter()
m.infer_scores(visual, audio)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("inference: %.0f clips/s (%d clips in %.2f ms)" % (n / dt, n, dt * 1e3))
