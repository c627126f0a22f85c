"""Where the parallel pkl shard loader spends its time (host side only; no GPU needed).
usage: python tools/exp/loader_probe.py [shards [workers]]"""
import glob
import os
import shutil
import sys
import time
from collections import OrderedDict
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed_worker(job):
    import numpy as np
    from multiprocessing import shared_memory
    from acav100m_amd import shards as io
    path, base, n_expect, views = job
    t0 = time.perf_counter()
    with open(path, 'rb') as f:
        raw = f.read()
    t1 = time.perf_counter()
    import pickle
    rows = pickle.loads(raw)
    t2 = time.perf_counter()
    columns = io._shard_columns_from_rows(rows, Path(path).stem)
    t3 = time.perf_counter()
    for view, d, name, total in views:
        mat = columns['views'][tuple(view)]
        shm = shared_memory.SharedMemory(name=name)
        try:
            np.ndarray((total, d), np.float32, buffer=shm.buf)[base:base + len(mat)] = mat
        finally:
            shm.close()
    t4 = time.perf_counter()
    return (t1 - t0, t2 - t1, t3 - t2, t4 - t3, os.getpid())


def main():
    import numpy as np
    import bench_streamed as bs
    from acav100m_amd import shards as io
    nsh = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    root = '/tmp/acav_loader_probe'
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(root)
    bs.write_shards(root, nsh * 1000, 1024, 16, rows_per_shard=1000)
    paths = sorted(glob.glob(root + '/features/*.pkl'))
    er = {Path(p).stem: 1000 for p in paths}
    ev = OrderedDict([(('audio', 'audio_model', 'layer_0'), 1024), (('video', 'visual_model', 'layer_0'), 1024)])
    # serial, in process
    t = time.perf_counter()
    for p in paths[:20]:
        io._shard_columns_from_rows(io.load_pickle(p), 'x')
    dt = time.perf_counter() - t
    print('serial in-process: %.1f ms per 1000-row shard (2 views) = %.0f rows/s' % (dt / 20 * 1e3, 20000 / dt), flush=True)
    for native in ('1', '0'):
        os.environ['ACAV_SHARD_NATIVE'] = native
        for rep in range(3):
            t = time.perf_counter()
            tab = io.load_feature_shards(paths, sidecar='off', workers=workers, expect_rows=er, expect_views=ev)
            dt = time.perf_counter() - t
            print('product loader (%s), %d workers: %.2f s = %.0f rows/s (shared memory: %s)' % (
                'native reader, acav_pkl_load_group' if native == '1' else 'pickle.load, worker processes', workers, dt, len(tab) / dt, hasattr(tab, '_shm')), flush=True)
            del tab
    os.environ['ACAV_SHARD_NATIVE'] = '1'
    os.environ['ACAV_SHARD_TIMING'] = '1'
    for w in (1, 8, 16, 32, 64):  # (the library clamps to 8 .. 64 threads unless workers says more)
        os.environ['ACAV_SHARD_THREADS'] = str(w)
        for rep in range(2):
            tab = io._load_native(paths, er, ev, w)
            del tab
    os.environ.pop('ACAV_SHARD_TIMING')
    os.environ.pop('ACAV_SHARD_THREADS')
    # the same pool, instrumented workers
    total = nsh * 1000
    shms = [io._SHM.acquire(total * 1024 * 4) for _ in ev]
    views = [(tuple(v), int(d), s.name, total) for (v, d), s in zip(ev.items(), shms)]
    jobs = [(p, i * 1000, 1000, views) for i, p in enumerate(paths)]
    pool = io._pool(workers)
    for cs in (max(1, len(jobs) // (workers * 4)), 1):
        t = time.perf_counter()
        res = list(pool.map(timed_worker, jobs, chunksize=cs))
        dt = time.perf_counter() - t
        a = np.array([r[:4] for r in res])
        print('instrumented map (chunksize %d): %.2f s = %.0f rows/s; per shard mean ms: read %.1f unpickle %.1f columns %.1f shm copy %.1f; '
              'distinct workers %d' % (cs, dt, total / dt, *(a.mean(0) * 1e3), len({r[4] for r in res})), flush=True)
    io._SHM.release(shms)


if __name__ == '__main__':
    main()
