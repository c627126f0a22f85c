"""Contrastive -- drop-in for the reference's baseline selector (subset_selection/code/measures/contrastive/
contrastive.py:56-256 on top of module.py:9-98): an audio-visual InfoNCE model over the penultimate features, trained
with AdamW and used to rank the clips by the cosine of their aligned audio / visual projections.

Same constructor and method surface (init, train, infer, save_cache / load_cache and the cache file names), the model
lives in libacav_hip.so (acav_contrastive_*: forward, backward, AdamW and inference as HIP kernels, no torch op).  The
initial parameters are nn.Linear's, drawn from the torch-stream generator (rng.Generator) in the reference's order, so a
seeded run starts from the reference's weights bit for bit.  Quirk kept: the reference's train loop never zeroes the
gradients (contrastive.py:92-101).  No CPU path.
"""
import csv
import ctypes as C
import math
from pathlib import Path

import numpy as np

from ... import _lib
from ... import shards as io
from ...rng import default_generator

PARAM_NAMES = ('visual_linear.weight', 'visual_linear.bias', 'audio_linear.weight', 'audio_linear.bias')


def lr_func_linear(current_step, num_training_steps, num_warmup_steps=3):
    """contrastive.py:42-45"""
    if current_step < num_warmup_steps:
        return float(current_step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))


def _linear_init(gen, out_features, in_features):
    """nn.Linear.reset_parameters on torch's CPU generator: weight ~ kaiming_uniform_(a=sqrt(5)), then bias, both
    U(-1/sqrt(in), 1/sqrt(in)); at::uniform_real_distribution<float> maps a 24-bit draw in double and rounds."""
    wb = math.sqrt(3.0) * math.sqrt(2.0 / 6.0) / math.sqrt(in_features)
    bb = 1.0 / math.sqrt(in_features)

    def draw(shape, bound):
        lo, hi = np.float32(-bound), np.float32(bound)
        r = gen.rand(*shape).astype(np.float64)
        return (r * np.float64(np.float32(hi - lo)) + np.float64(lo)).astype(np.float32)

    return draw((out_features, in_features), wb), draw((out_features,), bb)


def _device_index(device):
    s = str(device)
    if s == 'cpu':
        raise _lib.AcavError("acav100m_amd Contrastive has no CPU path: use device='cuda'")
    if ':' in s:
        return int(s.split(':')[1])
    try:
        import torch
        return torch.cuda.current_device()
    except Exception:
        return 0


class Contrastive:
    default_sizes = [2304, 128]  # video (slowfast) : 2304, audio (VGGish) : 128  (contrastive.py:76-79)

    def __init__(self, num_epochs=1, device='cuda', base_lr=1e-4, num_warmup_steps=3, distributed=False, sizes=None,
                 out_size=None, generator=None):
        self.num_epochs = num_epochs
        self.device = device
        self.base_lr = base_lr
        self.num_warmup_steps = num_warmup_steps
        self.distributed = distributed
        self.epoch = 0
        gen = generator if generator is not None else default_generator
        vis, aud = sizes if sizes is not None else self.default_sizes
        out = min(vis, aud) if out_size is None else out_size
        self.sizes = (int(vis), int(aud), int(out))
        wv, bv = _linear_init(gen, out, vis)   # module.py:22-23: visual_linear first, then audio_linear
        wa, ba = _linear_init(gen, out, aud)
        self._h = None
        self._create(np.concatenate([wv.ravel(), bv, wa.ravel(), ba]).astype(np.float32))

    def _create(self, flat):
        lib = _lib.load_library()
        if self._h is not None:
            lib.acav_contrastive_destroy(self._h)
            self._h = None
        h = C.c_void_p()
        vis, aud, out = self.sizes
        _lib.check(lib.acav_contrastive_create(C.byref(h), _device_index(self.device), vis, aud, out, _lib.ptr(flat), None))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h and _lib is not None and _lib._lib is not None:
            _lib._lib.acav_contrastive_destroy(h)

    def init(self, clustering_combinations, candidates):
        pass

    # ------------------------------------------------------------------ parameters (state_dict layout)
    def _split(self, flat):
        vis, aud, out = self.sizes
        cuts = np.cumsum([out * vis, out, out * aud, out])
        parts = np.split(flat, cuts[:-1])
        shapes = [(out, vis), (out,), (out, aud), (out,)]
        return {k: p.reshape(s).copy() for k, p, s in zip(PARAM_NAMES, parts, shapes)}

    def state_dict(self):
        vis, aud, out = self.sizes
        flat = np.empty(out * vis + out + out * aud + out, np.float32)
        _lib.check(_lib._lib.acav_contrastive_get_params(self._h, _lib.ptr(flat), None))
        return self._split(flat)

    def load_state_dict(self, sd):
        flat = np.concatenate([np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], 'detach') else sd[k],
                                          np.float32).ravel() for k in PARAM_NAMES])
        _lib.check(_lib._lib.acav_contrastive_set_params(self._h, _lib.ptr(np.ascontiguousarray(flat))))

    # ------------------------------------------------------------------ compute
    def train_batches(self, visual, audio, offsets, lr):
        """the batches [offsets[i], offsets[i+1]) of one stream of aligned (visual, audio) rows: forward, backward,
        AdamW step each -> (losses, accs) like the reference's per-batch loss.item() / acc.item()"""
        visual = np.ascontiguousarray(visual, np.float32) if not hasattr(visual, 'data_ptr') else visual
        audio = np.ascontiguousarray(audio, np.float32) if not hasattr(audio, 'data_ptr') else audio
        off = np.ascontiguousarray(offsets, np.int64)
        nb = len(off) - 1
        losses, accs = np.empty(nb, np.float32), np.empty(nb, np.float32)
        _lib.check(_lib._lib.acav_contrastive_train(self._h, _lib.ptr(visual), _lib.ptr(audio), _lib.ptr(off), nb, float(lr),
                                                    _lib.ptr(losses), _lib.ptr(accs)))
        return losses, accs

    # ------------------------------------------------------------------ distributed training (contrastive.py:92-124)
    def set_comm(self, comm):
        """an RCCL communicator (parallel.rccl_comm.Comm) of more than one rank: train_batches / step average the
        gradients over its ranks before every optimizer step (ContrastiveModule.average_gradient, module.py:96-101)"""
        _lib.check(_lib._lib.acav_contrastive_set_comm(self._h, comm._h if comm is not None else None))
        self._comm = comm

    @staticmethod
    def rank_rows(offsets, rank, world, training=True):
        """get_features under distributed=True (contrastive.py:117-124): rows rank::world of every batch.
        -> (row indices, offsets of the rank's batches)

        A batch with fewer rows than ranks (the short tail batch of drop_last=False, or one that in-batch de-duplication
        shrank) would leave the high ranks with an empty slice: in the reference cross_entropy over it is NaN on those ranks
        and poisons the averaged gradients.  training=True drops such a batch ON EVERY RANK -- decided from the offsets
        alone, so all ranks agree and nobody is left waiting in the gradient all-reduce; training=False (inference: no
        collective) keeps it and simply scores an empty slice on the ranks past its end, as the reference does."""
        idx, off = [], [0]
        for i in range(len(offsets) - 1):
            lo, hi = int(offsets[i]), int(offsets[i + 1])
            if training and hi - lo < world:
                continue
            r = np.arange(lo + rank, hi, world, dtype=np.int64)
            idx.append(r)
            off.append(off[-1] + len(r))
        return (np.concatenate(idx) if idx else np.zeros(0, np.int64)), np.asarray(off, np.int64)

    def backward(self, visual, audio):
        """forward + backward of ONE batch, no optimizer step -> (loss, acc); the gradients accumulate"""
        visual, audio = np.ascontiguousarray(visual, np.float32), np.ascontiguousarray(audio, np.float32)
        loss, acc = np.empty(1, np.float32), np.empty(1, np.float32)
        _lib.check(_lib._lib.acav_contrastive_backward(self._h, _lib.ptr(visual), _lib.ptr(audio), visual.shape[0],
                                                       _lib.ptr(loss), _lib.ptr(acc)))
        return float(loss[0]), float(acc[0])

    def _nparam(self):
        vis, aud, out = self.sizes
        return out * vis + out + out * aud + out

    def get_grads(self):
        g = np.empty(self._nparam(), np.float32)
        _lib.check(_lib._lib.acav_contrastive_get_grads(self._h, _lib.ptr(g)))
        return g

    def set_grads(self, g):
        g = np.ascontiguousarray(g, np.float32)
        assert g.size == self._nparam()
        _lib.check(_lib._lib.acav_contrastive_set_grads(self._h, _lib.ptr(g)))

    def step(self, lr):
        _lib.check(_lib._lib.acav_contrastive_step(self._h, float(lr)))

    def train_batches_distributed(self, visual, audio, offsets, lr, rank, world):
        """train_batch with distributed=True for every batch of the stream: this rank's rows of the batch, backward, the
        gradients averaged over the ranks, AdamW step.  With an RCCL communicator set the whole loop is one library call;
        otherwise (the ranks share a GPU, gloo) the flat gradient buffer takes the torch.distributed route per batch."""
        idx, off = self.rank_rows(offsets, rank, world)
        v, a = np.ascontiguousarray(np.asarray(visual)[idx], np.float32), np.ascontiguousarray(np.asarray(audio)[idx], np.float32)
        if getattr(self, '_comm', None) is not None:
            return self.train_batches(v, a, off, lr)
        import torch
        import torch.distributed as dist
        nb = len(off) - 1
        losses, accs = np.empty(nb, np.float32), np.empty(nb, np.float32)
        on_gpu = dist.get_backend() == "nccl"
        for i in range(nb):
            losses[i], accs[i] = self.backward(v[off[i]:off[i + 1]], a[off[i]:off[i + 1]])
            g = torch.from_numpy(self.get_grads())
            g = g.cuda() if on_gpu else g
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            self.set_grads((g / float(world)).cpu().numpy())
            self.step(lr)
        return losses, accs

    def infer_scores(self, visual, audio):
        visual = np.ascontiguousarray(visual, np.float32) if not hasattr(visual, 'data_ptr') else visual
        audio = np.ascontiguousarray(audio, np.float32) if not hasattr(audio, 'data_ptr') else audio
        n = visual.shape[0]
        out = np.empty(n, np.float32)
        _lib.check(_lib._lib.acav_contrastive_infer(self._h, _lib.ptr(visual), _lib.ptr(audio), n, _lib.ptr(out)))
        return out

    # ------------------------------------------------------------------ the reference's train / infer drivers
    def train(self, args, path, batches, log_every=1, verbose=True):
        """contrastive.py:117-143.  `batches` = (visual [n, vis], audio [n, aud], offsets): the de-duplicated batch
        stream of run_contrastive.feature_batches (what the reference's dataloader + get_features yield)."""
        visual, audio, offsets = batches
        nb = len(offsets) - 1
        rank, world = 0, 1
        if self.distributed:
            from ...parallel import world as _world
            rank, world = _world()
        for epoch in range(self.epoch, self.num_epochs):
            lr = lr_func_linear(epoch + 1, self.num_epochs + 1, self.num_warmup_steps) * self.base_lr  # update_lr (:48-51)
            if world > 1:
                losses, accs = self.train_batches_distributed(visual, audio, offsets, lr, rank, world)
            else:
                losses, accs = self.train_batches(visual, audio, offsets, lr)
            if verbose:
                for count in range(0, nb, max(1, log_every)):
                    print("(node {}) training epoch ({}/{}) iter ({}/{}) (lr: {:04f}, loss: {:04f}, acc: {:04f})".format(
                        rank, epoch, self.num_epochs, count, nb, lr, losses[count], accs[count]))
                print("(node {}) epoch ({}/{}) done (lr: {:04f}, loss: {:04f}, acc: {:04f})".format(
                    rank, epoch, self.num_epochs, lr, float(np.mean(losses)) if nb else float('nan'),
                    float(np.mean(accs)) if nb else float('nan')))
            self.epoch = epoch
            if rank == 0:  # the replicas are identical; the reference's ranks all write the same file name
                self.save_cache(args, path, epoch, verbose)

    def get_cache_path_run(self, args, epoch):
        cache_dir = Path(args.data.output.path).parent / 'caches'
        cache_dir.mkdir(parents=True, exist_ok=True)
        name = "contrastive_model_cache_epoch_{}_{}_{}_{}".format(epoch, args.parent_pid, args.node_rank, args.chunk_num)
        return str(cache_dir / (name + '.pkl')), str(cache_dir / (name + '.json'))

    def get_cache_path_load(self, args, path, epoch):
        cache_dir = Path(args.data.output.path).parent / 'caches'
        cache_dir.mkdir(parents=True, exist_ok=True)
        keys = {p.stem: set(io.load_json(p)) for p in cache_dir.glob("contrastive_model_cache_epoch_{}_*.json".format(epoch))}
        want = set(Path(p).stem for p in path)
        fits = [(k, len(v & want)) for k, v in keys.items() if len(want - v) == 0]
        if not fits:
            return None
        return cache_dir / (max(fits, key=lambda x: x[1])[0] + '.pkl')

    def save_cache(self, args, chunks, epoch, verbose=True):
        """torch.save({'epoch', 'base_lr', 'model': state_dict}) -- the reference's file (contrastive.py:170-182)"""
        import torch
        path, key_path = self.get_cache_path_run(args, epoch)
        dt = {'epoch': self.epoch, 'base_lr': self.base_lr,
              'model': {k: torch.from_numpy(v) for k, v in self.state_dict().items()}}
        if verbose:
            print("saved cache file: {}".format(Path(path).stem))
        torch.save(dt, path)
        io.dump_json([Path(p).stem for p in chunks], key_path)

    def load_cache(self, args, path, epoch):
        import torch
        path = self.get_cache_path_load(args, path, epoch)
        assert path is not None, 'no cache file'
        dt = torch.load(path, map_location='cpu', weights_only=False)
        self.epoch = dt['epoch']
        self.base_lr = dt['base_lr']
        self.load_state_dict(dt['model'])

    def infer(self, args, batches, metas_rows, json_metas, subset_size, verbose=True):
        """contrastive.py:204-256: scores of every clip, appended to the per-process inference cache csv
        (score, shard_name, filename, id, segment); returns (scores desc, ids, rows) like the reference's topk."""
        visual, audio, offsets = batches
        if self.distributed:  # every rank scores its rows of every batch and writes its own cache file (contrastive.py:117-124,243)
            from ...parallel import world as _world
            rank, world = _world()
            if world > 1:
                idx, _ = self.rank_rows(offsets, rank, world, training=False)
                visual, audio = np.asarray(visual)[idx], np.asarray(audio)[idx]
                metas_rows = [metas_rows[i] for i in idx]
        logits = self.infer_scores(visual, audio)
        self.save_inference(args, logits, metas_rows, json_metas)
        k = len(logits) if subset_size is None or subset_size > len(logits) else int(subset_size)
        order = np.argsort(-logits, kind='stable')[:k]
        return logits[order], order, metas_rows

    def save_inference(self, args, logits, metas, json_metas):
        cache_dir = Path(args.data.output.path).parent / 'caches'
        cache_dir.mkdir(parents=True, exist_ok=True)
        # one file per process (contrastive.py:243-246: local rank); the chunks of a rank append to it
        name = "{}_contrastive_inferred_cache_{}_{}.csv".format(Path(args.data.output.path).stem, args.parent_pid,
                                                                int(args.node_rank or 0))
        print("saving cache to {}".format(cache_dir / name))
        with open(cache_dir / name, 'a+', newline='') as f:
            writer = csv.writer(f)
            for score, row in zip(logits.tolist(), metas):
                meta = json_metas.get(row['shard_name'], {}).get(Path(row['filename']).stem)
                if meta is None:
                    meta = {'id': '-1', 'segment': [-1.0, -1.0]}
                writer.writerow([score, row['shard_name'], row['filename'], meta['id'], meta['segment']])
