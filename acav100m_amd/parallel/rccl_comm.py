"""acav_comm -- the path's collectives through RCCL behind the C ABI (acav100m_amd/csrc/acav_comm.hip), for the
one-process-per-GPU runs.  The 128-byte RCCL id is the only thing that needs a side channel: it travels through the
torch.distributed group the launcher set up (a store-backed broadcast), after which the rows of the DDP epochs move
GPU to GPU with no torch op in between.
"""
import ctypes as C

import numpy as np

from .. import _lib
from .collectives import world

_default = None


class Comm:
    def __init__(self, rank, world_size, id128, device):
        lib = _lib.load_library()
        self._h = None
        h = C.c_void_p()
        ident = np.ascontiguousarray(id128, np.uint8)
        assert ident.size == 128
        _lib.check(lib.acav_comm_init(C.byref(h), int(device), int(rank), int(world_size), _lib.ptr(ident), None))
        self._h, self.rank, self.world = h, int(rank), int(world_size)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:
            _lib._lib.acav_comm_destroy(h)

    @staticmethod
    def unique_id():
        ident = np.zeros(128, np.uint8)
        _lib.check(_lib.load_library().acav_comm_unique_id(_lib.ptr(ident)))
        return ident

    @classmethod
    def from_process_group(cls, device=None):
        """one communicator over the ranks of the default torch.distributed group (or a world of one without it)"""
        import torch
        import torch.distributed as dist
        rank, w = world()
        device = torch.cuda.current_device() if device is None else device
        # Every rank must take the SAME route (acav_comm or the torch.distributed fallback), or the next collectives do
        # not match and the run hangs: the id travels with a status byte (rank 0 could not create it -> everybody raises),
        # and the outcome of the init is agreed on with one MIN all-reduce before anybody uses the communicator.
        msg = np.zeros(129, np.uint8)
        err = None
        if rank == 0:
            try:
                msg[1:] = cls.unique_id()
                msg[0] = 1
            except _lib.AcavError as exc:
                err = exc
        if w > 1:
            on = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.from_numpy(msg).to(on)
            dist.broadcast(t, 0)
            msg = t.cpu().numpy()
        if msg[0] != 1:
            raise err if err is not None else _lib.AcavError("rank 0 could not create the RCCL id")
        comm, err = None, None
        try:
            comm = cls(rank, w, msg[1:].copy(), device)
        except _lib.AcavError as exc:
            err = exc
        if w > 1:
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=on)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                comm = None  # dropped here: __del__ releases a handle that other ranks could not match
                raise err if err is not None else _lib.AcavError("acav_comm_init failed on another rank")
        elif comm is None:
            raise err
        return comm

    # thin wrappers over device tensors (anything with data_ptr())
    def allreduce_(self, t):
        _lib.check(_lib._lib.acav_comm_allreduce_f32(self._h, _lib.ptr(t), t.numel()))
        return t

    def allgather(self, send, recv):
        _lib.check(_lib._lib.acav_comm_allgather(self._h, _lib.ptr(send), _lib.ptr(recv), send.numel() * send.element_size()))
        return recv

    def broadcast_(self, t, root):
        _lib.check(_lib._lib.acav_comm_broadcast(self._h, _lib.ptr(t), t.numel() * t.element_size(), int(root)))
        return t

    def synchronize(self):
        _lib.check(_lib._lib.acav_comm_sync(self._h))


def default_comm(slot=0):
    """the process-wide communicator number `slot` over the launcher's group, created on first use (every rank must ask
    for the slots in the same order); None when RCCL is not the backend in use (CPU tests under gloo keep the
    torch.distributed plumbing).  One communicator per clustering lets their exchanges proceed independently."""
    global _default
    if _default is None:
        _default = {}
    if slot not in _default:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl":
            return None
        try:
            _default[slot] = Comm.from_process_group()
        except _lib.AcavError as exc:  # e.g. librccl not loadable: from_process_group makes every rank fail alike -> torch.distributed route
            import warnings
            warnings.warn(f"acav_comm unavailable ({exc}); the collectives of this run go through torch.distributed")
            _default[slot] = None
    return _default[slot]
