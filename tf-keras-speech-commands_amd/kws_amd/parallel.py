"""Data-parallel exchange: one process per GPU.

The train step shards the global batch across ranks; the only collective on the data path is the sum-all-reduce of the
flat gradient buffer (134 932 floats for simple_cnn), issued as two buckets so that the first (conv4 + dense + head, 82 %
of the bytes, produced first by the backward pass) overlaps the rest of the backward.

On GPUs the exchange goes through the C ABI (`kws_comm_*`: RCCL over xGMI, enqueued by the train step itself on streams it
already uses -- `kws_train_args.comm` -- or by `kws_allreduce_grads`, csrc/kws_comm.hip); torch.distributed is only the bootstrap (it ships the 128-byte RCCL id) and the control
plane (barriers, logging sums).  CPU tensors (the gloo tests) take the torch.distributed path with the same bucket /
weight arithmetic."""
import ctypes
import os

from . import lib as _l


def _dist():
    import torch.distributed as dist
    return dist


def is_distributed():
    try:
        d = _dist()
        return d.is_available() and d.is_initialized() and d.get_world_size() > 1
    except Exception:
        return False


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them)."""
    # two HIP hardware queues: with an RCCL communicator in the process the default of four dealt the step's three streams so that
    # its fork/join branches serialised (0.72 -> 1.2 ms/step at B = 4096).  Only read by the HIP runtime at its
    # first call, so this must run before anything touches the GPU; a value already in the environment wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
    import torch
    d = _dist()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or d.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    d.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world)


class KwsComm(object):
    """Owner of a `kws_comm` handle (include/kws.h): an RCCL communicator on the current device (it owns no stream: every
    collective is enqueued on a stream the caller names)."""

    def __init__(self, rank, world, unique_id):
        self._L = _l.get_lib()
        self._h = ctypes.c_void_p()
        if len(unique_id) != _l.COMM_ID_BYTES:
            raise ValueError("the RCCL unique id has %d bytes" % _l.COMM_ID_BYTES)
        buf = ctypes.create_string_buffer(bytes(unique_id), _l.COMM_ID_BYTES)
        _l.check(self._L.kws_comm_init(int(rank), int(world), buf, ctypes.byref(self._h)))
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id():
        """128 bytes rank 0 creates and every rank passes to the constructor"""
        buf = ctypes.create_string_buffer(_l.COMM_ID_BYTES)
        _l.check(_l.get_lib().kws_comm_unique_id(buf))
        return buf.raw

    _store_seq = 0

    @classmethod
    def from_torch_group(cls, group=None):
        """Bootstrap over an initialised torch.distributed group (any backend).  Rank 0's 128-byte id travels through the group's
        key-value store when it is reachable (no collective, so torch does not build its own NCCL communicator -- and that one's
        streams -- in front of ours: the order in which streams are created decides which hardware queues they share), else as a
        broadcast object."""
        d = _dist()
        rank, world = d.get_rank(group), d.get_world_size(group)
        store = None
        if group is None:
            try:
                store = d.distributed_c10d._get_default_store()
            except Exception:
                store = None
        if store is not None:
            key = "kws_comm_id_%d" % cls._store_seq
            cls._store_seq += 1
            if rank == 0:
                store.set(key, cls.unique_id())
            return cls(rank, world, bytes(store.get(key)))
        box = [cls.unique_id() if rank == 0 else None]
        d.broadcast_object_list(box, src=d.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0])

    @classmethod
    def single(cls):
        """a one-rank communicator (the same code path as N ranks; used by the 1-GPU tests)"""
        return cls(0, 1, cls.unique_id())

    @property
    def rccl_version(self):
        v = ctypes.c_int()
        _l.check(self._L.kws_comm_info(self._h, None, None, ctypes.byref(v)))
        return v.value

    @property
    def handle(self):
        return self._h

    def allreduce_grads(self, grads, split=0, state=None, state_weight=1.0):
        """kws_allreduce_grads: the exchange as a call of its own, on the current stream (a train step given `comm=` does it
        itself, with the early bucket overlapped; ranks may mix the two forms: same collectives, same order)."""
        import torch
        _l.check(self._L.kws_allreduce_grads(self._h, grads.data_ptr(), grads.numel(), int(split or 0),
                                             state.data_ptr() if state is not None else None,
                                             state.numel() if state is not None else 0, float(state_weight),
                                             torch.cuda.current_stream().cuda_stream))

    def allreduce(self, t, op="sum"):
        import torch
        dt = {torch.float32: _l.DT_F32, torch.float64: _l.DT_F64, torch.int32: _l.DT_I32, torch.int64: _l.DT_I64}[t.dtype]
        ro = {"sum": _l.OP_SUM, "max": _l.OP_MAX, "avg": _l.OP_AVG}[op]
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("kws_comm_allreduce needs a contiguous CUDA tensor")
        _l.check(self._L.kws_comm_allreduce(self._h, t.data_ptr(), t.numel(), dt, ro, torch.cuda.current_stream().cuda_stream))
        return t

    def broadcast(self, t, src=0):
        """kws_comm_broadcast: in-place broadcast of a contiguous CUDA tensor (any dtype: bytes travel) from rank `src`"""
        import torch
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("kws_comm_broadcast needs a contiguous CUDA tensor")
        _l.check(self._L.kws_comm_broadcast(self._h, t.data_ptr(), t.numel() * t.element_size(), int(src),
                                            torch.cuda.current_stream().cuda_stream))
        return t

    def timing(self, on=True):
        _l.check(self._L.kws_comm_timing(self._h, 1 if on else 0))

    def last_us(self):
        """(early bucket us, late bucket us) of the most recent allreduce_grads; None where not issued / not timed"""
        a, b = ctypes.c_float(), ctypes.c_float()
        _l.check(self._L.kws_comm_last_us(self._h, ctypes.byref(a), ctypes.byref(b)))
        return (a.value if a.value >= 0 else None, b.value if b.value >= 0 else None)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.kws_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_process_comms = {}


def process_comm(group=None):
    """The process's communicator for `group` (None = the default group): built collectively on first use, then shared by every
    DataParallel.for_device() -- a second ncclCommInitRank per fit() would cost ~100 ms and leave the old communicator to the garbage
    collector.  Returns None when RCCL cannot be bound (every rank gets the same answer: the library is either loadable or not)."""
    key = id(group) if group is not None else None
    if key not in _process_comms:
        try:
            _process_comms[key] = KwsComm.from_torch_group(group)
        except _l.KwsError as e:
            if e.code != _l.ERR_COMM:
                raise
            import warnings
            warnings.warn("RCCL is not available behind the C ABI (%s): gradients go through torch.distributed" % e)
            _process_comms[key] = None
    return _process_comms[key]


def close_process_comm():
    """Destroy the communicators process_comm() built (collective in effect: call on every rank, after the last step)."""
    for k in list(_process_comms):
        c = _process_comms.pop(k)
        if c is not None:
            c.close()


import atexit  # noqa: E402

atexit.register(close_process_comm)


class DataParallel(object):
    """Sharding arithmetic + the per-step exchange.  `comm`: a KwsComm (GPU path through the C ABI).  Without one the
    exchange uses torch.distributed on `group` (CPU tensors under gloo in the tests).  `force` makes a one-rank world
    take the exchange path too (1-GPU test of the overlapped branch)."""

    def __init__(self, group=None, comm=None, force=False):
        d = _dist()
        self.group = group
        self.comm = comm
        if comm is not None:
            self.world, self.rank = comm.world, comm.rank
            self.active = self.world > 1 or force
            self._torch_dist = d.is_available() and d.is_initialized() and d.get_world_size(group) == self.world and self.world > 1
        else:
            self._torch_dist = d.is_available() and d.is_initialized() and d.get_world_size(group) > 1
            self.active = self._torch_dist
            self.world = d.get_world_size(group) if self.active else 1
            self.rank = d.get_rank(group) if self.active else 0
        self._comm_stream = None

    @classmethod
    def for_device(cls, group=None):
        """What `fit` uses: with an initialised multi-rank group and a GPU, THE process's RCCL communicator behind the C ABI
        (process_comm: created once, collectively, at the first call and reused by every later fit; closed by
        close_process_comm() or at interpreter exit, never at a garbage-collection point that differs per rank); else the plain
        torch.distributed path (or inactive).  If RCCL cannot be bound (KWS_ERR_COMM on every rank alike: librccl missing) the
        torch.distributed path is the fallback."""
        import torch
        d = _dist()
        if d.is_available() and d.is_initialized() and d.get_world_size(group) > 1 and torch.cuda.is_available():
            comm = process_comm(group)
            if comm is not None:
                return cls(group, comm=comm)
        return cls(group)

    @property
    def grad_scale(self):
        """equal shards: each rank scales its gradient of the LOCAL mean loss by 1/world, so the sum is the global mean"""
        return 1.0 / self.world

    def shard(self, n):
        """contiguous slice of a global batch of n items owned by this rank"""
        per = (n + self.world - 1) // self.world
        lo = min(n, self.rank * per)
        return lo, min(n, lo + per)

    def shard_plan(self, n):
        """(lo, hi, weight) for a global batch of n items: weight = local items / n is BOTH this rank's grad_scale (its
        local-mean gradient times weight, summed over ranks, is the gradient of the global-batch mean even when the
        last batch splits unevenly) and its weight in the mean of the BatchNormalization moving statistics.  An empty
        shard has weight 0: that rank contributes zeros and only keeps the collective count equal."""
        lo, hi = self.shard(n)
        return lo, hi, (hi - lo) / float(n) if n > 0 else 0.0

    def sync_grads(self, grads, split=None, bucket_event=None, state=None, state_weight=None):
        """In-place sum over ranks of the flat gradient tensor, early bucket grads[split:] overlapped with the rest of
        the backward pass when `bucket_event` is given; `state` (BatchNormalization moving statistics) becomes its
        weighted mean over ranks in the same exchange (state_weight defaults to 1/world)."""
        if not self.active:
            return
        w = (1.0 / self.world) if state_weight is None else float(state_weight)
        if self.comm is not None and grads.is_cuda:
            self.comm.allreduce_grads(grads, split or 0, state, w)
            return
        d = _dist()
        if grads.is_cuda and split and bucket_event is not None:
            import torch
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            main = torch.cuda.current_stream()
            self._comm_stream.wait_event(bucket_event)
            with torch.cuda.stream(self._comm_stream):
                d.all_reduce(grads[split:], group=self.group)
            d.all_reduce(grads[:split], group=self.group)          # late bucket, after the backward on the main stream
            main.wait_stream(self._comm_stream)
        elif split:
            d.all_reduce(grads[split:], group=self.group)          # same two buckets, in order (CPU tensors)
            d.all_reduce(grads[:split], group=self.group)
        else:
            d.all_reduce(grads, group=self.group)
        if state is not None:
            state.mul_(w)
            d.all_reduce(state, group=self.group)

    def mean_(self, tensor):
        """in-place mean over ranks (logged metrics)"""
        if self.active:
            self.sum_(tensor)
            tensor /= self.world
        return tensor

    def sum_(self, tensor):
        if not self.active:
            return tensor
        if self.comm is not None and tensor.is_cuda:
            return self.comm.allreduce(tensor, "sum")
        _dist().all_reduce(tensor, group=self.group)
        return tensor

    def broadcast_(self, tensor, src=0):
        """in place from rank `src`.  CUDA tensors with a communicator travel through the C ABI (kws_comm_broadcast), so torch never
        builds an NCCL communicator of its own -- whose streams would change how HIP deals the step's streams to hardware queues
        (DESIGN.md section 6)."""
        if not self.active or self.world <= 1:
            return tensor
        if self.comm is not None and tensor.is_cuda:
            return self.comm.broadcast(tensor, src)
        if self._torch_dist:
            _dist().broadcast(tensor, src, group=self.group)
        return tensor
