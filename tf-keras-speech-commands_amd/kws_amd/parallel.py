"""Data-parallel exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" on CPU
for tests).  The train step shards the global batch across ranks; the only collective on the data path is the
sum-all-reduce of the flat gradient buffer (134 932 floats for simple_cnn), issued as two buckets so that the first
(conv4 + dense + head, 82 % of the bytes, produced first by the backward pass) overlaps the rest of the backward."""
import os


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


class DataParallel(object):
    def __init__(self, group=None):
        d = _dist()
        self.group = group
        self.active = d.is_available() and d.is_initialized() and d.get_world_size(group) > 1
        self.world = d.get_world_size(group) if self.active else 1
        self.rank = d.get_rank(group) if self.active else 0
        self._comm_stream = None

    @property
    def grad_scale(self):
        """each rank scales its gradient of the LOCAL mean loss by 1/world, so the summed result is the global mean"""
        return 1.0 / self.world

    def shard(self, n):
        """contiguous slice of a global batch of n items owned by this rank"""
        per = (n + self.world - 1) // self.world
        lo = min(n, self.rank * per)
        return lo, min(n, lo + per)

    def sync_grads(self, grads, split=None, bucket_event=None):
        """sum-all-reduce the flat gradient tensor.  With `split` and a recorded `bucket_event` (CUDA tensors only) the
        early bucket grads[split:] is reduced on a side stream as soon as it is final."""
        if not self.active:
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
        else:
            d.all_reduce(grads, group=self.group)

    def mean_(self, tensor):
        """in-place mean over ranks (BatchNormalization moving statistics, logged metrics)"""
        if self.active:
            _dist().all_reduce(tensor, group=self.group)
            tensor /= self.world
        return tensor

    def sum_(self, tensor):
        if self.active:
            _dist().all_reduce(tensor, group=self.group)
        return tensor

    def broadcast_(self, tensor, src=0):
        if self.active:
            _dist().broadcast(tensor, src, group=self.group)
        return tensor
