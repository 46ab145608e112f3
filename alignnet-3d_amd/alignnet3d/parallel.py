"""Data-parallel plumbing (not in the reference, which is single-device: train.py:189).  One process per GPU;
`torch.distributed` is used only for rendezvous (broadcast of the 128-byte RCCL id, barriers, gathering the
small prediction arrays); the gradient all-reduce itself runs inside libalignnet_hip.so on RCCL over xGMI."""
import os


def world_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n, rank, world):
    """Contiguous [lo, hi) slice of n items for `rank`; sizes differ by at most one, union is exact."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_bytes(dist, payload, src=0):
    """Send a small bytes object from `src` to every rank (works on gloo and nccl process groups)."""
    box = [payload if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def broadcast_object(dist, obj, src=0):
    """Any picklable object from `src` to every rank (the epoch's permutation of the training ids)."""
    box = [obj if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def mean_scalar(dist, x):
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()


def average_ema_shadows(engine, dist=None):
    """Average the non-trainable variables (the BatchNorm EMA shadows) across ranks.  Local-BN data parallelism updates each rank's
    shadows from its own shard's batch statistics; averaging them (the mean of the per-rank EMAs is the EMA of the per-rank means)
    keeps every rank's eval-mode model -- and the checkpoint rank 0 writes -- identical.  With the engine's own communicator (RCCL or
    loopback) this is one all-reduce of the shadow segment on the device (alignnet_comm_average_shadows); without one, the values
    travel through `dist` (a torch.distributed process group) on the host."""
    n = sum(1 for _, _, trainable in engine.variables() if not trainable)
    if not n:
        return 0
    if engine.get_option("comm_world") > 1:
        engine.comm_average_shadows()
        return n
    if dist is None or not dist.is_initialized() or dist.get_world_size() <= 1:
        return 0   # one rank (no communicator, a world-1 communicator, or no process group): its shadows are the average
    import numpy as np
    import torch
    names = [(nm, s) for nm, s, trainable in engine.variables() if not trainable]
    flat = np.concatenate([np.asarray(engine.get_variable(nm), np.float32).ravel() for nm, _ in names])
    t = torch.from_numpy(flat)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    flat = (t.cpu().numpy() / dist.get_world_size()).astype(np.float32)
    off = 0
    for nm, (r, c) in names:
        engine.set_variable(nm, flat[off:off + r * c])
        off += r * c
    return len(names)


def init_comm(engine, dist, make_id=None, grad_communicator=False):
    """Create the RCCL communicator of `engine` across the ranks of the default process group.
    grad_communicator: also a second communicator for the gradient buckets alone (Engine.comm_init_grad) -- with sync_bn the per-layer
    sums and the buckets then do not serialise on one communicator."""
    make_id = make_id or type(engine).comm_unique_id
    uid = broadcast_bytes(dist, make_id() if dist.get_rank() == 0 else None)
    engine.comm_init(dist.get_rank(), dist.get_world_size(), uid)
    if grad_communicator:
        uid2 = broadcast_bytes(dist, make_id() if dist.get_rank() == 0 else None)
        engine.comm_init_grad(dist.get_rank(), dist.get_world_size(), uid2)
    return uid


def gather_rows(dist, local, counts):
    """All-gather row blocks of unequal length (prediction arrays in eval); returns the concatenation on every rank."""
    import numpy as np
    import torch
    world = dist.get_world_size()
    width = local.shape[1]
    cap = max(counts)
    buf = torch.zeros(cap, width, dtype=torch.float32)
    buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local, np.float32))
    if dist.get_backend() == "nccl":
        buf = buf.cuda()
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)], axis=0)
