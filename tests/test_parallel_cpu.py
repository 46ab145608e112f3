"""CPU, two processes over gloo: the data-parallel plumbing used by train.py / bench.py for N > 1.
The gradient all-reduce of the product runs inside libalignnet_hip.so on RCCL; here the same protocol is
exercised with the torch oracle standing in for the engine: shard -> local forward/backward (local-BN) ->
sum all-reduce -> scale by 1/world, and the communicator-id rendezvous with a recording fake engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "alignnet-3d_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

LABELS = ("translations", "rel_angles", "pc1_centers", "pc2_centers", "pc1_angles", "pc2_angles")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class FakeEngine:
    """Records what parallel.init_comm hands to the engine; holds a few variables for average_ema_shadows."""
    calls = []

    def __init__(self, rank=0):
        self.vars = {"w": np.full((2, 3), 7.0 + rank, np.float32), "a/bn/moving_mean": np.full(4, 1.0 + 2 * rank, np.float32),
                     "a/bn/moving_var": np.arange(4, dtype=np.float32) * (rank + 1)}

    def variables(self):
        return [("w", (2, 3), True), ("a/bn/moving_mean", (1, 4), False), ("a/bn/moving_var", (1, 4), False)]

    def get_variable(self, name):
        return self.vars[name]

    def set_variable(self, name, value):
        self.vars[name] = np.asarray(value, np.float32).reshape(self.vars[name].shape)

    def get_option(self, key):
        return {"comm_world": 1}[key]       # no engine communicator: average_ema_shadows goes through the process group

    @staticmethod
    def comm_unique_id():
        return bytes(range(128))

    def comm_init(self, rank, world, uid):
        self.calls.append((rank, world, uid))


def _shard_grads(spec, P, d, lo, hi):
    from oracle import alignnet_torch as T
    tp = T.to_torch(P, requires_grad=True)
    tm = T.TorchTp8(spec, tp)
    td = {k: torch.tensor(v[lo:hi]) for k, v in d.items()}
    B = hi - lo
    u = {k: torch.full((B, 16), 0.9, dtype=torch.float64) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
    loss = tm.loss(tm.forward(td["pcs1"], td["pcs2"], True, 0.5, u), *[td[k] for k in LABELS])
    loss.backward()
    names = sorted(k for k, v in tp.items() if v.requires_grad)
    return torch.cat([(tp[k].grad if tp[k].grad is not None else torch.zeros_like(tp[k])).reshape(-1) for k in names]), float(loss)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alignnet3d import parallel
    from oracle import alignnet_ref as R
    try:
        assert parallel.world_info() == (rank, rank, world)
        # 1. shards: exact cover, sizes differ by <= 1
        for n in (0, 1, 5, 8, 257):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
        # 2. communicator rendezvous
        eng = FakeEngine(rank)
        uid = parallel.init_comm(eng, dist)
        assert uid == bytes(range(128)) and eng.calls[-1] == (rank, world, uid)
        assert parallel.broadcast_bytes(dist, b"xyz" if rank == 0 else None) == b"xyz"
        # 2b. one permutation per epoch for all ranks (rank 0's), each rank takes its slice of every global batch
        perm = list(np.random.default_rng(100 + rank).permutation(16))          # ranks would disagree ...
        perm = parallel.broadcast_object(dist, perm)
        assert perm == list(np.random.default_rng(100).permutation(16))         # ... everyone now holds rank 0's
        B = 8
        lo_b, hi_b = parallel.shard_range(B, rank, world)
        mine = [perm[b * B + lo_b:b * B + hi_b] for b in range(2)]
        got = [None] * world
        dist.all_gather_object(got, mine)
        seen = sorted(int(x) for r in got for bt in r for x in bt)
        assert seen == list(range(16))                                           # an epoch is one pass: no duplicates, none skipped
        # 2c. EMA shadows (non-trainable variables) are averaged across ranks, trainable ones are left alone
        assert parallel.average_ema_shadows(eng, dist) == 2
        np.testing.assert_allclose(eng.vars["a/bn/moving_mean"], np.full(4, 2.0))           # mean of 1 and 3
        np.testing.assert_allclose(eng.vars["a/bn/moving_var"], np.arange(4) * 1.5)
        np.testing.assert_array_equal(eng.vars["w"], np.full((2, 3), 7.0 + rank, np.float32))
        assert abs(parallel.mean_scalar(dist, 1.0 + rank) - 1.5) < 1e-12
        # 3. unequal row gather (eval predictions)
        counts = [3, 2]
        local = np.full((counts[rank], 4), float(rank + 1), np.float32)
        full = parallel.gather_rows(dist, local, counts)
        assert full.shape == (5, 4) and np.all(full[:3] == 1.0) and np.all(full[3:] == 2.0)
        # 4. data-parallel step: local-BN shards, sum all-reduce, 1/world scale == mean of the per-shard gradients
        spec = R.NetSpec(num_points=32, num_bins=6, s1_conv=(8, 16, 24), s2_conv=(8, 16, 32), emb_conv=(8, 16, 40), s1_fc=(16, 16),
                         s2_fc=(16, 16), rem_fc=(16, 16))
        P = R.init_params(spec, 0)
        R.randomize_bn(P)
        d = R.synth_pairs(8, 32, dtype=np.float64)
        lo, hi = parallel.shard_range(8, rank, world)
        g, loss = _shard_grads(spec, P, d, lo, hi)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        g /= world
        if rank == 0:
            ref = sum(_shard_grads(spec, P, d, *parallel.shard_range(8, r, world))[0] for r in range(world)) / world
            assert torch.allclose(g, ref, rtol=1e-12, atol=1e-14)
            assert float(g.abs().max()) > 0
        # 5. "sync_bn" + "global_loss": BatchNorm moments over all ranks (differentiable all-reduce of the sums: its backward all-reduces
        #    the gradient sums, which is what the engine's all-reduced (dbeta, dgamma) totals are), the loss on the gathered batch with
        #    the gradient into this rank's rows only, gradients SUMMED -> the single-process gradient at the global batch
        import torch.distributed.nn.functional as dfn
        from oracle import alignnet_torch as T

        class GlooSync:
            world, rank = dist.get_world_size(), dist.get_rank()
            @staticmethod
            def allreduce(t):
                return dfn.all_reduce(t.contiguous())
            @staticmethod
            def gather(t):
                out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
                dist.all_gather(out, t.detach().contiguous())
                return out

        def grads_of(lo_, hi_, sync):
            tp = T.to_torch(P, requires_grad=True)
            tm = T.TorchTp8(spec, tp, sync=sync)
            td = {k: torch.tensor(v[lo_:hi_]) for k, v in d.items()}
            u = {k: torch.full((hi_ - lo_, 16), 0.9, dtype=torch.float64) for k in ("s1_0", "s2_0", "s1_1", "s2_1", "rem")}
            ep = tm.forward(td["pcs1"], td["pcs2"], True, 0.5, u)
            labels = [td[k] for k in LABELS]
            lossv = tm.loss_global(ep, labels) if sync is not None else tm.loss(ep, *labels)
            lossv.backward()
            names = sorted(k for k, v in tp.items() if v.requires_grad)
            flat = torch.cat([(tp[k].grad if tp[k].grad is not None else torch.zeros_like(tp[k])).reshape(-1) for k in names])
            return flat, float(lossv), {k: v.clone() for k, v in tm.ema_updates.items()}

        gs_, loss_s, ema_s = grads_of(lo, hi, GlooSync)
        dist.all_reduce(gs_, op=dist.ReduceOp.SUM)           # summed, not averaged: the loss is already the global batch's
        ref_g, ref_loss, ref_ema = grads_of(0, 8, None)      # the same batch on one device
        assert abs(loss_s - ref_loss) <= 1e-12 * abs(ref_loss), (loss_s, ref_loss)
        assert torch.allclose(gs_, ref_g, rtol=1e-9, atol=1e-12 * float(ref_g.abs().max())), float((gs_ - ref_g).abs().max())
        for k in ref_ema:
            assert torch.allclose(ema_s[k], ref_ema[k], rtol=1e-12, atol=1e-14), k
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_process_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
