"""world_size-2 gloo tests (CPU) of the data-parallel path: the flat-buffer all-reduce used by
CaraEngine.train_step gives the gradients of the concatenated batch, and the epoch sharder
partitions an epoch without overlap.  Per-rank gradients come from the CPU oracle (tests may)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cara_amd.dist import allreduce_mean_, epoch_shard, flat_views
from oracle import cara_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    w = O.synthetic_backbone(depth=1, img=32, num_classes=10)
    cp = O.synthetic_cp(rank=4)
    x, y = O.synthetic_batch(batch=4, img=32, num_classes=10)
    return w, cp, x, y


def _grads(w, cp, x, y):
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    _, _, g = O.train_step_as_written(x, y, w, cp, head, s=0.1, depth=1)
    return g


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    w, cp, x, y = _case()
    per = x.shape[0] // world
    g = _grads(w, cp, x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per])
    names = list(O.CP_NAMES) + ["head.weight", "head.bias"]
    flat, views = flat_views([(n, g[n].shape) for n in names], "cpu")
    for n in names:
        views[n].copy_(g[n])
    allreduce_mean_(flat)
    if rank == 0:
        torch.save({n: views[n].clone() for n in names}, out)
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_full_batch(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    w, cp, x, y = _case()
    full = _grads(w, cp, x, y)   # mean-CE over the concatenated batch == mean of equal-size shard means
    for n, g in got.items():
        assert torch.allclose(g, full[n], rtol=1e-4, atol=1e-5 * full[n].abs().max().item()), n   # fp32 summation order
    assert sum(v.numel() for v in got.values()) == 2526 * 4 + 4608 + 768 * 10 + 10


def test_epoch_shard_partitions():
    world, per = 2, 64
    a = epoch_shard(1000, epoch=3, rank=0, world=world, per_rank_batch=per, seed=14)
    b = epoch_shard(1000, epoch=3, rank=1, world=world, per_rank_batch=per, seed=14)
    assert len(a) == len(b) == 7                      # 500 per rank // 64, drop_last
    seen = torch.cat(a + b)
    assert len(torch.unique(seen)) == len(seen) == 7 * 64 * 2
    assert not torch.equal(a[0], epoch_shard(1000, 4, 0, world, per, 14)[0])   # reshuffled per epoch
    one = epoch_shard(1000, 0, 0, 1, 64, 14)
    assert len(one) == 15                             # the reference's 15 steps/epoch (vtab.py:84-88)
