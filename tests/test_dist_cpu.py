"""world_size-2 gloo tests (CPU) of the data-parallel path: the flat-buffer all-reduce used by
CaraEngine.train_step gives the gradients of the concatenated batch, and the epoch sharder
partitions an epoch without overlap.  Per-rank gradients come from the CPU oracle (tests may)."""
import os
import socket

import pytest
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


# ---- the engine's own flat buffer, p.grad views and optimizer step under gloo ---------------------------------
def _engine_worker(rank, world, port, out, prescaled=False):
    """Each rank: an adapted model (CPU construction), the ENGINE's flat gradient buffer and views
    (CaraEngine._grad_buffers), oracle gradients of this rank's shard written into the views, then the tail of
    CaraEngine.train_step -- p.grad binding, ONE all-reduce of the flat buffer, AdamW -- through _apply_gradients."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from cara_amd import cara, create_model
    from cara_amd.dist import broadcast_parameters
    torch.manual_seed(5)              # the frozen backbone is the same file on every rank ...
    vit = create_model("vit_base_patch16_224_in21k", depth=1, img_size=32, num_classes=10)
    head0 = {k: v.clone() for k, v in vit.head.state_dict().items()}
    torch.manual_seed(100 + rank)     # ... the adapters and the head are drawn from DIFFERENT seeds: the broadcast must make them equal
    m = cara({"model": vit, "rank": 4, "scale": 0.1, "l_mu": 1.5, "l_std": 0.1})
    with torch.no_grad():
        m.CP_A2.normal_(0, 0.05)      # (non-zero adapters, as the bench uses them)
        m.CP_P2.normal_(0, 0.05)
        m.head.weight.add_(0.01 * rank)
    eng = m._cara_engine
    trainable = eng.trainable_parameters()
    broadcast_parameters(trainable)
    w, _, x, y = _case()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    cp = {k: sd[k] for k in O.CP_NAMES}
    per = x.shape[0] // world
    wb = {k: v for k, v in sd.items() if not k.startswith("CP_")}
    head = {"weight": sd["head.weight"], "bias": sd["head.bias"]}
    _, _, g = O.train_step_as_written(x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per], wb, cp, head, s=0.1, depth=1)
    views = eng._grad_buffers(m, torch.device("cpu"))
    # prescaled: what train_step does -- dlogits leaves the cross-entropy divided by the world size (cara_cross_entropy_ex), so
    # every rank's gradients are 1/world of its shard's and the collective is a plain SUM with no scaling launch behind it
    f = 1.0 / world if prescaled else 1.0
    for n in eng.cp_fields:
        views[n].copy_(g["CP_" + n] * f)
    views["head_w"].copy_(g["head.weight"] * f)
    views["head_b"].copy_(g["head.bias"] * f)
    views["_found_inf"].fill_(float(rank))   # the word behind the gradients travels with them: raised on ONE rank, seen by all
    opt = torch.optim.AdamW(trainable, lr=1e-2, weight_decay=1e-4)
    eng._apply_gradients(opt, prescaled=prescaled)
    assert all(p.grad.data_ptr() == views[n].data_ptr() for n, p in zip(list(eng.cp_fields) + ["head_w", "head_b"], trainable))
    torch.save({"params": [p.detach().clone() for p in trainable], "flat": eng._flat_grad.clone(), "sd0": sd}, out + f".{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("prescaled", [False, True])
def test_engine_flat_views_allreduce_and_step(tmp_path, prescaled):
    out = str(tmp_path / "eng")
    mp.spawn(_engine_worker, args=(2, _free_port(), out, prescaled), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["flat"], r1["flat"])                                     # one buffer, the same mean on both ranks
    assert all(torch.equal(a, b) for a, b in zip(r0["params"], r1["params"]))      # identical replicas after the step
    assert all(torch.equal(r0["sd0"][k], r1["sd0"][k]) for k in O.CP_NAMES)        # ... and before it (broadcast)
    # the all-reduced buffer is the full-batch gradient
    w, _, x, y = _case()
    sd = r0["sd0"]
    full = O.train_step_as_written(x, y, {k: v for k, v in sd.items() if not k.startswith("CP_")}, {k: sd[k] for k in O.CP_NAMES},
                                   {"weight": sd["head.weight"], "bias": sd["head.bias"]}, s=0.1, depth=1)[2]
    ref = torch.cat([full[n].reshape(-1) for n in list(O.CP_NAMES) + ["head.weight", "head.bias"]])
    assert torch.allclose(r0["flat"][:-1], ref, rtol=1e-4, atol=1e-5 * ref.abs().max().item())
    assert r0["flat"][-1].item() == (1.0 if prescaled else 0.5)                   # the found-inf word: SUM / mean of {0, 1}
