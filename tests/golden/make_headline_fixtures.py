#!/usr/bin/env python3
"""Expected values of the two whole-model cases at the BASELINE.json sizes (ViT-B/16 depth 12, batch 64, 197 tokens:
configs[1] rank 16 and configs[3] rank 64), computed ONCE in the build container by the CPU oracle's as-written fp32
algorithm (oracle/cara_oracle.py, itself pinned to the reference by make_golden.py) and committed as small .npz files:

    tests/golden/headline_b64_r16.npz, tests/golden/headline_b64_r64.npz
        loss, logits [64, 100] (fp32 as written), logits_bf16_sim (the oracle with the device path's rounding points),
        grad_<name> for the 12 CP tensors and the head, the DropPath multipliers that were used

The GPU test (tests/test_model_gpu.py::test_headline_batch_64_whole_model) then needs no 100-second CPU forward +
backward per rank on the GPU box.  Inputs are the seeded synthetic tensors of SURVEY 8d (oracle.synthetic_*), so the
test regenerates them bit for bit.  Usage:  python tests/golden/make_headline_fixtures.py [16 64]   |   ... vitl
(`vitl`: the expected values of the ViT-L/16 @384 parity test, tests/golden/vit_large_384_b2_r16.npz;
 `vitl32`: the eval logits of BASELINE.json configs[4] at its REAL per-GPU batch, 32 x 577 = 18 464 token rows,
 tests/golden/vit_large_384_b32_r16_logits.npz)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cara_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def keep_masks(depth, B, seed=11):
    """the same draw as tests/test_model_gpu.py::_keep"""
    g = torch.Generator().manual_seed(seed)
    rates = torch.linspace(0, 0.1, depth)
    keep = (1 - rates).reshape(-1, 1, 1)
    return ((keep + torch.rand(depth, 2, B, generator=g)).floor() / keep).float()


def main():
    ranks = [int(a) for a in sys.argv[1:]] or [16, 64]
    B = 64
    torch.set_num_threads(os.cpu_count() or 8)
    w = O.synthetic_backbone()
    x, y = O.synthetic_batch(batch=B)
    keep = keep_masks(12, B)
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    for rank in ranks:
        t0 = time.time()
        cp = O.synthetic_cp(rank=rank)
        loss, logits, grads = O.train_step_as_written(x, y, w, cp, head, s=0.1, drop_path_keep=keep)
        with torch.no_grad():
            sim = O.vit_cara_forward(x, w, cp, s=0.1, drop_path_keep=keep, factored=True, bf16_sim=True)
        out = {"loss": np.float64(loss.item()), "logits": logits.numpy(), "logits_bf16_sim": sim.numpy(), "droppath": keep.numpy(),
               "rank": np.int64(rank), "batch": np.int64(B)}
        for k, v in grads.items():
            out["grad_" + k] = v.numpy()
        path = os.path.join(HERE, f"headline_b64_r{rank}.npz")
        np.savez_compressed(path, **out)
        print(f"rank {rank}: loss {loss.item():.6f}, {os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s", flush=True)


def vit_large_b32():
    """BASELINE.json configs[4] at its real batch (32 images @384: M = 18 464 token rows, 115.4 row tiles of 160): eval-mode logits
    of the fp32 as-written oracle and of its bf16-rounded form, 8 samples at a time (samples are independent in eval mode)."""
    t0 = time.time()
    dims = dict(depth=24, dim=1024, heads=16)
    w = O.synthetic_backbone(img=384, **dims)
    cp = O.synthetic_cp(rank=16, **dims)
    x, _ = O.synthetic_batch(batch=32, img=384)
    ref, sim = [], []
    with torch.no_grad():
        for i in range(0, 32, 8):
            ref.append(O.vit_cara_forward(x[i:i + 8], w, cp, s=0.1, depth=24, num_heads=16))
            sim.append(O.vit_cara_forward(x[i:i + 8], w, cp, s=0.1, depth=24, num_heads=16, factored=True, bf16_sim=True))
            print(f"  samples {i}..{i + 7}: {time.time() - t0:.0f} s", flush=True)
    path = os.path.join(HERE, "vit_large_384_b32_r16_logits.npz")
    np.savez_compressed(path, logits=torch.cat(ref).numpy(), logits_bf16_sim=torch.cat(sim).numpy())
    print(f"ViT-L/16 @384, batch 32: {os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s", flush=True)


def vit_large():
    """tests/test_model_gpu.py::test_vit_large_384_against_oracle: ViT-L/16 @384 dimensioning, batch 2, rank 16, eval mode:
    fp32 logits, bf16-rounded logits, gradients of the mean cross-entropy (60 s of CPU on the GPU box otherwise)."""
    t0 = time.time()
    dims = dict(depth=24, dim=1024, heads=16)
    w = O.synthetic_backbone(img=384, **dims)
    cp = O.synthetic_cp(rank=16, **dims)
    x, y = O.synthetic_batch(batch=2, img=384)
    head = {"weight": w["head.weight"], "bias": w["head.bias"]}
    with torch.no_grad():
        ref = O.vit_cara_forward(x, w, cp, s=0.1, depth=24, num_heads=16)
        sim = O.vit_cara_forward(x, w, cp, s=0.1, depth=24, num_heads=16, factored=True, bf16_sim=True)
    _, _, grads = O.train_step_as_written(x, y, w, cp, head, s=0.1, depth=24, num_heads=16)
    out = {"logits": ref.numpy(), "logits_bf16_sim": sim.numpy()}
    for k, v in grads.items():
        if k != "head.bias":
            out["grad_" + k] = v.numpy()
    path = os.path.join(HERE, "vit_large_384_b2_r16.npz")
    np.savez_compressed(path, **out)
    print(f"ViT-L/16 @384: {os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "vitl32":
    torch.set_num_threads(os.cpu_count() or 8)
    vit_large_b32()
    sys.exit(0)
if __name__ == "__main__":
    if sys.argv[1:] == ["vitl"]:
        torch.set_num_threads(os.cpu_count() or 8)
        vit_large()
    else:
        main()
