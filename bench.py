#!/usr/bin/env python3
"""Headline benchmark: fine-tune images/sec, ViT-B/16 + CaRA rank 16 @224, bs 64 per GPU.

One "step" = the reference's train step (/root/reference/image_classification/vit_cp.py:45-50):
forward, mean cross-entropy, backward into the 12 CP tensors + head, (N > 1: one RCCL all-reduce
of the flat gradient buffer), AdamW.  Synthetic data and random-init weights of the named
architecture; inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
``roofline`` (dominant kernel = the fc1 forward GEMM, timed with HIP events on the compute stream
inside the timed region) and ``cpu_baseline`` (the oracle's as-written fp32 algorithm on the
host cores, bounded sample, rank 0 at N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"
# algorithmic GFLOP per image, SURVEY.md 8(d) / BASELINE.md section 3 (ViT-B/16, R=16)
GF_PER_IMG = {"fwd": 36.06, "bwd": 38.18, "step": 74.24}
# --model vit_large_patch16_384 (BASELINE.json configs[4], bs 32, rank 16): informational runs only, the
# reported metric stays the ViT-B configuration
# the roofline bracket (three HIP event records around the fc1 GEMM) idles the chip ~15 us per use: inside the
# timed region it goes around the fc1 GEMM of every 6th block only (2 launches per ViT-B step, 4 per ViT-L step)
PROFILE_EVERY = 6
GF_PER_IMG_L384 = {"fwd": 389.39, "bwd": 428.47, "step": 817.87}


def build_model(rank, scale, num_classes, device, seed, name="vit_base_patch16_224_in21k"):
    from cara_amd import cara, create_model
    torch.manual_seed(seed)
    vit = create_model(name, drop_path_rate=0.1, num_classes=num_classes)
    vit = cara({"model": vit, "rank": rank, "scale": scale, "l_mu": 1.5, "l_std": 0.1})   # cifar row of vtab_config.py:2-8
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():  # non-zero adapters (zero-init would make the K-extension trivially zero)
        vit.CP_A2.copy_(0.05 * torch.randn(vit.CP_A2.shape, generator=g))
        vit.CP_P2.copy_(0.05 * torch.randn(vit.CP_P2.shape, generator=g))
    vit = vit.to(device).train()
    trainable = []
    for n, p in vit.named_parameters():   # vit_cp.py:175-183
        if "CP" in n or "head" in n:
            trainable.append(p)
        else:
            p.requires_grad = False
    return vit, trainable


def cpu_baseline(rank, scale):
    """The as-written reference algorithm (dense dW + second GEMM per linear, fp32 autograd, AdamW)
    restated by the oracle, on the host cores.  Bounded: batch 16, 1 warm-up + 2 timed steps."""
    from oracle import cara_oracle as O
    # a 1-GPU box's CPU share is 16 cores; more threads than that only oversubscribes
    nthreads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(nthreads)
    bs = 16
    w = O.synthetic_backbone()
    cp = O.synthetic_cp(rank=rank)
    x, y = O.synthetic_batch(batch=bs)
    head = {"weight": w["head.weight"].clone(), "bias": w["head.bias"].clone()}
    params = [torch.nn.Parameter(v.clone()) for v in list(cp.values()) + list(head.values())]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-4)
    names = list(cp.keys())
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        cpd = {n: p for n, p in zip(names, params[:len(names)])}
        ww = dict(w)
        ww["head.weight"], ww["head.bias"] = params[-2], params[-1]
        logits = O.vit_cara_forward(x, ww, cpd, s=scale)
        loss = torch.nn.functional.cross_entropy(logits, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / 2
    return {"value": round(bs / dt, 3), "unit": "images/sec", "cores": nthreads, "kind": "port",
            "sample": f"2 timed train steps (fwd+bwd+AdamW) of batch {bs}, rank {rank}, fp32, reference's as-written "
                      f"dense-dW algorithm restated in oracle/cara_oracle.py; {dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weight-dropout", default="off", choices=["off", "exact"],
                    help="exact = the reference's train-mode Dropout(0.1) on the materialised adapters (merged weights + "
                         "dense dW gradients); informational, the reported metric uses the factored default")
    ap.add_argument("--model", default="vit_base_patch16_224_in21k",
                    help="vit_large_patch16_384 runs BASELINE.json configs[4] (use --batch 32); informational only")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # CARA_BENCH_REHEARSAL=1: rehearse the multi-rank plumbing on a ONE-GPU box (all ranks on
        # cuda:0, gloo instead of RCCL).  Never used for reported numbers.
        rehearsal = os.environ.get("CARA_BENCH_REHEARSAL") == "1"
        if rehearsal:
            local = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.gpus != world:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from cara_amd import _lib
    lib = _lib.lib()
    scale, ncls = 0.1, 100
    large = args.model == "vit_large_patch16_384"
    gf, img, tokens, dim = (GF_PER_IMG_L384, 384, 577, 1024) if large else (GF_PER_IMG, 224, 197, 768)
    model, trainable = build_model(args.rank, scale, ncls, dev, seed=14, name=args.model)  # identical replicas on every rank
    eng = model._cara_engine
    eng.weight_dropout = args.weight_dropout
    try:
        opt = torch.optim.AdamW(trainable, lr=1e-3, weight_decay=1e-4, fused=True)
    except Exception:
        opt = torch.optim.AdamW(trainable, lr=1e-3, weight_decay=1e-4)
    gx = torch.Generator().manual_seed(1000 + rank)   # each rank its own shard of the global batch
    x = torch.randn(args.batch, 3, img, img, generator=gx).to(dev)
    y = torch.randint(0, ncls, (args.batch,), generator=gx).to(dev)

    def step():
        return eng.train_step(x, y, opt)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.check(lib.cara_profile_fc1(PROFILE_EVERY), "cara_profile_fc1")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        loss = step()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    avg_ms, marker_ms, nl = C.c_float(0), C.c_float(0), C.c_int(0)
    if args.weight_dropout == "off":
        _lib.check(lib.cara_profile_fc1_read2(C.byref(avg_ms), C.byref(marker_ms), C.byref(nl)), "cara_profile_fc1_read2")
    else:   # the informational exact mode runs its linears through other entry points: no bracket
        avg_ms.value = float("nan")
    lib.cara_profile_fc1(0)
    t = torch.tensor([wall], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = t.item()

    # forward-only rate (SURVEY 8d asks for it next to the train rate): eval mode, no autograd, same batch;
    # outside the timed region above, rank 0's own clock
    fwd_ms = None
    if rank == 0:
        model.eval()
        with torch.no_grad():
            for _ in range(3):
                model(x)
            torch.cuda.synchronize()
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
            for _ in range(10):
                model(x)
            f1.record()
            torch.cuda.synchronize()
        fwd_ms = f0.elapsed_time(f1) / 10
        model.train()

    if rank == 0:
        ms_step = wall * 1e3 / args.steps
        ips = world * args.batch * args.steps / wall
        M, D = args.batch * tokens, dim
        fl_launch = 2.0 * M * (4 * D) * (D + args.rank)          # algorithmic: K = dim + rank (not the padded Rp)
        ach = fl_launch / (avg_ms.value * 1e-3) / 1e12 if avg_ms.value == avg_ms.value else 0.0
        # HBM-side bytes per launch of that kernel from a committed rocprofv3 PMC run (separate
        # FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 note of the microarch
        # guide); only valid for the headline shape
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_pmc_traffic_fc1.json")
        if os.path.exists(tj) and args.batch == 64 and args.rank == 16 and not large:
            with open(tj) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_launch_corrected")
        out = {
            "metric": ("fine-tune images/sec ViT-L/16+CaRA r=16 @384, bs=32/GPU (BASELINE.json configs[4], informational)" if large
                       else "fine-tune images/sec ViT-B/16+CaRA r=16 @224, bs=64/GPU, 1/2/4/8 MI355X"),
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"ViT-L/16 + CaRA rank={args.rank}, synthetic 384x384, bs={args.batch}/GPU, bf16 "
                                    "(BASELINE.json configs[4], informational); fwd + CE + bwd + AdamW, drop-path 0.1, factored adapters"
                                    if large else
                                    f"ViT-B/16 + CaRA rank={args.rank}, synthetic 224x224, bs={args.batch}/GPU, bf16 "
                                    "(BASELINE.json configs[1]); fwd + CE + bwd + AdamW, drop-path 0.1, "
                                    + ("factored adapters" if args.weight_dropout == "off" else
                                       "EXACT weight-space dropout 0.1 (merged weights, dense dW; informational)")),
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "step_algorithmic_gflop": round(gf["step"] * args.batch, 1),
                       "step_tflops_per_gpu": round(gf["step"] * args.batch / ms_step, 1),
                       "step_frac_of_mfma_peak": round(gf["step"] * args.batch / ms_step / PEAK_BF16_TFLOPS, 4),
                       "gpu_event_ms_per_step": round(ev_ms / args.steps, 3), "loss": float(loss),
                       "forward_only_ms": round(fwd_ms, 3),
                       "forward_only_images_per_sec_per_gpu": round(args.batch / fwd_ms * 1e3, 1),
                       "forward_frac_of_mfma_peak": round(gf["fwd"] * args.batch / fwd_ms / PEAK_BF16_TFLOPS, 4)},
            "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "kernel": f"gemm32_kernel<CARA_EPI_GELU> (fc1 forward, M={M} N={4 * D} K={D}+{args.rank}; the rocprofv3 name is gemm32_kernel<2>)",
                         "avg_launch_ms": round(avg_ms.value, 4) if avg_ms.value == avg_ms.value else None, "launches_timed": nl.value,
                         "event_marker_ms_subtracted": round(marker_ms.value, 4)},
        }
        if world == 1 and not args.no_cpu_baseline and not large:
            out["cpu_baseline"] = cpu_baseline(args.rank, scale)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
