#!/usr/bin/env python3
"""Headline benchmark: fine-tune images/sec, ViT-B/16 + CaRA rank 16 @224, bs 64 per GPU.

One "step" = the reference's train step (/root/reference/image_classification/vit_cp.py:45-50):
forward, mean cross-entropy, backward into the 12 CP tensors + head, (N > 1: one RCCL all-reduce
of the flat gradient buffer), AdamW.  Synthetic data and random-init weights of the named
architecture; inputs are resident in HBM before the timed region.  Train mode: DropPath 0.1 (per-rank
random streams), factored adapters WITHOUT the reference's Dropout(0.1) on the materialised dW (that
mode is ``--weight-dropout exact``, informational).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script starts
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>`` as a CHILD process
(before anything touches the GPU) and exits with its code; under torchrun it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* itself.

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
``roofline``      the dominant kernel by measured time share (picked from a bracketed warm-up step), timed with
                  HIP events on the compute stream INSIDE the timed region;
``roofline_top``  the three kernel sites with the largest time share (name, algorithmic GFLOP, avg us, fraction
                  of the 2.5 PFLOP/s dense bf16 MFMA peak), from bracketed steps after the timed region;
``roofline_hbm``  the HBM-bound adapter-contraction / LayerNorm kernels in GB/s against 8 TB/s, same steps;
``cpu_baseline``  the oracle's as-written fp32 algorithm on the host cores (SURVEY 8d protocol, bounded sample,
                  rank 0 at N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0      # HBM3E spec (6.3 TB/s is what a copy achieves), same guide
# algorithmic GFLOP per image, SURVEY.md 8(d) / BASELINE.md section 3 (ViT-B/16, R=16)
GF_PER_IMG = {"fwd": 36.06, "bwd": 38.18, "step": 74.24}
# --model vit_large_patch16_384 (BASELINE.json configs[4], bs 32, rank 16): informational runs only, the
# reported metric stays the ViT-B configuration
GF_PER_IMG_L384 = {"fwd": 389.39, "bwd": 428.47, "step": 817.87}
# a bracket (three HIP event records) idles the chip ~15 us: inside the timed region only the dominant site is
# bracketed, on every 12th block (1 launch per ViT-B step, 2 per ViT-L step; every 6th until r05: same-box 7.869 -> 7.831 ms,
# profiles/r05_p_bracket_every_6_vs_12.txt)
PROFILE_EVERY = int(os.environ.get("CARA_BENCH_PROFILE_EVERY", "12"))   # (the variable: diagnostics only)

SITES = ["qkv_fwd", "proj_fwd", "fc1_fwd", "fc2_fwd", "qkv_bwd", "proj_bwd", "fc1_bwd", "fc2_bwd", "attn_fwd", "attn_bwd",
         "ln1_fwd", "ln2_fwd", "ln1_bwd", "ln2_bwd", "skinny_fwd", "skinny_bwd"]   # include/cara_hip.h CARA_SITE_*
SITE_KERNEL = {
    "qkv_fwd": "gemm32_kernel<BF16> (qkv forward, [xn1 | T][W | Vs]^T)",
    "proj_fwd": "gemm32ft_kernel<RESID> (proj forward, T = X U inside)",
    "fc1_fwd": "gemm32_kernel<GELU, MI=5> (fc1 forward, 160 x 128 tiles, two bf16 outputs u and gelu(u))",
    "fc2_fwd": "gemm8_kernel<G8<3,2>, RESID, MODE 2> (fc2 forward on the 160 x 256 x 64 tile, one workgroup per CU, T = X U inside)",
    "qkv_bwd": "gemm8_ts_kernel<G8<3,2>, BF16, MODE 2> (qkv dX on the 160 x 256 x 64 tile with G' = dY Vs inside; its dVs + proj's dU as workgroups behind the tiles)",
    "proj_bwd": "gemm32_ts_kernel<BF16,true> (proj dX + its dVs / dc + fc1's dU riding in the launch)",
    "fc1_bwd": "gemm32ft_ts_kernel<BF16,true> (fc1 dX with G' = dY Vs inside + its dVs / dc + fc2's dU riding in the launch)",
    "fc2_bwd": "gemm32_ts_kernel<DGELU,true, MI=5> (fc2 dX, 160 x 128 tiles, gelu' epilogue + its dVs / dc + the dU of the qkv above)",
    "attn_fwd": "attn_fwd_p2_kernel<7> (persistent, 3 heads per CU)", "attn_bwd": "attn_bwd_fused_kernel<true> (dK/dV sweep then dQ sweep per head)",
    "ln1_fwd": "ln_fwd_kernel<XU> (LayerNorm 1 + T = LN(x) U of qkv)", "ln2_fwd": "ln_fwd_kernel<XU> (LayerNorm 2 + T of fc1)",
    "ln1_bwd": "ln_bwd_kernel<XU> (LayerNorm 1 backward + G' of the fc2 below)", "ln2_bwd": "ln_bwd_kernel<XU> (LayerNorm 2 backward + G' of proj)",
    "skinny_fwd": "skinny_xu_sliced_kernel (T = X U)", "skinny_bwd": "skinny_xu_sliced_kernel (G' = dY Vs where no dX GEMM computes it: block 0's qkv)",
}


def site_work(M, D, R, B, H, N):
    """Algorithmic work per launch of every site: ('mfma', FLOP) or ('hbm', bytes).  R = the rank (not the padded Rp)."""
    def fwd(i, o, t_inside):   # [X | T][W | Vs]^T, plus T = X U when the GEMM computes it
        return 2.0 * M * o * (i + R) + (2.0 * M * i * R if t_inside else 0.0)

    # what a dX launch carries with the default switches: dX = [dY | G'][W^T | U]^T, its OWN dVs = dY^T T
    # (K1 = o), the dU = X^T G' of the PREVIOUS linear of the pass (K1 = du_k), and -- fc1 / qkv -- its own G' = dY Vs inside
    def bwd(i, o, du_k, g_inside):
        return 2.0 * M * i * (o + R) + 2.0 * M * R * (o + du_k) + (2.0 * M * o * R if g_inside else 0.0)
    att = 4.0 * B * H * N * N * 64
    return {
        "qkv_fwd": ("mfma", fwd(D, 3 * D, False)), "proj_fwd": ("mfma", fwd(D, D, True)),
        "fc1_fwd": ("mfma", fwd(D, 4 * D, False)), "fc2_fwd": ("mfma", fwd(4 * D, D, True)),
        "qkv_bwd": ("mfma", bwd(D, 3 * D, D, True)), "proj_bwd": ("mfma", bwd(D, D, D, False)),
        "fc1_bwd": ("mfma", bwd(D, 4 * D, 4 * D, True)), "fc2_bwd": ("mfma", bwd(4 * D, D, D, False)),
        "attn_fwd": ("mfma", att), "attn_bwd": ("mfma", 2.5 * att),     # S, dP, dV, dK, dQ: five products of 2 B H N^2 64
        # LayerNorm forward: read x fp32, write xn bf16; backward: read dy bf16 + x fp32 + dx fp32, write dx fp32 + dyb bf16
        "ln1_fwd": ("hbm", M * D * 6.0), "ln2_fwd": ("hbm", M * D * 6.0), "ln1_bwd": ("hbm", M * D * 16.0), "ln2_bwd": ("hbm", M * D * 16.0),
        # skinny contractions read their [M, K] operand once (bf16): forward K = D or 4 D (unused by default); backward, with the
        # default switches, only block 0's qkv (K = 3 D) still runs the pass (the other G' come out of dX GEMMs / LayerNorms)
        "skinny_fwd": ("hbm", M * 2.5 * D * 2.0), "skinny_bwd": ("hbm", M * 3.0 * D * 2.0),
    }


def site_bytes(M, D):
    """Algorithmic HBM bytes per launch of the matrix-core sites (every tensor a launch must read or write counted once, bf16
    activations, fp32 residual stream; the skinny adapter operands are noise): with `site_work` they say which roof binds a site --
    attention at 197 tokens and the N = K = dim products are HBM-bound kernels although they run on the matrix cores."""
    md, dd = float(M) * D, float(D) * D
    return {
        "qkv_fwd": 8 * md + 6 * dd,        # xn1 in, qkv out
        "proj_fwd": 10 * md + 2 * dd,      # ao in, fp32 residual stream read + written
        "fc1_fwd": 18 * md + 8 * dd,       # xn2 in, u and gelu(u) out
        "fc2_fwd": 16 * md + 8 * dd,       # h in, fp32 residual stream read + written
        "qkv_bwd": 10 * md + 6 * dd,       # dQKV in, dXn out, ao for proj's dU
        "proj_bwd": 6 * md + 2 * dd,       # dY in, dAO out, xn2 for fc1's dU
        "fc1_bwd": 18 * md + 8 * dd,       # dH in, dX out, h for fc2's dU
        "fc2_bwd": 20 * md + 8 * dd,       # dY and u in, dH out, xn1 for the qkv above's dU
        "attn_fwd": 8 * md,                # qkv in, out
        "attn_bwd": 16 * md,               # qkv, out, dout in; dqkv out
    }


def executed_gflop(gf_per_img, batch, D, R, H, N, depth):
    """What the device EXECUTES of the algorithmic work (SURVEY 8d counts every block on every token): the last block's proj /
    fc1 / fc2 (forward and dX) and its attention run for the cls rows only -- only that row reaches the logits, logits bitwise
    unchanged -- and block 0, which has nothing trainable upstream, skips its qkv dX.  Returns (fwd, step) GFLOP per batch: the
    honest numerators of the utilisation figures (images/sec is unaffected)."""
    M, B = batch * N, batch
    lin = lambda i, o: 2.0 * i * o + 2.0 * R * (i + o)                     # per row: base product + factored adapter
    rows_saved = (M - B) * (lin(D, D) + lin(D, 4 * D) + lin(4 * D, D))     # proj + fc1 + fc2 of the last block on B rows instead of M
    att = 4.0 * B * H * N * N * 64
    att_saved = att * (1.0 - 1.0 / N)                                      # one query per (sample, head) in the last block
    fwd_saved = rows_saved + att_saved
    bwd_saved = rows_saved + 2.5 * att_saved                               # dX of the same three linears, attention backward
    # (the adapter-gradient products of the skipped rows -- 2 x the adapter forward -- go with them)
    bwd_saved += (M - B) * 2.0 * (2.0 * R * (D + D) + 2.0 * R * (D + 4 * D) + 2.0 * R * (4 * D + D))
    bwd_saved += M * lin(3 * D, D)                                         # block 0's qkv dX
    fwd = gf_per_img["fwd"] * batch - fwd_saved / 1e9
    step = gf_per_img["step"] * batch - (fwd_saved + bwd_saved) / 1e9
    return fwd, step


def build_model(rank, scale, num_classes, device, seed, name="vit_base_patch16_224_in21k", cp_length=4):
    from cara_amd import cara, create_model
    torch.manual_seed(seed)
    vit = create_model(name, drop_path_rate=0.1, num_classes=num_classes)
    vit = cara({"model": vit, "rank": rank, "scale": scale, "l_mu": 1.5, "l_std": 0.1,            # cifar row of vtab_config.py:2-8
                "cp_length": cp_length})
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():  # non-zero adapters (zero-init would make the K-extension trivially zero)
        vit.CP_A2.copy_((0.05 if cp_length != 2 else 0.02) * torch.randn(vit.CP_A2.shape, generator=g))
        vit.CP_P2.copy_(0.05 * torch.randn(vit.CP_P2.shape, generator=g))
    vit = vit.to(device).train()
    trainable = []
    for n, p in vit.named_parameters():   # vit_cp.py:175-183
        if "CP" in n or "head" in n:
            trainable.append(p)
        else:
            p.requires_grad = False
    return vit, trainable


def kernel_source_sha16():
    """sha256 (first 16 hex digits) over the kernel sources of the library, in a fixed order: identifies what a committed
    PMC measurement was taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cara_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _host_cores():
    """(physical cores of the host, logical CPUs this process may run on, cgroup CPU quota or None)."""
    phys = set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("physical id"):
                    pid = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":")[1].strip()
                elif not line.strip() and pid is not None:
                    phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
            if q != "max":
                quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    return (len(phys) or (os.cpu_count() or 1)), usable, quota


def _cpu_train_steps(O, w, scale, rank, bs, warm, timed, deadline):
    """median seconds per train step (fwd + bwd + AdamW) of the as-written algorithm, and the number of steps timed"""
    cp = O.synthetic_cp(rank=rank)
    x, y = O.synthetic_batch(batch=bs)
    head = {"weight": w["head.weight"].clone(), "bias": w["head.bias"].clone()}
    params = [torch.nn.Parameter(v.clone()) for v in list(cp.values()) + list(head.values())]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-4)
    names = list(cp.keys())
    times = []
    for it in range(warm + timed):
        if it >= warm and time.perf_counter() > deadline and len(times) >= 3:
            break   # bound the sample on a slow host
        t0 = time.perf_counter()
        cpd = {n: p for n, p in zip(names, params[:len(names)])}
        ww = dict(w)
        ww["head.weight"], ww["head.bias"] = params[-2], params[-1]
        logits = O.vit_cara_forward(x, ww, cpd, s=scale)
        loss = torch.nn.functional.cross_entropy(logits, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it >= warm:
            times.append(time.perf_counter() - t0)
    cpd = {n: p.detach() for n, p in zip(names, params[:len(names)])}
    return statistics.median(times), len(times), (x, cpd)


def cpu_baseline(scale):
    """the as-written reference algorithm (dense dW + second GEMM per linear, fp32 autograd, AdamW)
    restated by the oracle, on the host cores: config-1 shape (R = 8, bs 16) and the headline shape (R = 16, bs 64),
    2 warm-ups + median of 5 timed train steps each, on ALL the physical cores this process may use (the count is
    stated: a 1-GPU box hands out a share of the host), with the 16-thread / bs 16 sample of the earlier rounds
    beside it, plus the eval-forward rate.  Baseline only."""
    from oracle import cara_oracle as O
    phys, usable, quota = _host_cores()
    nthreads = max(1, min(phys, usable, quota or phys))
    w = O.synthetic_backbone()
    samples = {}
    t_begin = time.perf_counter()
    torch.set_num_threads(nthreads)
    med, n, _ = _cpu_train_steps(O, w, scale, 8, 16, 2, 5, t_begin + 40)
    samples["cfg1_R8_bs16"] = {"images_per_sec": round(16 / med, 3), "s_per_step": round(med, 3), "timed_steps": n, "threads": nthreads}
    med, n, (x, cpd) = _cpu_train_steps(O, w, scale, 16, 64, 2, 5, time.perf_counter() + 110)
    samples["headline_R16_bs64"] = {"images_per_sec": round(64 / med, 3), "s_per_step": round(med, 3), "timed_steps": n, "threads": nthreads}
    head = samples["headline_R16_bs64"]
    with torch.no_grad():
        O.vit_cara_forward(x[:16], w, cpd, s=scale)
        t0 = time.perf_counter()
        O.vit_cara_forward(x[:16], w, cpd, s=scale)
        samples["eval_forward_R16_bs16"] = {"images_per_sec": round(16 / (time.perf_counter() - t0), 3), "threads": nthreads}
    if nthreads != 16 and min(usable, quota or usable) >= 16:   # the sample the earlier rounds reported, for continuity
        torch.set_num_threads(16)
        med, n, _ = _cpu_train_steps(O, w, scale, 16, 16, 1, 3, time.perf_counter() + 25)
        samples["headline_R16_bs16_16threads"] = {"images_per_sec": round(16 / med, 3), "s_per_step": round(med, 3), "timed_steps": n, "threads": 16}
    return {"value": head["images_per_sec"], "unit": "images/sec", "cores": nthreads, "kind": "port",
            "sample": f"median of {head['timed_steps']} train steps (fwd+bwd+AdamW) after 2 warm-ups, batch 64, rank 16, fp32, the "
                      "reference's as-written dense-dW algorithm restated in oracle/cara_oracle.py (cara.py:15-95, vit_cp.py:45-50), "
                      f"torch.set_num_threads({nthreads}) = min(physical cores {phys}, CPUs this process may use {usable}, "
                      f"cgroup quota {quota})",
            "samples": samples, "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(), "host_physical_cores": phys,
            "usable_cpus": usable, "cgroup_cpu_quota": quota, "seconds": round(time.perf_counter() - t_begin, 1),
            "torch_parallel_info": " | ".join(l.strip() for l in torch.__config__.parallel_info().splitlines()[:4])}


def self_launch(args):
    """``python bench.py --gpus N`` without torchrun: start it ourselves, as a child, before any GPU call."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this host driver
    # eight Python enqueuers share the cgroup's cores with their helper threads: cap the host-side thread pools per rank
    env.setdefault("OMP_NUM_THREADS", "2")
    env.setdefault("MKL_NUM_THREADS", "2")
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks through torch.distributed.run", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def read_sites(lib, names):
    from cara_amd import _lib
    out = {}
    for n in names:
        avg, mark, cnt = C.c_float(0), C.c_float(0), C.c_int(0)
        _lib.check(lib.cara_profile_site_read(SITES.index(n), C.byref(avg), C.byref(mark), C.byref(cnt)), "cara_profile_site_read")
        if cnt.value > 0:
            out[n] = {"avg_ms": avg.value, "marker_ms": mark.value, "brackets": cnt.value}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-info-legs", action="store_true", help="skip the informational exact-dropout / rank-64 legs behind the timed region")
    ap.add_argument("--graph", action="store_true",
                    help="N = 1: capture the whole step (forward, cross-entropy, backward, AdamW) once per synthetic batch into a hipGraph and "
                         "replay it in the timed loop (cara_vit_forward / _backward only enqueue: DESIGN.md section 1); eager is the default")
    ap.add_argument("--no-precision-matched", action="store_true", help="skip the second timed region (precision = 'fp16': the 1e-3 build)")
    ap.add_argument("--all-sites", action="store_true", help="diagnostic: roofline_top lists every bracketed MFMA site, not the top three")
    ap.add_argument("--weight-dropout", default="off", choices=["off", "exact"],
                    help="exact = the reference's train-mode Dropout(0.1) on the materialised adapters (merged weights + "
                         "dense dW gradients); informational, the reported metric uses the factored default")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16"],
                    help="fp16 = the IEEE-half operand build (the mode inside north_star's 1e-3); diagnostic, the reported metric is bf16")
    ap.add_argument("--model", default="vit_base_patch16_224_in21k",
                    help="vit_large_patch16_384 runs BASELINE.json configs[4] (use --batch 32); informational only")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("CARA_BENCH_REHEARSAL") == "1"
    if args.gpus != world:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # CARA_BENCH_REHEARSAL=1: rehearse the multi-rank plumbing on a ONE-GPU box (all ranks on
        # cuda:0, gloo instead of RCCL).  Never used for reported numbers.
        if rehearsal:
            local = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:   # (the enqueue of a step is single-threaded Python; intra-op CPU pools only fight over the cores)
        torch.set_num_threads(1 if rank > 0 else 2)

    from cara_amd import _lib
    from cara_amd import dist as cdist
    lib = _lib.lib("fp16" if args.precision == "fp16" else "bf16")   # (the site brackets live in the library that runs the step)
    scale, ncls = 0.1, 100
    large = args.model == "vit_large_patch16_384"
    gf, img, tokens, dim, heads = (GF_PER_IMG_L384, 384, 577, 1024, 16) if large else (GF_PER_IMG, 224, 197, 768, 12)
    model, trainable = build_model(args.rank, scale, ncls, dev, seed=14, name=args.model)  # identical replicas on every rank
    eng = model._cara_engine
    eng.weight_dropout = args.weight_dropout
    eng.precision = args.precision
    cdist.broadcast_parameters(trainable)   # replicas identical by construction; this makes it a fact (outside the timed region)
    eng.seed_rank_streams(2024, rank)       # per-rank DropPath / weight-dropout masks
    # vit_cp.py:185's AdamW as one HIP launch (cara_amd/optim.py; CARA_BENCH_TORCH_ADAMW=1: torch's fused one, for A/B runs)
    from cara_amd.optim import AdamW
    use_graph = args.graph and world == 1 and args.weight_dropout == "off"
    if os.environ.get("CARA_BENCH_TORCH_ADAMW") == "1":
        opt = torch.optim.AdamW(trainable, lr=1e-3, weight_decay=1e-4, fused=True)
        use_graph = False
    else:
        # (capturable: step count and learning rates in device memory, so that a captured launch replays correctly)
        opt = AdamW(trainable, lr=1e-3, weight_decay=1e-4, capturable=use_graph)
    gx = torch.Generator().manual_seed(1000 + rank)   # each rank its own shard of the global batch
    # four different synthetic batches, resident in HBM before the timed region, fed in turn (one fixed batch would be
    # memorised within a few steps: loss 0.13 after 25 steps)
    NB = 4
    xs = [torch.randn(args.batch, 3, img, img, generator=gx).to(dev) for _ in range(NB)]
    ys = [torch.randint(0, ncls, (args.batch,), generator=gx).to(dev) for _ in range(NB)]
    x, y = xs[0], ys[0]
    fed = [0]

    # how many ranks the collective really spans: all-reduce of a one
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, device="cpu" if rehearsal else dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    graphs = []

    def step():
        i = fed[0] % NB
        fed[0] += 1
        if use_graph:
            opt.advance()
            if graphs:                      # (captured below, after the warm-up: one graph per resident batch)
                graphs[i][0].replay()
                return graphs[i][1]
        return eng.train_step(xs[i], ys[i], opt)

    def capture_graphs():
        torch.cuda.synchronize()
        for i in range(NB):
            gr = torch.cuda.CUDAGraph()
            gen = eng._device_generator(dev)
            if gen is not None:             # this rank's DropPath stream: its state advances under replay like the default generator's
                gr.register_generator_state(gen)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(gr, stream=side):
                    lg = eng.train_step(xs[i], ys[i], opt)
            torch.cuda.current_stream().wait_stream(side)
            graphs.append((gr, lg))
        torch.cuda.synchronize()

    M = args.batch * tokens
    work = site_work(M, dim, args.rank, args.batch, heads, tokens)
    sbytes = site_bytes(M, dim)
    all_mask = (1 << len(SITES)) - 1
    factored = args.weight_dropout == "off"
    for _ in range(max(args.warmup - 1, 0)):
        step()
    # last warm-up step with every site bracketed: which site has the largest time share?  (the exact mode runs
    # its linears through other entry points; its GEMM sites then hold several kernels and are not reported)
    dominant = "fc2_bwd"
    if factored and args.warmup > 0:
        _lib.check(lib.cara_profile_sites(C.c_ulonglong(all_mask), 1), "cara_profile_sites")
        step()
        torch.cuda.synchronize()
        pre = read_sites(lib, SITES)
        mf = {n: v for n, v in pre.items() if work[n][0] == "mfma"}
        if mf:
            dominant = max(mf, key=lambda n: mf[n]["avg_ms"] * mf[n]["brackets"])
    elif args.warmup > 0:
        step()
    if use_graph:
        lib.cara_profile_sites(C.c_ulonglong(0), 1)      # (no event brackets inside a captured step)
        capture_graphs()
        for _ in range(NB):
            step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.check(lib.cara_profile_sites(C.c_ulonglong(1 << SITES.index(dominant)) if (factored and not use_graph) else C.c_ulonglong(0), PROFILE_EVERY),
               "cara_profile_sites")
    # one event per step boundary (a record between two kernels of a stream idles the chip ~3 us: 0.03 % of a step):
    # ms_per_step_median next to the mean that `value` is computed from
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        loss = step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    e0, e1 = evs[0], evs[-1]
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    dom = read_sites(lib, [dominant]).get(dominant) if (factored and not use_graph) else None
    graphs_live = list(graphs)
    graphs.clear()           # everything behind the timed region runs eager (site brackets, the fp16 region, the informational legs)
    t = torch.tensor([wall], device="cpu" if (world > 1 and rehearsal) else dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = t.item()

    # after the timed region, rank 0's own clock: (i) three steps with every site bracketed on every block ->
    # roofline_top / roofline_hbm; (ii) the forward-only rate (SURVEY 8d asks for it next to the train rate)
    post, fwd_ms = {}, None
    if rank == 0 or world > 1:   # (every rank runs the bracketed steps: they contain the all-reduce)
        if factored:
            _lib.check(lib.cara_profile_sites(C.c_ulonglong(all_mask), 1), "cara_profile_sites")
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            post = read_sites(lib, SITES)
            if use_graph:            # the dominant site's launch time: from these bracketed eager steps (a replayed graph carries no brackets)
                dom = post.get(dominant)
    lib.cara_profile_sites(C.c_ulonglong(0), 1)
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

    # ---- the precision-matched object: the SAME step with precision = "fp16" (libcara_hip_f16.so: the same kernels with IEEE-half
    # MFMA operands, the build whose logits sit inside north_star's 1e-3 of the fp32 reference on every configuration the parity
    # tests run), timed under the same protocol as the headline region -- barrier + synchronize on both sides, max over ranks --
    # with the dominant site bracketed in THAT library.  Every rank runs it (it contains the all-reduce).
    pm = None
    if factored and args.precision == "bf16" and not args.no_precision_matched:
        try:
            lib16 = _lib.lib("fp16")
            eng.precision = "fp16"
            for _ in range(3):
                step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            _lib.check(lib16.cara_profile_sites(C.c_ulonglong(1 << SITES.index(dominant)), PROFILE_EVERY), "cara_profile_sites")
            pevs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
            pt0 = time.perf_counter()
            pevs[0].record()
            for i in range(args.steps):
                ploss = step()
                pevs[i + 1].record()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            pwall = time.perf_counter() - pt0
            pstep_ms = [pevs[i].elapsed_time(pevs[i + 1]) for i in range(args.steps)]
            pdom = read_sites(lib16, [dominant]).get(dominant)
            lib16.cara_profile_sites(C.c_ulonglong(0), 1)
            tt = torch.tensor([pwall], device="cpu" if (world > 1 and rehearsal) else dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            pm = {"wall": tt.item(), "step_ms": pstep_ms, "dom": pdom, "loss": float(ploss), "skipped": eng.skipped_steps,
                  "loss_scale": eng.loss_scale}
        except Exception as exc:   # noqa: BLE001 -- behind the headline measurement: recorded, never fatal
            pm = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        finally:
            eng.precision = args.precision
            eng._ws.clear()
            torch.cuda.empty_cache()

    # informational legs (rank 0, N = 1, the headline configuration only; never the headline number): the reference's
    # train-mode arithmetic (exact weight-space dropout) and BASELINE.json configs[3] (rank 64), 2 warm-ups + 5 steps each
    info = {}
    if rank == 0 and world == 1 and factored and not large and args.batch == 64 and args.rank == 16 and not args.no_info_legs:
        def timed_steps(fn, warm=2, n=5):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n
        # every leg on its own: a failure there (memory on a shared box, an environment switch the leg does not support) is
        # recorded and must not cost the headline line, which is already measured
        def leg(name, fn):
            try:
                fn()
            except Exception as exc:   # noqa: BLE001 -- diagnostics behind the measurement
                info[name + "_error"] = f"{type(exc).__name__}: {exc}"[:300]
            finally:
                eng.weight_dropout = "off"
                torch.cuda.empty_cache()

        def leg_exact():
            eng.weight_dropout = "exact"
            info["exact_dropout_ms_per_step"] = round(timed_steps(step), 3)
            eng.weight_dropout = "off"
            eng._ws.clear()

        def leg_rank64():
            m64, tr64 = build_model(64, scale, ncls, dev, seed=14, name=args.model)
            e64 = m64._cara_engine
            e64.seed_rank_streams(2024, rank)
            o64 = AdamW(tr64, lr=1e-3, weight_decay=1e-4)
            info["rank64_ms_per_step"] = round(timed_steps(lambda: e64.train_step(x, y, o64)), 3)
            info["rank64_images_per_sec"] = round(args.batch / info["rank64_ms_per_step"] * 1e3, 1)
            info["rank64_step_frac_of_mfma_peak"] = round(82.61 * args.batch / info["rank64_ms_per_step"] / PEAK_BF16_TFLOPS, 4)

        def leg_order2():
            # order 2 of dim_experiment.py (dense dim x dim QKV deltas, 9.4 M more parameters): the dense-delta form of the QKV linear
            m2, tr2 = build_model(args.rank, scale, ncls, dev, seed=14, name=args.model, cp_length=2)
            e2 = m2._cara_engine
            e2.seed_rank_streams(2024, rank)
            o2 = AdamW(tr2, lr=1e-3, weight_decay=1e-4)
            info["order2_qkv_ms_per_step"] = round(timed_steps(lambda: e2.train_step(x, y, o2)), 3)

        leg("exact_dropout", leg_exact)
        leg("rank64", leg_rank64)
        leg("order2_qkv", leg_order2)

    if rank == 0:
        ms_step = wall * 1e3 / args.steps
        ips = world * args.batch * args.steps / wall

        def entry(n, v, per_step):
            kind, amount = work[n]
            us = v["avg_ms"] * 1e3
            d = {"site": n, "kernel": SITE_KERNEL[n], "avg_launch_us": round(us, 2), "launches_per_step": per_step,
                 "share_of_step": round(us * per_step / (ms_step * 1e3), 4)}
            if kind == "mfma":
                d.update(algorithmic_gflop=round(amount / 1e9, 2), achieved_tflops=round(amount / us / 1e6, 1),
                         frac_of_mfma_peak=round(amount / us / 1e6 / PEAK_BF16_TFLOPS, 4))
                nb = sbytes.get(n)
                if nb:   # the roof that binds this launch: its MFMA time at peak or its HBM time at peak, whichever is longer
                    t_mfma, t_hbm = amount / (PEAK_BF16_TFLOPS * 1e6), nb / (PEAK_HBM_GBS * 1e3)   # us
                    d.update(algorithmic_mb=round(nb / 1e6, 1), frac_of_hbm_peak=round(nb / us / 1e3 / PEAK_HBM_GBS, 4),
                             binding_roof="hbm" if t_hbm > t_mfma else "mfma", frac_of_binding_roof=round(max(t_mfma, t_hbm) / us, 4))
            else:
                d.update(algorithmic_mb=round(amount / 1e6, 1), achieved_gbs=round(amount / us / 1e3, 1),
                         frac_of_hbm_peak=round(amount / us / 1e3 / PEAK_HBM_GBS, 4))
            return d
        ex_fwd, ex_step = executed_gflop(gf, args.batch, dim, args.rank, heads, tokens, 24 if large else 12)
        per_step = {n: v["brackets"] // 3 for n, v in post.items()}
        ranked = sorted((n for n in post if work[n][0] == "mfma"), key=lambda n: -post[n]["avg_ms"] * per_step[n])
        top = [entry(n, post[n], per_step[n]) for n in (ranked if args.all_sites else ranked[:3])]
        hbm = [entry(n, post[n], per_step[n]) for n in post if work[n][0] == "hbm"]
        if dom:
            fl = work[dominant][1]
            ach = fl / (dom["avg_ms"] * 1e-3) / 1e12
        else:
            fl, ach = 0.0, 0.0
        # HBM-side bytes per launch of the dominant kernel from a committed rocprofv3 PMC run (separate FETCH_SIZE /
        # WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 note of the microarch guide); headline shape only
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic_by_site.json")
        if os.path.exists(tj) and args.batch == 64 and args.rank == 16 and not large:
            with open(tj) as fh:
                tjd = json.load(fh)
            traffic = tjd.get(dominant, {}).get("hbm_bytes_per_launch_corrected")
            # where the number comes from, and whether the kernels have changed since: the PMC run records a hash of the
            # kernel sources it measured (tools/pmc_traffic.sh); a different hash today means the value is stale
            meta = tjd.get("_meta", {})
            traffic_src = {"file": "profiles/pmc_traffic_by_site.json", "profile": meta.get("profile"),
                           "kernel_source_sha16_at_measurement": meta.get("kernel_source_sha16"),
                           "kernel_source_sha16_now": kernel_source_sha16(),
                           "stale": meta.get("kernel_source_sha16") != kernel_source_sha16()}
        out = {
            "metric": ("fine-tune images/sec ViT-L/16+CaRA r=16 @384, bs=32/GPU (BASELINE.json configs[4], informational)" if large
                       else "fine-tune images/sec ViT-B/16+CaRA r=16 @224, bs=64/GPU, 1/2/4/8 MI355X"),
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "ms_per_step_median": round(statistics.median(step_ms), 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"ViT-L/16 + CaRA rank={args.rank}, synthetic 384x384, bs={args.batch}/GPU, bf16 "
                                    "(BASELINE.json configs[4], informational); fwd + CE + bwd + AdamW, drop-path 0.1, factored adapters"
                                    if large else
                                    f"ViT-B/16 + CaRA rank={args.rank}, synthetic 224x224, bs={args.batch}/GPU, bf16 "
                                    "(BASELINE.json configs[1]); fwd + CE + bwd + AdamW, drop-path 0.1, "
                                    + ("factored adapters, no weight-space dropout" if factored else
                                       "EXACT weight-space dropout 0.1 (merged weights, dense dW; informational)")),
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "batches": f"{NB} seeded synthetic batches per rank, resident in HBM before the timed region, fed in turn",
                       "launch": ("hipGraph replay: the whole step captured once per resident batch" if use_graph else "eager"),
                       "ranks_in_allreduce": ranks_seen, "backend": ("gloo-rehearsal" if rehearsal else "rccl") if world > 1 else "none",
                       "step_algorithmic_gflop": round(gf["step"] * args.batch, 1),
                       "step_tflops_per_gpu": round(gf["step"] * args.batch / ms_step, 1),
                       "step_frac_of_mfma_peak": round(gf["step"] * args.batch / ms_step / PEAK_BF16_TFLOPS, 4),
                       "gpu_event_ms_per_step": round(ev_ms / args.steps, 3), "loss": float(loss),
                       "forward_only_ms": round(fwd_ms, 3),
                       "forward_only_images_per_sec_per_gpu": round(args.batch / fwd_ms * 1e3, 1),
                       "forward_frac_of_mfma_peak": round(gf["fwd"] * args.batch / fwd_ms / PEAK_BF16_TFLOPS, 4),
                       # the same two fractions over what the device EXECUTES (executed_gflop: the cls-row shortcut of the last block
                       # and block 0's skipped qkv dX taken out of the numerator): hardware utilisation, not algorithmic throughput
                       "step_executed_gflop": round(ex_step, 1),
                       "step_frac_of_mfma_peak_executed": round(ex_step / ms_step / PEAK_BF16_TFLOPS, 4),
                       "forward_executed_gflop": round(ex_fwd, 1),
                       "forward_frac_of_mfma_peak_executed": round(ex_fwd / fwd_ms / PEAK_BF16_TFLOPS, 4)},
            "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": f"{SITE_KERNEL[dominant]}; site {dominant}: the largest time share of the step among the "
                                   f"bracketed sites; M={M}",
                         "algorithmic_gflop_per_launch": round(fl / 1e9, 2),
                         "avg_launch_ms": round(dom["avg_ms"], 4) if dom else None, "launches_timed": dom["brackets"] if dom else 0,
                         "event_marker_ms_subtracted": round(dom["marker_ms"], 4) if dom else None},
            "roofline_top": top,
            "roofline_hbm": hbm,
        }
        if args.all_sites:   # every site was bracketed: the step's sites against the sum of their OWN binding roofs
            rows = top + hbm
            t_sites = sum(r["avg_launch_us"] * r["launches_per_step"] for r in rows)
            t_roofs = sum(r["avg_launch_us"] * r["launches_per_step"] * r.get("frac_of_binding_roof", r.get("frac_of_hbm_peak", r.get("frac_of_mfma_peak", 0.0)))
                          for r in rows)
            out["sites_vs_their_roofs"] = {"sites_us_per_step": round(t_sites, 1), "roofs_us_per_step": round(t_roofs, 1),
                                           "frac": round(t_roofs / t_sites, 4) if t_sites else None,
                                           "note": "sum over the bracketed sites of launches x max(MFMA time at 2.5 PF/s, HBM time at 8 TB/s) "
                                                   "of the site's algorithmic work, over the sum of the measured launch times"}
        if pm is not None:
            if "error" in pm:
                out["precision_matched"] = {"dtype": "fp16", "error": pm["error"]}
            else:
                pms = pm["wall"] * 1e3 / args.steps
                pfl = work[dominant][1]
                pach = pfl / (pm["dom"]["avg_ms"] * 1e-3) / 1e12 if pm["dom"] else 0.0
                out["precision_matched"] = {
                    "what": ("the same step with precision = 'fp16': libcara_hip_f16.so, the same kernels with IEEE-half MFMA operands at the "
                             "same MFMA rate, fp32 accumulation / residual stream, device-side dynamic loss scale -- the build whose logits are "
                             "within north_star's 1e-3 (rel-L2) of the fp32 reference with every class index equal on every configuration of "
                             "tests/test_model_gpu.py (PRECISIONS); same timing protocol as `value`"),
                    "dtype": "fp16", "value": round(world * args.batch * args.steps / pm["wall"], 2), "unit": "images/sec",
                    "ms_per_step": round(pms, 3), "ms_per_step_median": round(statistics.median(pm["step_ms"]), 3),
                    "vs_bf16_step": round(pms / ms_step, 4), "loss": pm["loss"],
                    "loss_scale": pm["loss_scale"], "steps_skipped_for_overflow": pm["skipped"],
                    "step_frac_of_mfma_peak": round(gf["step"] * args.batch / pms / PEAK_BF16_TFLOPS, 4),
                    "roofline": {"bound": "mfma", "achieved": round(pach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(pach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                                 "kernel": f"{SITE_KERNEL[dominant]}; site {dominant} (the bf16 region's dominant site) in the fp16 library; M={M}",
                                 "algorithmic_gflop_per_launch": round(pfl / 1e9, 2),
                                 "avg_launch_ms": round(pm["dom"]["avg_ms"], 4) if pm["dom"] else None,
                                 "launches_timed": pm["dom"]["brackets"] if pm["dom"] else 0}}
        if info.get("exact_dropout_ms_per_step"):
            # the reference's own recipe keeps the model in eval mode after the first evaluation (vit_cp.py:60,75): weight-space
            # dropout and DropPath are live for 165 of its 1 500 steps, so a run of that recipe costs this per step on average
            info["recipe_blended_ms_per_step"] = round((165 * info["exact_dropout_ms_per_step"] + 1335 * ms_step) / 1500, 3)
            # ... and that, not `value`, is the throughput of the reference's recipe AS WRITTEN (vit_cp.py:19-70 + cara.py:35,57,81,92)
            out["config"]["recipe_blended_ms_per_step"] = info["recipe_blended_ms_per_step"]
            out["config"]["recipe_blended_images_per_sec"] = round(world * args.batch / info["recipe_blended_ms_per_step"] * 1e3, 1)
        if info:
            info["note"] = ("informational, 5 steps each after 2 warm-ups, same box and process: exact = the reference's train-mode "
                            "Dropout(0.1) on the materialised dW (cara.py:35,57,81,92); rank64 = BASELINE.json configs[3] (82.61 GF/image); order2_qkv = cp_length 2 of dim_experiment.py (dense dim x dim QKV deltas) at the headline rank and batch; "
                            "blended = (165 exact + 1335 factored) / 1500, the reference's stuck-in-eval recipe (SURVEY 3.3)")
            out["informational"] = info
        if world == 1 and not args.no_cpu_baseline and not large:
            out["cpu_baseline"] = cpu_baseline(scale)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
