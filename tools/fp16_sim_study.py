#!/usr/bin/env python3
"""CPU-only: what the device path's rounding points cost in logit accuracy with bf16 operands (the product) and with fp16
operands (v_mfma_f32_16x16x32_f16 runs at the bf16 rate, 10 mantissa bits instead of 7) -- VERDICT r03 item 4.
The oracle's factored forward with `sim_dtype` rounding at the points the HIP path rounds, against the fp32 as-written forward.
Usage: python tools/fp16_sim_study.py [--depth 12] [--batch 4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cara_oracle as O  # noqa: E402


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--categories", action="store_true", help="the error budget: one rounding category at a time, and all but one")
    args = ap.parse_args()
    torch.set_num_threads(8)
    w = O.synthetic_backbone(depth=args.depth)
    cp = O.synthetic_cp(rank=args.rank, depth=args.depth)
    x, _ = O.synthetic_batch(batch=args.batch)
    with torch.no_grad():
        ref = O.vit_cara_forward(x, w, cp, s=0.1, depth=args.depth)
        fac = O.vit_cara_forward(x, w, cp, s=0.1, depth=args.depth, factored=True)
        out = {}
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            out[name] = O.vit_cara_forward(x, w, cp, s=0.1, depth=args.depth, factored=True, sim_dtype=dt)
    print(f"depth {args.depth} batch {args.batch} rank {args.rank}: logits rel-L2 vs the fp32 as-written forward")
    print(f"  factored fp32          {rel(fac, ref):.3e}")
    for name, o in out.items():
        print(f"  rounding model {name:5s}   {rel(o, ref):.3e}   argmax differs on {(o.argmax(1) != ref.argmax(1)).sum().item()} of {args.batch}")
    if args.categories:
        cats = ["images", "weights", "xn", "T", "qkv", "P", "ao", "h", "head"]
        with torch.no_grad():
            for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
                print(f"  {name}: one category alone | all but that category")
                for c in cats:
                    a = O.vit_cara_forward(x, w, cp, s=0.1, depth=args.depth, factored=True, sim_dtype=dt, sim_only={c})
                    b = O.vit_cara_forward(x, w, cp, s=0.1, depth=args.depth, factored=True, sim_dtype=dt, sim_skip={c})
                    print(f"    {c:8s} {rel(a, ref):.3e} | {rel(b, ref):.3e}", flush=True)
    print(f"  max |activation| seen by fp16 is bounded by its range 65504: check the fp16 run for inf/nan -> "
          f"{'finite' if torch.isfinite(out['fp16']).all() else 'NOT FINITE'}")


if __name__ == "__main__":
    main()
