#!/usr/bin/env python3
"""Busy/idle analysis of a rocprofv3 --kernel-trace CSV: how much of the wall time between the first and last
kernel of the LAST `--steps` training steps has no kernel running, and after which kernels the gaps sit.
A step is delimited by prep_kernel launches (one per forward)."""
import argparse, csv, collections, glob, sys

ap = argparse.ArgumentParser()
ap.add_argument("path", help="directory holding *_kernel_trace.csv")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--skip-last", type=int, default=0,
                help="training steps to leave out at the end of the trace (bench.py ends with 3 steps in which EVERY kernel "
                     "site is bracketed by HIP events: ~15 us of idle chip per bracket, not part of the timed region)")
ap.add_argument("--list", action="store_true", help="also print the ordered kernels of the last step of the window")
a = ap.parse_args()
f = glob.glob(a.path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
has_bwd = [i for i, e in enumerate(ev) if "attn_bwd" in e[2]]
last_bwd = has_bwd[-1]
if a.skip_last:   # cut the trace before the prep_kernel of the first skipped step
    allp = [i for i, e in enumerate(ev) if "prep_kernel" in e[2] and i < last_bwd]
    cut = allp[-a.skip_last]
    ev = ev[:cut]
    has_bwd = [i for i, e in enumerate(ev) if "attn_bwd" in e[2]]
    last_bwd = has_bwd[-1]
preps = [i for i, e in enumerate(ev) if "prep_kernel" in e[2] and i < last_bwd]
# the last `steps` forward+backward steps: from the (steps)th-last prep before the last backward kernel
start_i = preps[-a.steps]
end_i = last_bwd
while end_i + 1 < len(ev) and "prep_kernel" not in ev[end_i + 1][2]:
    end_i += 1
win = ev[start_i:end_i + 1]
t0, t1 = win[0][0], max(e[1] for e in win)
busy, cur_end, gaps = 0, t0, collections.Counter()
gapn = collections.Counter()
prev = None
for s, e, n, q in win:
    if s > cur_end:
        gaps[(prev or "")[:60] + " -> " + n[:60]] += s - cur_end
        gapn[(prev or "")[:60] + " -> " + n[:60]] += 1
        cur_end = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        prev = n
tot = t1 - t0
print(f"window {tot/1e6:.3f} ms over {a.steps} steps = {tot/1e6/a.steps:.3f} ms/step; busy {busy/1e6:.3f} ms ({100*busy/tot:.1f}%), idle {(tot-busy)/1e6/a.steps*1e3:.0f} us/step; kernels/step {len(win)/a.steps:.0f}")
qs = collections.Counter(q for _, _, _, q in win)
print("queues:", dict(qs))
for k, v in gaps.most_common(25):
    print(f"{v/1e3/a.steps:8.1f} us/step  x{gapn[k]/a.steps:5.1f}  {k}")

if a.list:   # the ordered kernels of the LAST step of the window: offset, duration, gap before, name
    lp = [i for i, e in enumerate(win) if "prep_kernel" in e[2]][-1]
    step = win[lp:]
    s0, prev_end = step[0][0], step[0][0]
    print(f"\nlast step: {len(step)} kernels, {(max(e[1] for e in step) - s0) / 1e3:.1f} us")
    for s_, e_, n_, q_ in step:
        print(f"{(s_ - s0) / 1e3:9.1f} us  {(e_ - s_) / 1e3:7.1f} us  gap {max(0, s_ - prev_end) / 1e3:5.1f}  {n_[:90]}")
        prev_end = max(prev_end, e_)
