#!/usr/bin/env python3
"""Per-kernel averages of FETCH_SIZE / WRITE_SIZE (KiB) from the two rocprofv3 --pmc passes of tools/pmc_traffic.sh,
and the corrected HBM bytes per launch of the fc1 forward GEMM (the figure bench.py reports as roofline.traffic).
gfx950 correction (MI355X_MICROARCH.md, checked on streaming kernels in DESIGN.md section 7): FETCH_SIZE counts
half the bytes of wide coalesced reads -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections, csv, glob, json, sys

def avg(path, counter):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}

fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, (0, 0))[0] + write.get(k, (0, 0))[0]) * max(fetch.get(k, (0, 0))[1], 1)):
    fa, n = fetch.get(k, (0.0, 0)); wa, _ = write.get(k, (0.0, 0))
    rows.append((k, n, fa, wa))
print(f"{'kernel':72s} {'calls':>6s} {'read MB (x2)':>13s} {'write MB':>9s}")
for k, n, fa, wa in rows[:16]:
    print(f"{k[:72]:72s} {n:6d} {2 * fa * 1024 / 1e6:13.1f} {wa * 1024 / 1e6:9.1f}")
def series(path, counter, key):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter and key in r["Kernel_Name"]]

# the fc1 GEMM is also launched by bench.py's forward-only leg, where it writes ONE output (inference): keep the
# launches of the training steps (two outputs, > 120 MB written); the i-th launch of the two passes is the same launch
fs, wsz = series(sys.argv[1], "FETCH_SIZE", "gemm32_kernel<2, 4, 4>"), series(sys.argv[2], "WRITE_SIZE", "gemm32_kernel<2, 4, 4>")
train = [i for i in range(min(len(fs), len(wsz))) if wsz[i] * 1024 > 120e6]
if train:
    n = len(train)
    fa, wa = sum(fs[i] for i in train) / n, sum(wsz[i] for i in train) / n
    print(f"fc1 forward GEMM, training launches only: {n} launches, read {2 * fa * 1024 / 1e6:.1f} MB (x2), write {wa * 1024 / 1e6:.1f} MB")
    out = {"kernel": "gemm32_kernel<CARA_EPI_GELU, 4, 4> fc1 forward, M=12608 N=3072 K=768+32, weights from the K-panel-major image, grouped tile order (8 rows)",
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py (tools/pmc_traffic.sh + tools/pmc_summarize.py), %d launches averaged" % n,
           "fetch_size_kib": round(fa, 1), "write_size_kib": round(wa, 1),
           "correction": "gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "hbm_bytes_per_launch_corrected": int((2 * fa + wa) * 1024),
           "algorithmic_bytes": 179011584}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out))
