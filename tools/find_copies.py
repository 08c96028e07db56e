#!/usr/bin/env python3
"""Which HIP API calls stand behind the __amd_rocclr_copyBuffer / fillBuffer kernels of a step: counts of hipMemcpy* / hipMemset* calls
in a rocprofv3 --hip-trace run of bench.py.  Usage: find_copies.py <rocprof output dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True)
c = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        n = r.get("Function", "")
        if "Memcpy" in n or "Memset" in n or "EventRecord" in n or "StreamWait" in n or "Synchronize" in n:
            c[n] += 1
for k, v in c.most_common():
    print(f"{v:8d}  {k}")
