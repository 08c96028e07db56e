import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find last adamw occurrences
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
i0 = idx[-3]
prev_end = None
for r in rows[i0 - 12: i0 + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1000 if prev_end else 0
    print(f"gap {gap:7.1f} us  dur {(e - s) / 1000:7.1f} us  {r['Kernel_Name'][:100]}")
    prev_end = e
