import csv, glob, os, sys, re
p = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = []
with open(p) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"])[:50]))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
win = rows[adam[-2] + 1: adam[-1] + 1]
gaps = []
for a, b in zip(win[:-1], win[1:]):
    g = (b[0] - a[1]) / 1e3
    if g > 3: gaps.append((g, a[2], b[2]))
print("span", (win[-1][1] - win[0][0]) / 1e3, "us; gaps > 3us:", len(gaps), "total", round(sum(g[0] for g in gaps), 1), "us")
for g in sorted(gaps, reverse=True)[:25]: print("  %.1f us  after %-50s before %s" % g)
