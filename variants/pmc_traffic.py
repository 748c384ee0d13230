"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.
usage: pmc_traffic.py <dir with *counter_collection.csv files> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB... the unit is checked against the calibration kernels
(variants/pmc_calib.py) when they are present in the same directory tree."""
import csv, glob, json, os, re, sys
root, out = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = re.sub(r"\(.*", "", name).replace("void ", "")
        short = re.sub(r"<.*", "", short).replace("asurf::", "")
        key = (short, r["Counter_Name"])
        a = acc.setdefault(key, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
res = {}
for (k, c), (tot, n) in sorted(acc.items()):
    res.setdefault(k, {})[c] = dict(avg=tot / n, launches=n)
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    if any(t in k for t in ("k_", "elementwise", "fill", "copy", "Fill", "Memcpy")):
        print(k[:60], {c: round(x["avg"], 1) for c, x in v.items()}, [x["launches"] for x in v.values()])
