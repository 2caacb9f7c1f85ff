"""Reduce rocprofv3 --pmc counter_collection CSVs to per-kernel per-dispatch averages (KB as rocprofv3 reports them).
   python tools/pmc_summary.py <dir-with-csvs> [substring-filter ...]   -> JSON on stdout"""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]
filters = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection*.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
            name = re.sub(r"\([^()]*\)$", "", name)          # drop the argument list, keep template arguments
            if filters and not any(s in name for s in filters):
                continue
            c = row.get("Counter_Name")
            v = float(row.get("Counter_Value") or 0)
            a = acc[name][c]
            a[0] += 1
            a[1] += v
out = {k: {c: {"dispatches": n, "avg": s / n} for c, (n, s) in v.items()} for k, v in acc.items()}
print(json.dumps(out, indent=1, sort_keys=True))
