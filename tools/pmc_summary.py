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
            name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").strip()
            if name.endswith(")"):                            # drop the argument list (matched parentheses: `unsigned int __vector(4)*` nests), keep template arguments
                depth = 0
                for i in range(len(name) - 1, -1, -1):
                    depth += name[i] == ")"
                    depth -= name[i] == "("
                    if depth == 0:
                        name = name[:i].rstrip()
                        break
            if filters and not any(s in name for s in filters):
                continue
            c = row.get("Counter_Name")
            v = float(row.get("Counter_Value") or 0)
            a = acc[name][c]
            a[0] += 1
            a[1] += v
out = {k: {c: {"dispatches": n, "avg": s / n} for c, (n, s) in v.items()} for k, v in acc.items()}
print(json.dumps(out, indent=1, sort_keys=True))
