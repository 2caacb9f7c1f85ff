"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py --mode train`: per training step (delimited by adam_kernel) the wall span, the time at
least one kernel runs (union), the summed durations (overlap = sum - union), the idle gaps and which kernels surround the largest ones.
   python tools/trace_gaps.py <dir with *_kernel_trace.csv> [n_last_steps]"""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
print(f"{len(rows)} dispatches, {len(adam)} adam launches")
adam = adam[-(nlast + 1):]
gap_by_pair = collections.Counter()
tot = collections.Counter()
for a, b in zip(adam[:-1], adam[1:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][1]
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    prev_name = short(rows[a][2])
    last_end_name = prev_name
    cur_e = rows[a][1]
    for s, e, n, q in seg:
        if s > cur_e:
            gaps.append((s - cur_e, last_end_name, short(n)))
            busy += 0
            cur_e = e
            last_end_name = short(n)
            busy += e - s
        else:
            if e > cur_e:
                busy += e - cur_e
                cur_e = e
                last_end_name = short(n)
    ssum = sum(e - s for s, e, _, _ in seg)
    idle = sum(g[0] for g in gaps)
    tot["span"] += t1 - t0; tot["busy"] += busy; tot["sum"] += ssum; tot["idle"] += idle; tot["n"] += 1; tot["gaps"] += len(gaps)
    tot["queues"] = len(set(q for *_, q in seg))
    for g, p, n in gaps:
        gap_by_pair[(p, n)] += g
n = tot["n"]
print(f"per step over {n} steps: span {tot['span']/n/1e6:.3f} ms, union-busy {tot['busy']/n/1e6:.3f} ms, idle {tot['idle']/n/1e6:.3f} ms in {tot['gaps']/n:.0f} gaps, "
      f"summed durations {tot['sum']/n/1e6:.3f} ms (overlap {(tot['sum']-tot['busy'])/n/1e6:.3f} ms), queues {tot['queues']}")
print("largest idle time by (kernel before -> kernel after), us per step:")
for (p, nx), g in gap_by_pair.most_common(25):
    print(f"  {g/n/1e3:8.1f}  {p}  ->  {nx}")
# the first launches of the last step: name, duration, gap to the previous kernel's end (us), grid
a, b = adam[-2], adam[-1]
print("first 16 launches after the optimiser step:")
prev_e = rows[a][1]
for s, e, nme, q in rows[a + 1:a + 17]:
    print(f"  gap {max(0, s - prev_e)/1e3:7.1f} us  dur {(e - s)/1e3:7.1f} us  q{q}  {short(nme)}")
    prev_e = max(prev_e, e)
