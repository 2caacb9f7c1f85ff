"""Per-queue view of a rocprofv3 --kernel-trace CSV of `bench.py --mode train` (default schedule: main stream + weight-gradient side stream + auxiliary
stream): per training step (delimited by adam_kernel) each queue's busy time and idle time inside the step, the main queue's largest kernel families
WITH the slowdown they suffer beside the other queues (in-step duration against the --serial-wgrad stats if given), and the step's tail after the
main queue's last backward kernel.
   python tools/trace_queues.py <dir with *_kernel_trace.csv> [n_last_steps]"""
import collections
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return n[:64]


adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
adam = adam[-(nlast + 1):]
agg = collections.defaultdict(lambda: [0.0, 0])
qbusy = collections.Counter()
span = 0.0
conc = collections.Counter()
for a, b in zip(adam[:-1], adam[1:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][1]
    span += t1 - t0
    for s, e, n, q in seg:
        agg[(q, short(n))][0] += e - s
        agg[(q, short(n))][1] += 1
        qbusy[q] += e - s
    # concurrency histogram: time with k kernels in flight
    ev = []
    for s, e, n, q in seg:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    k, prev = 0, t0
    for t, d in ev:
        conc[k] += t - prev
        prev = t
        k += d
    conc[k] += t1 - prev
# the tail: last kernel of the busiest (main) queue before adam -> adam start, and what ran meanwhile
main_q = max(qbusy, key=qbusy.get)
tails = []
for a, b in zip(adam[:-1], adam[1:]):
    seg = rows[a + 1:b]
    last_main = max((e for s_, e, n_, q in seg if q == main_q and "adam" not in n_), default=rows[b][0])
    first_bwd = None
    tails.append((rows[b][0] - last_main) / 1e3)
print("tail (last main-queue kernel end -> adam start), us per step:", " ".join(f"{t:.0f}" for t in tails))
n = len(adam) - 1
print(f"{n} steps, span {span / n / 1e6:.3f} ms per step")
print("time with k kernels in flight (ms per step): " + "  ".join(f"{k}: {v / n / 1e6:.3f}" for k, v in sorted(conc.items())))
for q, v in sorted(qbusy.items(), key=lambda kv: -kv[1]):
    print(f"queue {q}: busy {v / n / 1e6:.3f} ms per step")
    fam = sorted(((k[1], v2) for k, v2 in agg.items() if k[0] == q), key=lambda kv: -kv[1][0])
    for name, (tt, cnt) in fam[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
        print(f"     {tt / n / 1e3:9.1f} us  {cnt / n:6.1f} x {tt / cnt / 1e3:8.1f} us   {name}")
