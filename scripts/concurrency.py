"""Reads a rocprofv3 kernel-trace CSV of m-loam_amd/host/framebench `pipes` (K = 1, 2, 4, 8 pipelines one after the other) and says, per phase, what the GPU did:
per kernel name the mean duration at each K, the sum of kernel time per frame, and how much of the wall time had 0 / 1 / 2+ kernels in flight.
usage: python scripts/concurrency.py <kernel_trace.csv> [frames per pipeline]"""
import csv, sys, collections, re
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Thread_Id"]), int(r["Queue_Id"])))
rows.sort()
# phases: the set of host threads that launch changes with K (framebench starts K fresh threads per phase)
by_thread = collections.defaultdict(list)
for s, e, n, t, q in rows:
    by_thread[t].append((s, e, n))
threads = sorted(by_thread, key=lambda t: by_thread[t][0][0])
spans = {t: (by_thread[t][0][0], by_thread[t][-1][1], len(by_thread[t])) for t in threads}
# group threads whose spans overlap heavily into phases
phases = []
for t in threads:
    s, e, n = spans[t]
    if n < 1000:
        continue
    for ph in phases:
        if s < ph["e"] - 0.5 * (ph["e"] - ph["s"]):
            ph["threads"].append(t); ph["s"] = min(ph["s"], s); ph["e"] = max(ph["e"], e)
            break
    else:
        phases.append(dict(threads=[t], s=s, e=e))
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.replace("mlh::", "")
    return n[:60]
print("phases (host threads launching at the same time):", [len(p["threads"]) for p in phases])
stats = {}
for ph in phases:
    K = len(ph["threads"])
    ev = []
    per = collections.defaultdict(list)
    # skip the first 20 % of the phase (warm-up frames)
    t_lo = ph["s"] + 0.25 * (ph["e"] - ph["s"])
    for t in ph["threads"]:
        for s, e, n in by_thread[t]:
            if s < t_lo: continue
            ev.append((s, 1)); ev.append((e, -1))
            per[short(n)].append((e - s) / 1e3)
    ev.sort()
    busy = collections.Counter()
    depth, last = 0, ev[0][0]
    for t, d in ev:
        busy[min(depth, 4)] += t - last
        last = t; depth += d
    wall = ev[-1][0] - ev[0][0]
    ksum = sum(sum(v) for v in per.values())
    stats[K] = dict(per=per, wall=wall, ksum=ksum, busy=busy)
    print(f"K = {K}: wall {wall / 1e6:.2f} ms, kernel time summed {ksum / 1e3:.2f} ms ({ksum * 1e3 / wall:.2f} kernels in flight on average); "
          + "fraction of wall with 0/1/2/3/4+ kernels in flight: " + " / ".join(f"{busy[i] / wall:.2f}" for i in range(5)))
Ks = sorted(stats)
names = sorted(stats[Ks[0]]["per"], key=lambda n: -sum(stats[Ks[0]]["per"][n]))
print("%-62s" % "kernel: launches per phase at K=1 | mean us at K = " + ", ".join(str(k) for k in Ks))
for n in names[:28]:
    line = "%-62s %6d |" % (n, len(stats[Ks[0]]["per"][n]))
    for k in Ks:
        v = stats[k]["per"].get(n, [])
        line += " %8.2f" % (sum(v) / len(v) if v else 0.0)
    print(line)
