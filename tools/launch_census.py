"""Per-step kernel census from two `rocprofv3 --kernel-trace --stats` runs of bench.py that differ only in --steps:
(calls_B - calls_A) / (steps_B - steps_A) removes model construction, warm-up and the roofline leg.  Prints launches per
step, the ATen share, and GPU kernel time per step by kernel.
    python tools/launch_census.py A_kernel_stats.csv stepsA B_kernel_stats.csv stepsB [out.csv]"""
import csv
import sys


def load(path):
    d = {}
    if path.endswith(".db"):          # rocprofv3's default rocpd output: same numbers, straight from the dispatch table
        import sqlite3
        for name, calls, total in sqlite3.connect(path).execute("select name, count(*), sum(duration) from kernels group by name"):
            d[name] = (int(calls), float(total))
        return d
    for r in csv.DictReader(open(path)):
        d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return d


def main():
    a, sa, b, sb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
    ds = sb - sa
    rows = []
    for k, (cb, tb) in b.items():
        ca, ta = a.get(k, (0, 0.0))
        if cb - ca > 0:
            rows.append((k, (cb - ca) / ds, (tb - ta) / ds / 1e3))
    rows.sort(key=lambda r: -r[2])
    tot_l = sum(r[1] for r in rows)
    tot_t = sum(r[2] for r in rows)
    aten = [r for r in rows if "at::native" in r[0] or "rocclr" in r[0]]
    print("launches/step %.1f   kernel-time/step %.2f ms   ATen+copy launches/step %.1f (%.1f %%)  ATen time %.2f ms"
          % (tot_l, tot_t / 1e3, sum(r[1] for r in aten), 100 * sum(r[1] for r in aten) / tot_l, sum(r[2] for r in aten) / 1e3))
    out = open(sys.argv[5], "w") if len(sys.argv) > 5 else None
    if out:
        out.write("kernel,launches_per_step,us_per_step,avg_us\n")
    for k, l, t in rows:
        line = "%8.1f %10.1f us %8.1f us/launch  %s" % (l, t, t / l, k[:150])
        print(line)
        if out:
            out.write('"%s",%.2f,%.1f,%.2f\n' % (k.replace('"', "'"), l, t, t / l))


if __name__ == "__main__":
    main()
