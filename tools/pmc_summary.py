"""Summarise rocprofv3 --pmc counter_collection CSVs of tools/conv16_micro.py runs: per-kernel mean of each counter over the launches
(skipping warm-up) and the derived HBM bytes per launch = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE (KiB).
    python tools/pmc_summary.py <dir with *_counter_collection.csv files> [kernel substring ...]
    python tools/pmc_summary.py <dir> --write-json <source note>     (re-keys profiles/dominant_kernel_traffic.json for bench.py)"""
import csv, glob, os, sys, collections
d = sys.argv[1]
write_json = None
if "--write-json" in sys.argv:
    i = sys.argv.index("--write-json"); write_json = sys.argv[i + 1]; del sys.argv[i:i + 2]
subs = sys.argv[2:] or ["conv16s_kernel", "wgrad16_kernel"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        for s in subs:
            if s in k:
                acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
for s in subs:
    print("== %s ==" % s)
    c = acc[s]
    for name in sorted(c):
        v = c[name][2:] if len(c[name]) > 4 else c[name]          # skip warm-up launches
        print("%-32s launches=%d mean=%g" % (name, len(v), sum(v) / len(v)))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f = c["FETCH_SIZE"][2:]; w = c["WRITE_SIZE"][2:]
        print("HBM bytes per launch (2*FETCH+WRITE) = %.1f MB" % ((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e6))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        m = c["SQ_VALU_MFMA_BUSY_CYCLES"][2:]; b = c["SQ_BUSY_CYCLES"][2:]
        print("MFMA busy = %.1f %%  (SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs... see guide) raw ratio %.3f)" % (100 * sum(m) / sum(b) / 16, sum(m) / sum(b)))

if write_json is not None:
    import json
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, REPO)
    import bench
    c = acc["conv16s_kernel"]
    f = c["FETCH_SIZE"][2:]; w = c["WRITE_SIZE"][2:]
    fb, wb = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
    rec = {"bf16x3": {"hbm_bytes_per_launch": 2 * fb + wb,
                      "source": "%s (tools/pmc_conv16s.sh: FETCH_SIZE x2 + WRITE_SIZE = %.1f MB + %.1f MB, conv16s_kernel<true>)" % (write_json, 2 * fb / 1e6, wb / 1e6),
                      "conv16s_sha16": bench.conv16s_source_sha()}}
    json.dump(rec, open(os.path.join(REPO, "profiles", "dominant_kernel_traffic.json"), "w"), indent=1)
    print("wrote profiles/dominant_kernel_traffic.json:", rec)
