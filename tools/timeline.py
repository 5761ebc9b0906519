"""Per-step timeline from a rocprofv3 kernel trace of bench.py: stream busy times, main-stream occupancy per millisecond, largest gaps,
launch mix of the low-occupancy windows.   python tools/timeline.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ad = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
a, b = ad[-back - 1], ad[-back]
t0 = int(rows[a]['End_Timestamp']); t1 = int(rows[b]['End_Timestamp'])
print('step %.2f ms, %d launches' % ((t1 - t0) / 1e6, b - a))
st = collections.defaultdict(list)
for r in rows[a + 1:b + 1]:
    st[r['Stream_Id']].append(r)
main_id = max(st.items(), key=lambda kv: len(kv[1]))[0]
for s, rs in st.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6
    print('stream %s%s: %d launches, busy %.2f ms, %.2f..%.2f ms' % (s, ' (main)' if s == main_id else '', len(rs), busy, (int(rs[0]['Start_Timestamp']) - t0) / 1e6, (int(rs[-1]['End_Timestamp']) - t0) / 1e6))
main = st[main_id]
dur = (t1 - t0) / 1e6
bk = [0.0] * (int(dur) + 1); cnt = [0] * (int(dur) + 1)
for r in rows[a + 1:b + 1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e6; e = (int(r['End_Timestamp']) - t0) / 1e6
    if int(s) < len(cnt): cnt[int(s)] += 1
    if r['Stream_Id'] != main_id: continue
    i = int(s)
    while i < e and i < len(bk):
        bk[i] += max(0, min(e, i + 1) - max(s, i)); i += 1
print('main-stream busy %% per ms : ' + ' '.join('%3d' % round(100 * x) for x in bk))
print('launches per ms (all)    : ' + ' '.join('%3d' % c for c in cnt))
gaps = sorted(((int(q['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3, (int(p['End_Timestamp']) - t0) / 1e6, p['Kernel_Name'][:34], q['Kernel_Name'][:34]) for p, q in zip(main, main[1:]))
print('main-stream gaps: total %.2f ms, > 20 us: %.2f ms' % (sum(g[0] for g in gaps if g[0] > 0) / 1e3, sum(g[0] for g in gaps if g[0] > 20) / 1e3))
for g in gaps[-6:][::-1]:
    print('  %6.0f us at %6.2f ms  %s -> %s' % g)
