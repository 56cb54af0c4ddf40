"""Summaries of rocprofv3 result databases (run_results.db) as CSV on stdout.
   python tools/rocpd_summary.py kernels <db>          -> per-kernel calls / total / mean / min / max ns (the --stats table)
   python tools/rocpd_summary.py counters <db> [...]   -> per-kernel mean of every counter collected (summed over its instances per dispatch)"""
import csv, sqlite3, sys, collections

mode, paths = sys.argv[1], sys.argv[2:]
w = csv.writer(sys.stdout)
if mode == 'kernels':
    cur = sqlite3.connect(paths[0]).cursor()
    rows = cur.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc').fetchall()
    tot = sum(r[2] for r in rows)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
        w.writerow([r[0], r[1], r[2], '%.1f' % r[3], '%.2f' % (100.0 * r[2] / tot), r[4], r[5]])
else:
    w.writerow(['kernel', 'counter', 'dispatches', 'avg_per_dispatch'])
    for p in paths:
        cur = sqlite3.connect(p).cursor()
        per = collections.defaultdict(float)
        for name, ctr, disp, val in cur.execute('select kernel_name, counter_name, dispatch_id, value from counters_collection'):
            per[(name, ctr, disp)] += val
        agg = collections.defaultdict(list)
        for (name, ctr, disp), v in per.items():
            agg[(name, ctr)].append(v)
        for (name, ctr), v in sorted(agg.items()):
            w.writerow([name, ctr, len(v), '%.1f' % (sum(v) / len(v))])
