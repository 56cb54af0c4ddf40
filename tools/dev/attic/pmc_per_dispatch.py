import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
per = collections.defaultdict(float)
for name, ctr, disp, val in cur.execute('select kernel_name, counter_name, dispatch_id, value from counters_collection'):
    per[(name[:50], ctr, disp)] += val
for k, v in sorted(per.items()):
    if 'traceback' in k[0] or 'gather' in k[0]: print(k, round(v, 1))
