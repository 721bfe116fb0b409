"""Print per-kernel PMC counter sums from a rocprofv3 (rocpd sqlite) result."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print([t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()])
    sys.exit(0)
cur = c.execute(f"select * from {view} limit 1")
cols = [d[0] for d in cur.description]
kn = next(cn for cn in cols if "kernel" in cn.lower() and "name" in cn.lower())
cn_ = next(cn for cn in cols if cn.lower() in ("counter_name", "name") and cn != kn)
vn = next(cn for cn in cols if cn.lower() in ("value", "counter_value"))
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for k, n, v in c.execute(f"select {kn}, {cn_}, {vn} from {view}"):
    agg[k][n] += v
    cnt[k][n] += 1
import re
flt = re.compile(sys.argv[2] if len(sys.argv) > 2 else "")
for k in agg:
    if flt.search(k):
        print(k[:90])
        for n in sorted(agg[k]):
            print(f"   {n:32s} total {agg[k][n]:.4g}   per-dispatch {agg[k][n] / cnt[k][n]:.4g}  (n={cnt[k][n]})")
