"""Summarise a rocprofv3 --pmc run (rocpd sqlite) per kernel: average counter values per dispatch.
    python profiles/pmc_summary.py <results.db> [kernel-name-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
pick = lambda key: [t for t in tabs if key in t]
kd, ks = pick("kernel_dispatch")[0], pick("info_kernel_symbol")[0]
pe, pi = pick("pmc_event")[0], pick("info_pmc")[0]
cols = lambda t: [r[1] for r in c.execute(f"pragma table_info({t})")]
if len(sys.argv) > 3:
    for t in (kd, pe, pi):
        print(t, cols(t))
ev_cols, pi_cols = cols(pe), cols(pi)
ev_key = "event_id" if "event_id" in ev_cols else "dispatch_id"
kd_key = "event_id" if ev_key == "event_id" and "event_id" in cols(kd) else "id"
name_col = "symbol" if "symbol" in pi_cols else "name"
q = (f"select s.kernel_name, p.{name_col}, count(*), avg(e.value), avg(d.end - d.start) from {pe} e join {pi} p on e.pmc_id = p.id "
     f"join {kd} d on e.{ev_key} = d.{kd_key} join {ks} s on d.kernel_id = s.id group by 1, 2")
rows = list(c.execute(q))
want = sys.argv[2] if len(sys.argv) > 2 else ""
by = {}
for k, n, cnt, v, dur in rows:
    if want in k:
        by.setdefault(k, {"n": cnt, "us": dur / 1e3})[n] = v
for k, d in sorted(by.items(), key=lambda kv: -kv[1]["n"] * kv[1]["us"])[:12]:
    print(k[:90])
    print("   ", {a: (round(b, 1) if isinstance(b, float) else b) for a, b in d.items()})
