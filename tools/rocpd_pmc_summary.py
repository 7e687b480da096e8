"""tools/rocpd_pmc_summary.py — per-kernel average of the PMC counters in a rocprofv3 (rocpd SQLite) run.
usage: python tools/rocpd_pmc_summary.py <results.db> [out.csv]"""
import sqlite3, sys

def main(db_path, out_path=None):
    db = sqlite3.connect(db_path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda prefix: [x for x in tabs if x.startswith(prefix)][0]
    ev, info, disp, sym = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    q = f"""select s.kernel_name, i.name, i.units, count(*), avg(e.value), min(e.value), max(e.value)
            from {ev} e join {info} i on e.pmc_id = i.id join {disp} d on d.event_id = e.event_id join {sym} s on d.kernel_id = s.id
            group by s.kernel_name, i.name order by 5 desc"""
    rows = list(cur.execute(q))
    lines = ["Kernel,Counter,Units,Dispatches,AvgPerDispatch,Min,Max"] + ['"%s",%s,%s,%d,%.1f,%.1f,%.1f' % r for r in rows]
    txt = "\n".join(lines) + "\n"
    if out_path: open(out_path, "w").write(txt)
    print(txt)

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
