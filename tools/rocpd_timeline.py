"""tools/rocpd_timeline.py <results.db> [step] — the kernel timeline of ONE pipelined control step out of a rocprofv3 kernel trace (rocpd SQLite): start offset, duration,
stream-independent GPU idle time (no kernel of the process running).  Used to see what the step's critical path is made of (profiles/*_timeline_*.txt)."""
import sqlite3, sys

def main(db_path, which=-2):
    db = sqlite3.connect(db_path); cur = db.cursor()
    rows = list(cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
    short = lambda n: n.split("(")[0].replace("qm_", "").replace("_kernel", "")
    starts = [i for i, r in enumerate(rows) if "qm_grid_kernel" in r[0]]
    if len(starts) < 3: print("fewer than three steps in the trace"); return
    a, b = starts[which], starts[which + 1]
    t0 = rows[a][1]; busy_until = t0; idle = 0
    print("step of %.3f ms (grid kernel to grid kernel)" % ((rows[b][1] - t0) * 1e-6))
    for name, s, e in rows[a:b]:
        gap = max(0, s - busy_until); idle += gap; busy_until = max(busy_until, e)
        print("%-22s start %8.1f us  dur %8.1f us  %s" % (short(name), (s - t0) * 1e-3, (e - s) * 1e-3, ("idle before: %.1f us" % (gap * 1e-3)) if gap > 0 else ""))
    print("GPU idle inside the step: %.1f us" % (idle * 1e-3))

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -2)
