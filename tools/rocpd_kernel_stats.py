"""tools/rocpd_kernel_stats.py — summarise a rocprofv3 (rocpd SQLite) kernel trace into a per-kernel stats table
(the `--stats` CSV equivalent), so the summary can be committed under profiles/."""
import sqlite3, sys

def main(db_path, out_path=None):
    db = sqlite3.connect(db_path); cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
                  max(d.workgroup_size_x), max(d.grid_size_x), max(d.group_segment_size), max(d.private_segment_size) 
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
    try:
        rows = list(cur.execute(q))
    except sqlite3.OperationalError as e:
        print("schema:", cols); raise
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,WorkgroupSize,GridSize,LDSBytes,ScratchBytesPerLane"]
    for r in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9]))
    txt = "\n".join(lines) + "\n"
    if out_path: open(out_path, "w").write(txt)
    print(txt)

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
