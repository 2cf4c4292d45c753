"""Kernel statistics from a rocprofv3 rocpd SQLite database (same columns as --stats's kernel_stats.csv)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in db.execute(f"pragma table_info({sym})")]
name_col = "display_name" if "display_name" in cols else "kernel_name"
rows = db.execute(f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                  f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for n, c, t, lo, hi in rows:
    print(f'"{n}",{c},{t},{t / c:.1f},{100.0 * t / total:.2f},{lo},{hi}')
