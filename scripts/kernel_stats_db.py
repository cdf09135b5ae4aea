"""Per-kernel totals of a rocprofv3 --kernel-trace run whose output is the rocpd SQLite database (rocprofv3's default format on
this image): name, launches, total ms, average us.  Usage: kernel_stats_db.py RESULTS.db [N]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc limit ?", (n,)).fetchall()
for name, k, total, avg in rows:
    print("%-70s n=%6d total %10.3f ms avg %10.1f us" % (name.split("(")[0][:70], k, total, avg))
