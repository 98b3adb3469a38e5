"""summarise a rocprofv3 rocpd database: per-kernel count / avg / min / max / total, optionally per grid"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
bygrid = len(sys.argv) > 2
q = "select name, %s count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name %s order by 6 desc" % (
    ("grid_x||'x'||grid_y||'x'||grid_z," if bygrid else "'',"), (", grid_x, grid_y, grid_z" if bygrid else ""))
tot = 0
for r in c.execute(q):
    print("%-64s %-16s n=%4d avg=%9.1f us min=%8.1f max=%8.1f tot=%9.2f ms" % (r[0][:64], r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6] / 1e6))
    tot += r[6]
print("total %.2f ms" % (tot / 1e6))
