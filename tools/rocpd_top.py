import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
for r in cur.execute("select name, count(*), avg(end-start)/1e3 from kernels group by name order by sum(end-start) desc limit 12"):
    print(f"{r[1]:6d} calls  avg {r[2]:8.2f} us  {r[0][:80]}")
