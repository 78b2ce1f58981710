import sqlite3,glob,sys
db=glob.glob(sys.argv[1])[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
cols=[r[1] for r in c.execute(f"pragma table_info({kd})")]
gz='d.grid_size_z' if 'grid_size_z' in cols else ('d.grid_z' if 'grid_z' in cols else '0')
gx='d.grid_size_x' if 'grid_size_x' in cols else ('d.grid_x' if 'grid_x' in cols else '0')
q=f"select s.kernel_name, {gx}, {gz}, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name, {gx}, {gz} order by min(d.start)"
for r in c.execute(q): print("%-70s grid=(%s,%s) n=%d avg=%.0f min=%d max=%d"%(r[0][:70],r[1],r[2],r[3],r[4],r[5],r[6]))
