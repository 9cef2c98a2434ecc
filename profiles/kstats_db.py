import sqlite3,glob,sys
for db in glob.glob(sys.argv[1]+'/**/*.db', recursive=True):
    c=sqlite3.connect(db)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q=f"select s.kernel_name, d.grid_size_x, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x order by 4 desc"
    for r in c.execute(q): print("%-60s grid %8d n=%3d avg %9.1f us min %9.1f"%(r[0][:60],r[1],r[2],r[3],r[4]))
