import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]
ksym=[t for t in tabs if 'info_kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6, s.arch_vgpr_count, s.accum_vgpr_count from {kd} d join {ksym} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
for r in c.execute(q): print(r[0][:70], r[1], round(r[2],3), round(r[3],3), r[4], r[5])
