import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
for name, typ, sql in con.execute("select name, type, sql from sqlite_master where name like '%kernel%' or name like 'top%'"):
    print(typ, name, (sql or "")[:1500].replace("\n", " "))
    print()
print(con.execute("select count(*) from top_kernels").fetchone())
for row in con.execute("select S.kernel_name, S.display_name, count(*), sum(K.end-K.start)/1000.0 from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid where S.kernel_name like '%softmin%' or S.kernel_name like '%random%' or S.kernel_name like '%leading%' group by S.kernel_name"):
    print(row)
print("distinct symbols", con.execute("select count(distinct kernel_name), count(distinct display_name) from rocpd_info_kernel_symbol").fetchone())
