import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    short="R" if "create_requests" in n else "H" if "handle_vis" in n else "I" if "integrate" in n else "T" if "compute_points" in n else "N" if "compute_normals" in n else None
    if short: rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),short,r.get("Queue_Id","?")))
rows.sort()
t0=rows[400][0]
for s,e,k,q in rows[400:430]:
    print(f"{k} q{q} start {(s-t0)/1e3:8.1f} end {(e-t0)/1e3:8.1f} dur {(e-s)/1e3:6.1f}")
