import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
out=[]
for r in rows:
    n=r["Kernel_Name"]
    if "ihp::" not in n: continue
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    out.append((s,e,n.split("ihp::")[1][:28], r.get("Workgroup_Size_X") or r.get("Workgroup_Size"), r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("LDS_Block_Size")))
last=out[-int(sys.argv[2]):]
t0=last[0][0]
for s,e,n,w,g,l in last: print("%8.3f -> %8.3f ms  dur %7.3f  %-28s grid %s lds %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n,g,l))
