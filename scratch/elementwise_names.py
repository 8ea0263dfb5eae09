import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda:[0,0])
for r in rows[len(rows)//2:]:
    n=r["Kernel_Name"]
    if "elementwise_kernel_manual_unroll" in n or "copyBuffer" in n:
        agg[(n[100:330], r["Grid_Size_X"])][0]+=1
        agg[(n[100:330], r["Grid_Size_X"])][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,(c,t) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:8]:
    print(f"{t/c/1e3:8.1f} us x {c:4d} grid {k[1]:>10s}  {k[0]}")
