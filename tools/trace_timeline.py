import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sel=rows[-int(sys.argv[2]):]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    nm=r["Kernel_Name"]
    nm=nm.replace("_ZN8tcnn_amd12_GLOBAL__N_1","").replace("void tcnn_amd::(anonymous namespace)::","").replace("tcnn_amd::(anonymous namespace)::","")
    print("%-30s q=%s  %8.1f -> %8.1f  (%6.1f us)"%(nm[:30], r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
