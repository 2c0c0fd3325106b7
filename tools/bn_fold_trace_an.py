import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "bn_relu" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
out=[]
for r in rows:
    n=r["Kernel_Name"]; k="bwd" if "bwd" in n else "fwd"
    fold = "Lb1" in n.split("apply_kernel")[1][:20] if "_ZN" in n else ("true" in n)
    out.append((k, n[-40:], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, int(r["Grid_Size_X"])//256, r["LDS_Block_Size"]))
for i in range(0,len(out),3):
    print(out[i][0], out[i][1][-28:], " ".join(f"{o[2]:.1f}" for o in out[i:i+3]), out[i][3], out[i][4])
