import sys, csv, glob, collections
d=sys.argv[1]
f=glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
fam=collections.OrderedDict()
steps=float(sys.argv[2]) if len(sys.argv)>2 else 20.0
for r in rows:
    n=r["Name"]
    for k in ("bn_relu_bwd_apply","bn_relu_apply_pool","bn_relu_apply","bn_relu_bwd_reduce","bn_finalize","bn_bwd_finalize","bn_fold","bn_running","conv3x3_flow","conv3x3_stream","conv3x3_plane","wgrad12s","wgrad12_","wgrad_reduce","maxpool2_bwd","upsample2_bwd","conv_splitk"):
        if k in n:
            a=fam.setdefault(k,[0,0.0]); a[0]+=int(r["Calls"]); a[1]+=float(r["TotalDurationNs"]); break
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print(f"total {tot/1e6/steps:.3f} ms/step, {calls/steps:.0f} launches/step")
for k,(c,t) in fam.items(): print(f"{k:22s} {c/steps:6.1f} launches  {t/1e6/steps:7.3f} ms/step  {t/1e3/c:7.1f} us avg")
