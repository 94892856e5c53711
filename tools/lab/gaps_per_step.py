import csv,glob,re,sys
kt=sorted(csv.DictReader(open(glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0])), key=lambda r:int(r["Start_Timestamp"]))
def short(n):
    n=n.replace("(anonymous namespace)::","").replace("void ","").replace("mt::","")
    return re.sub(r"\(.*","",n)[:40]
starts=[i for i,r in enumerate(kt) if "stem_mfma_kernel" in r["Kernel_Name"]]
print(len(starts),"steps")
for k in range(2,len(starts)-1):
    seg=kt[starts[k]:starts[k+1]+1]
    t0=int(seg[0]["Start_Timestamp"]); span=(int(seg[-1]["Start_Timestamp"])-t0)/1e6
    ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),short(r["Kernel_Name"])) for r in seg)
    end=ev[0][1]; gaps=[]; last=ev[0]; tot=0
    for s,e,n in ev[1:]:
        if s-end>3000: tot+=s-end
        if s-end>15000: gaps.append((round((s-end)/1e3,1), round((end-t0)/1e6,2), last[2][:24], n[:24]))
        if e>end: end,last=e,(s,e,n)
    print(f"step {k}: {span:.2f} ms, idle(>3us) {tot/1e6:.2f} ms; >15us:", gaps[:5])
