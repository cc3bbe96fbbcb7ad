"""From a rocprofv3 --kernel-trace directory: per hardware queue, the kernels longer than 30 ms and the pauses longer than 60 ms between
consecutive kernels -- a forward lane that stands still while a beam search runs shows as a pause as long as that search.
usage: lane_gap_trace.py <dir>"""
import csv, glob, sys
d=sys.argv[1]
p=glob.glob(d+"/*/*_kernel_trace.csv")[0]
ev=[]
for r in csv.DictReader(open(p)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id","?")))
ev.sort()
t0=ev[0][0]
# per queue: find gaps > 50 ms between consecutive kernels while others run; and kernels with duration > 30 ms
byq={}
for e in ev: byq.setdefault(e[3],[]).append(e)
for q,l in byq.items():
    names={}
    for e in l: names[e[2]]=names.get(e[2],0)+1
    print("queue",q,len(l),"kernels",sorted(names.items(), key=lambda x:-x[1])[:3])
    long=[e for e in l if e[1]-e[0]>30e6]
    print("   kernels > 30 ms:", len(long), [(round((e[0]-t0)/1e6), round((e[1]-e[0])/1e6), e[2][:22]) for e in long[:8]])
    gaps=[(round((l[i][1]-t0)/1e6), round((l[i+1][0]-l[i][1])/1e6)) for i in range(len(l)-1) if l[i+1][0]-l[i][1]>60e6]
    print("   gaps > 60 ms (at ms, length):", gaps[:12])
