"""From a rocprofv3 --kernel-trace directory of a pipelined global-mode run: how much of the long beam searches' time the forward's conv
kernels were running (1.0 = the next group's forward fully covers the search), conv durations inside / outside those windows, queues.
usage: decode_overlap_trace.py <dir>   (round 5: found the busy-slot stall of open_slot; profiles/r05_policy_probe.txt)"""
import csv, glob, sys
d=sys.argv[1]
p=glob.glob(d+"/*/*_kernel_trace.csv")[0]
ev=[]
for r in csv.DictReader(open(p)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","?")))
dec=[e for e in ev if "beam_search" in e[2] and e[1]-e[0]>30e6]
conv=sorted([e for e in ev if "tcn_gemm" in e[2]])
tot=0; busy_tot=0
import bisect
starts=[c[0] for c in conv]
for d0,d1,name,q in dec:
    # conv busy time (union) inside [d0,d1]
    i=bisect.bisect_left(starts,d0-50_000_000)
    segs=[]
    for c in conv[i:]:
        if c[0]>d1: break
        a=max(c[0],d0); b=min(c[1],d1)
        if b>a: segs.append((a,b))
    segs.sort(); u=0; cur=None
    for a,b in segs:
        if cur is None: cur=[a,b]
        elif a<=cur[1]: cur[1]=max(cur[1],b)
        else: u+=cur[1]-cur[0]; cur=[a,b]
    if cur: u+=cur[1]-cur[0]
    tot+=d1-d0; busy_tot+=u
q="queue" if any("queue_kernel" in e[2] for e in dec) else "plain"
nq=sum("queue_kernel" in e[2] for e in dec)
print(f"{len(dec)} long beam searches ({nq} through the work queue), {tot/1e6:.0f} ms in total; conv kernels busy during them: {busy_tot/1e6:.0f} ms = {busy_tot/max(1,tot):.2f}")
# conv kernel durations inside / outside the long beam searches (same kernel name only: the 256-channel conv)
import statistics
wins=[(e[0],e[1]) for e in dec]
def inside(t):
    for a,b in wins:
        if a<=t<=b: return True
    return False
main=[c for c in conv if "EPI" not in c[2]]
din=[(c[1]-c[0])/1e3 for c in main if inside((c[0]+c[1])//2)]
dout=[(c[1]-c[0])/1e3 for c in main if not inside((c[0]+c[1])//2)]
if din and dout:
    print(f"conv kernel duration (us): during a long beam search median {statistics.median(din):.0f} (n={len(din)}), otherwise median {statistics.median(dout):.0f} (n={len(dout)})")
qs={}
for c in conv: qs[c[3]]=qs.get(c[3],0)+1
print("conv kernels per queue:", qs, "| beam searches per queue:", {q:sum(1 for e in dec if e[3]==q) for q in set(e[3] for e in dec)})
# per search: start (ms), duration (ms), share of it with a conv kernel running
t00 = min(e[0] for e in ev)
rows = []
for d0, d1, name, q in sorted(dec):
    i = bisect.bisect_left(starts, d0 - 50_000_000)
    segs = []
    for c in conv[i:]:
        if c[0] > d1: break
        a = max(c[0], d0); b = min(c[1], d1)
        if b > a: segs.append((a, b))
    segs.sort(); u = 0; cur = None
    for a, b in segs:
        if cur is None: cur = [a, b]
        elif a <= cur[1]: cur[1] = max(cur[1], b)
        else: u += cur[1] - cur[0]; cur = [a, b]
    if cur: u += cur[1] - cur[0]
    rows.append(f"{(d0 - t00) / 1e6:.0f}:{(d1 - d0) / 1e6:.0f}ms:{u / (d1 - d0):.2f}")
print("per search start:duration:covered ->", " ".join(rows))
