"""From a rocprofv3 --kernel-trace directory of a pipelined global-mode run: the kernel timeline (queue, kind, start, duration; runs of conv
kernels compressed) from 30 ms before to 30 ms after one long beam search.  usage: trace_excerpt.py <dir> [index of the search = -6]"""
import csv, glob, sys
d=sys.argv[1]
p=glob.glob(d+"/*/*_kernel_trace.csv")[0]
ev=[]
for r in csv.DictReader(open(p)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","?")))
ev.sort()
t0=ev[0][0]
dec=[e for e in ev if "beam_search" in e[2] and e[1]-e[0]>30e6]
# pick the search named by argv[2] (index into the long searches; default -6)
d0=dec[int(sys.argv[2]) if len(sys.argv) > 2 else -6]
lo=d0[0]-30e6; hi=d0[1]+30e6
def short(n):
    for k,s in (("tcn_gemm_kernel<2","head"),("tcn_gemm","conv"),("beam_search","BEAM"),("copyBuffer","copy"),("fillBuffer","fill"),("mad_normalise","norm"),("assemble","asm"),("clear_zero","zero")):
        if k in n: return s
    return n[:20]
print(f"decode at {(d0[0]-t0)/1e6:.1f} ms for {(d0[1]-d0[0])/1e6:.1f} ms on queue {d0[3]}")
last={}
for e in ev:
    if e[0]<lo or e[0]>hi: continue
    s=short(e[2])
    if s=="conv":
        # compress: print only first conv after a gap > 5ms on that queue
        if e[3] in last and e[0]-last[e[3]] < 5e6:
            last[e[3]]=e[1]; continue
    last[e[3]]=e[1]
    print(f"  q{e[3]} {s:5s} start {(e[0]-t0)/1e6:9.2f} dur {(e[1]-e[0])/1e6:7.2f}")
