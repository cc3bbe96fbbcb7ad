"""The shader clock the chip holds INSIDE the conv kernel during a sustained pipelined run (MI355X_MICROARCH.md "DVFS give-back" (6):
delta s_memtime / delta s_memrealtime per workgroup; rocm-smi's sclk is not the test), with and without the decode partition.
Needs the stamped diagnostic build:  python -m radian_amd.build -DRD_CLOCK_STAMPS -oradian_amd/variants/lib_clock.so
usage (GPU box): RADIAN_HIP_LIB=$PWD/radian_amd/variants/lib_clock.so python tools/conv_clock.py [partK ...]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from radian_amd import Backend, synthetic, weights, _lib


def main():
    n, L, W = 64, 4096, 1
    lib = ctypes.CDLL(_lib.LIB_PATH)
    stamped = hasattr(lib, "rd_debug_conv_stamps")     # (without the diagnostic build: just the loop, e.g. under rocprofv3)
    be = Backend(0)
    be.load_weights(weights.synthetic_weights(seed=1234))
    be.load_lm(np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** 5), 5)
    bufs = []
    for b in range(4):
        norm = np.stack([synthetic.mad_normalise(r, 4) for r in synthetic.synthetic_reads(n, L, seed=1000 + b)]).astype(np.float32)
        d = be.dev_alloc(norm.nbytes); be.h2d(d, norm); bufs.append(d)
    off = np.arange(n + 1, dtype=np.int64) * L
    lab_off = np.ascontiguousarray(off[:-1])
    ring = [(np.zeros(n * L + 1, np.uint8), np.zeros(n, np.int32)) for _ in range(40)]
    parts = [int(a[4:]) for a in sys.argv[1:] if a.startswith("part")] or [0, 4, 8]
    chunk = "chunk" in sys.argv      # the headline's loop instead: chunk-mode W = 10 (or "w1"), beam search on the whole chip beside the forwards
    fwdonly = "fwdonly" in sys.argv  # round 4: the forward alone on two lanes (rd_forward_reads_resident), no beam search anywhere
    if fwdonly:
        parts = [-1]
    if chunk:
        parts = [-1]
        Wc = 1 if "w1" in sys.argv else 10
        be.pipe_config(8); be.pipe_set_lanes(2)
        outs = [(np.zeros((n * 8, 1024), dtype=np.uint8), np.full(n * 8, -1, dtype=np.int32)) for _ in range(16)]
    for part in parts:
        if not chunk and not fwdonly:
            be.pipe_flush(); be.set_decode_partition(part); be.pipe_config(3)

        def run(k):
            if fwdonly:
                for i in range(k):
                    be.forward_reads_resident(bufs[i % 4], off, n, 1024, 512, "chunk", lane=i % 2)
                be.sync()
                return
            if chunk:
                th, tmin = 0.0, []
                for i in range(k):
                    lab, ln = outs[i % 16]
                    t_ = time.perf_counter()
                    be.pipe_submit_reads(bufs[i % 4], off, n, 1024, 512, Wc, lab, ln)
                    th += time.perf_counter() - t_
                    tmin.append(time.perf_counter() - t_)
                be.pipe_flush(); be.sync()
                if k >= 100:
                    print(f"  host time inside pipe_submit_reads: mean {th / k * 1e6:.0f} us per step (includes waiting for a free slot: the host runs ahead of the device), "
                          f"median of the fastest quarter {np.median(np.sort(tmin)[: k // 4]) * 1e6:.0f} us (13 launches + an event record)")
                return
            for i in range(k):
                lab, ln = ring[i % len(ring)]
                be.pipe_submit_reads_global(bufs[i % 4], off, n, 1024, 512, W, True, 0.5, 0.5, lab, lab_off, ln)
                if i >= 32:
                    be.pipe_progress(be.pipe_submitted() - 32)
            be.pipe_flush(); be.sync()
        run(8)
        t0 = time.perf_counter(); run(300); dt = time.perf_counter() - t0     # ~3 s of back-to-back launches
        if not stamped:
            print(f"partition {part}: {300 * n * L / dt / 1e6:6.2f} M samples/s (no stamps in this build)")
            continue
        st = np.zeros(32 * 4096 * 4, dtype=np.uint64)
        rc = lib.rd_debug_conv_stamps(st.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)))
        st = st.reshape(-1, 4)
        if "dump" in sys.argv:
            np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"conv_stamps_part{part}.npy"), st)
        st = st[(st[:, 0] > 0) & (st[:, 1] > st[:, 0])]
        t0s, t1s, clk, where = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64), st[:, 2].astype(np.float64), st[:, 3]
        dur = (t1s - t0s).astype(np.float64)          # 100-MHz ticks
        lo, hi = np.percentile(t0s, 30), np.percentile(t1s, 70)      # a window the ring covers completely (every launch that overlaps it is in it)
        hw = (where & np.uint64(0xffffffff)).astype(np.int64)
        cu_key = (((where >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
        occ, gaps = [], []
        for key in np.unique(cu_key):
            m = cu_key == key
            a, b = t0s[m], t1s[m]
            occ.append(np.clip(np.minimum(b, hi) - np.maximum(a, lo), 0, None).sum() / (2.0 * (hi - lo)))
            order = np.argsort(a)
            slot_end = [None, None]                  # two workgroups per CU: give each start the slot that freed last before it
            for i in order:
                if a[i] < lo or a[i] > hi:
                    k = 0 if slot_end[0] is None or (slot_end[1] is not None and slot_end[0] <= slot_end[1]) else 1
                    slot_end[k] = b[i]
                    continue
                cands = [k for k in (0, 1) if slot_end[k] is not None and slot_end[k] <= a[i] + 50]
                if cands:
                    k = max(cands, key=lambda kk: slot_end[kk])
                    gaps.append((a[i] - slot_end[k]) / 100.0)
                else:
                    k = 0 if slot_end[0] is None or (slot_end[1] is not None and slot_end[0] <= slot_end[1]) else 1
                slot_end[k] = b[i]
        gaps = np.array(gaps) if gaps else np.zeros(1)
        if "launches" in sys.argv:   # per launch (output buffer, dilation, epilogue) inside 5 ms of the window: when its workgroups started and ended
            tag = (where >> np.uint64(36)).astype(np.int64)
            m = (t0s >= lo) & (t0s < lo + 500000)
            rows = []
            for tg in np.unique(tag[m]):
                mm = m & (tag == tg)
                a, b = np.sort(t0s[mm]), t1s[mm]
                # the same layer of the same lane comes back every step: split at pauses of > 1 ms between starts
                cuts = np.flatnonzero(np.diff(a) > 100000)
                for seg in np.split(np.arange(len(a)), cuts + 1):
                    st = a[seg]
                    en = np.sort(t1s[mm][np.argsort(t0s[mm])][seg])
                    rows.append((st[0], tg, len(seg), st[0], st[-1], en[-1], en[len(en) // 2]))
            rows.sort()
            print("  lane(out)  dil epi   wgs   first_start  last_start  last_end (us from the window's start)   dispatch_span  tail")
            for _, tg, nw, f, l, e, _ in rows:
                print(f"  {tg >> 12 & 0xffff:#06x} {tg >> 4 & 0xff:4d} {tg & 0xf:3d} {nw:6d} {(f - lo) / 100:12.0f} {(l - lo) / 100:11.0f} {(e - lo) / 100:9.0f} {(l - f) / 100:30.0f} {(e - l) / 100:6.0f}")
            ts = lo + np.arange(250) * 2000
            act = [(int(((t0s <= t) & (t1s > t)).sum())) for t in ts]
            print("  in flight every 20 us from the window's start:", " ".join(str(x) for x in act))
        if "gantt" in sys.argv:      # one CU's workgroups over 4 ms: (start, duration) in us from the window's start
            for key in np.unique(cu_key)[[0, 77]]:
                m = (cu_key == key) & (t0s >= lo) & (t0s < lo + 400000)
                o = np.argsort(t0s[m])
                print(f"  CU {key:#x}: " + " ".join(f"{(a - lo) / 100:.0f}+{(b - a) / 100:.0f}" for a, b in zip(t0s[m][o], t1s[m][o])))
        if "gaps" in sys.argv:
            g = np.sort(gaps)
            tot = g.sum()
            print("  gaps (us) p50/p75/p90/p95/p99/max: " + " / ".join(f"{np.percentile(g, q):.1f}" for q in (50, 75, 90, 95, 99, 100))
                  + f"; share of the idle slot time in gaps > 30 us: {g[g > 30].sum() / tot:.2f} ({(g > 30).mean() * 100:.1f} % of the gaps), > 100 us: {g[g > 100].sum() / tot:.2f}")
        if "series" in sys.argv:      # workgroups in flight, sampled every 20 us over 4 ms of the window (of 2 x CUs slots)
            ts = lo + (hi - lo) * 0.4 + np.arange(200) * 2000
            act = [(int(((t0s <= t) & (t1s > t)).sum())) for t in ts]
            print("  in flight every 20 us:", " ".join(str(a) for a in act))
        print(f"partition {part} CUs per XCD: {300 * n * L / dt / 1e6:6.2f} M samples/s, {dt / 300 / (n * L) * 1e9:.2f} ns per row | {len(np.unique(cu_key))} CUs ran conv; "
              f"in-kernel clock {np.median(clk / dur) * 0.1:.3f} GHz; conv workgroup median {np.median(dur) / 100:.1f} us (p10 {np.percentile(dur, 10) / 100:.1f}, "
              f"p90 {np.percentile(dur, 90) / 100:.1f}); slot occupancy (2 per CU) over a {(hi - lo) / 1e5:.1f}-ms window: mean {np.mean(occ):.3f}; "
              f"gap between workgroups on a slot: median {np.median(gaps):.1f} us, mean {np.mean(gaps):.1f} us, p90 {np.percentile(gaps, 90):.1f} us (rc {rc})", flush=True)
    be.close()


main()
