"""Per-rank host budget of a one-process-per-GPU job: which cores a rank's threads may run on, and how many threads it starts.

The reference is one single-threaded process (basecall.py:70-141).  Here a rank runs a driver thread, a read-ahead thread (fast5
parsing), a device thread, a host-stage thread and -- chunk mode -- the native stitch's threads; eight ranks that each size
themselves by the NODE's core count ask for 8 x 16 stitch threads on a machine that may have 64 cores, and the scheduler moves
them across sockets, away from the memory their GPU's DMA engine reads.  So every rank, BEFORE its first GPU call:

  1. takes the cores this process may use (sched_getaffinity: a cgroup / taskset / launcher binding is respected, never widened);
  2. groups the local ranks by the NUMA node of their GPU (/sys/class/kfd/kfd/topology: GPU nodes in HIP's device order ->
     PCI address -> /sys/bus/pci/devices/*/numa_node) and splits each node's usable cores evenly, in contiguous runs, among the
     ranks whose GPU hangs off that node; when the topology cannot be read, or a node has fewer usable cores than ranks, the
     usable cores are split evenly by local rank instead;
  3. binds itself to its slice (threads started later inherit it: the HIP runtime's helpers, the stitch pool, the reader) and derives
     its thread counts from the slice's size -- never from the node's.

Placement rule, in one line: rank r gets cores[node(gpu_r)][i * n / k : (i + 1) * n / k], i = r's index among the k ranks on that node.
`--cpu-affinity none` (or RADIAN_CPU_AFFINITY=none) leaves the process where its launcher put it; the thread counts are then still
derived from usable cores / local world.
"""
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in (text or "").strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return out


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every GPU in HIP's enumeration order (KFD topology order), or None when it cannot be read.  -1 entries: the
    platform reports no affinity for that GPU."""
    top = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        ids = sorted(int(x) for x in os.listdir(top) if x.isdigit())
    except OSError:
        return None
    out = []
    for i in ids:
        props = _read(os.path.join(top, str(i), "properties"))
        if props is None:
            return None
        kv = {}
        for line in props.splitlines():
            p = line.split()
            if len(p) == 2:
                kv[p[0]] = p[1]
        try:
            if int(kv.get("simd_count", "0")) == 0:
                continue                                   # a CPU node
            loc, dom = int(kv["location_id"]), int(kv.get("domain", "0"))
        except (KeyError, ValueError):
            return None
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        node = _read(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node"))
        try:
            out.append(int(node.strip()) if node is not None else -1)
        except ValueError:
            out.append(-1)
    return out or None


def _visible_order(n_gpus):
    """HIP device index -> physical GPU index under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they are plain index lists"""
    order = list(range(n_gpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if not v:
            continue
        try:
            idx = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return None                                    # UUIDs: no mapping without asking the runtime
        if any(i < 0 or i >= len(order) for i in idx):
            return None
        order = [order[i] for i in idx]
    return order


def node_cpus(node, sysfs="/sys"):
    return parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")))


def _even_slice(items, i, k):
    n = len(items)
    return items[i * n // k: (i + 1) * n // k]


def plan(local_rank, local_world, usable=None, gpu_nodes="auto", devices=None, sysfs="/sys"):
    """-> {"cpus": sorted core list of this rank's slice, "numa_node": node or None, "how": "numa" | "even" | "all"}.
    usable: cores this process may use (default: sched_getaffinity).  gpu_nodes: NUMA node per HIP device ("auto": read sysfs).
    devices: HIP device index of every local rank (default: rank r -> device r modulo the devices seen)."""
    if usable is None:
        usable = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    usable = sorted(usable)
    if local_world <= 1:
        return {"cpus": usable, "numa_node": None, "how": "all"}
    even = {"cpus": _even_slice(usable, local_rank, local_world) or usable, "numa_node": None, "how": "even"}
    if gpu_nodes == "auto":
        phys = gpu_numa_nodes(sysfs)
        gpu_nodes = None
        if phys:
            order = _visible_order(len(phys))
            if order:
                gpu_nodes = [phys[i] for i in order]
    if not gpu_nodes:
        return even
    if devices is None:
        # A visible list shorter than the local world means the launcher narrowed every process to its own device(s)
        # (HIP_VISIBLE_DEVICES=<r> per rank): this process cannot see which GPU its peers use, every rank would count all the others as
        # neighbours on ITS node and take 1 / local_world of that node's cores -- on a two-socket box half the cores idle.  Even split.
        if len(gpu_nodes) < local_world:
            return even
        devices = [r for r in range(local_world)]
    nodes = [gpu_nodes[d] if 0 <= d < len(gpu_nodes) else -1 for d in devices]
    if any(n < 0 for n in nodes):
        return even
    mine = nodes[local_rank]
    peers = [r for r in range(local_world) if nodes[r] == mine]
    uset = set(usable)
    cores = [c for c in node_cpus(mine, sysfs) if c in uset]
    # every node that hosts ranks must be able to give each of them a core, or nobody uses the NUMA split (mixed rules could overlap)
    for n in set(nodes):
        k = sum(1 for x in nodes if x == n)
        if len([c for c in node_cpus(n, sysfs) if c in uset]) < k:
            return even
    return {"cpus": _even_slice(cores, peers.index(local_rank), len(peers)), "numa_node": mine, "how": "numa"}


def threads_for(n_cores, decode_type="global"):
    """thread counts of one rank from the size of its slice: the driver, the read-ahead reader, the device thread and the host stage
    are one thread each and mostly wait on each other or on the GPU; what scales is the native stitch (chunk mode)."""
    n = max(1, int(n_cores))
    return {"stitch_threads": max(1, min(16, n - 2)) if decode_type == "chunk" else 0, "reader_threads": 1, "host_stage_threads": 1,
            "total": 4 + (max(1, min(16, n - 2)) if decode_type == "chunk" else 0)}


def _bind_all_threads(cpus):
    """sched_setaffinity(0, ...) binds the CALLING thread only; threads that exist already (a BLAS / OpenMP pool started at `import numpy`
    under a launcher that did not cap it) keep the node-wide mask.  Bind every thread of the process."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    ok = True
    for t in tids or [0]:
        try:
            os.sched_setaffinity(t, cpus)
        except OSError:
            ok = ok and t != 0 and t != os.getpid()     # (a helper thread that ended meanwhile is no failure)
    return ok


def apply(local_rank, local_world, mode="auto", devices=None):
    """Bind this process -- every thread it has so far; later ones inherit -- to its slice (mode 'auto'; 'none': leave it) and return the plan
    with its thread counts.  Call before the first GPU call.  devices: the HIP device of every local rank when it is not rank r -> device r
    (an override such as RD_CLI_DEVICE / RD_BENCH_DEVICE that puts several ranks on one GPU)."""
    mode = os.environ.get("RADIAN_CPU_AFFINITY", mode)
    # (BLAS pools: nothing in a rank calls BLAS; a pool that starts later must not be sized by the node)
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    p = plan(local_rank, local_world, devices=devices)
    if mode != "none" and local_world > 1 and p["cpus"] and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, p["cpus"])
            p["bound"] = _bind_all_threads(p["cpus"])
        except OSError:
            p["bound"] = False
    else:
        p["bound"] = False
    return p
