"""bench.py's secondary_reference_defaults leg alone, with a Python stack dump if it stalls.  argv: [n_reads] [extra CLI flags...]"""
import faulthandler, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
faulthandler.dump_traceback_later(45, exit=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
extra = [a for a in sys.argv[1:] if not a.isdigit()]
table = np.random.default_rng(0).dirichlet([0.3] * 4, size=4 ** 11)
from radian_amd import backend as _bk
_close = _bk.Backend.close
def _close_and_tell(self):
    if self._h is not None:
        try:
            print("policy at close:", {op: self.pipe_policy(6, op, True) for op in (1, 2, 3, 0)}, flush=True)
        except Exception as e:
            print("policy read failed:", e)
    _close(self)
_bk.Backend.close = _close_and_tell
import time; t0 = time.time()
r = bench.driver_leg(0, None, ["--rna-threshold", "0.5"] + extra, bench.ragged_lengths(n, 72), 70002, bench.soft_head_weights(), lm=(table, 11), desc="repro")
print("total", time.time() - t0); print({k: v for k, v in r.items() if k != "path"})
