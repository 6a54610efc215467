"""Developer probe (GPU box): wall clock of the class-surface calls for ONE robot (B = 1, host buffers -- the reference's operating mode: CentroidalMPCBlock.cpp:407, 579,
609, 615, 622) against the solve kernel's own duration: what cmpc_set_* / cmpc_advance / cmpc_get_output cost around the launch (pageable copies, synchronisation)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
n = int(os.environ.get("PROBE_N", "200"))
cfg, P, X0 = cm.synthetic.config3_external_push(n, seed=7)     # (swing phase + push: the lists of synthetic._walking_lists)
from cmpc_amd.synthetic import _walking_lists
L = cm.layout.Layout(cfg.N)
lib = cm._capi.lib()
mpc = cm.CentroidalMPC(batch=1)
assert mpc.initialize(cfg), mpc.last_error
h = mpc._h
assert mpc.set_contact_phase_list(_walking_lists(cfg, 6, 8)), mpc.last_error
f0 = np.empty((1, 2, 4, 3), np.float32); p0 = np.empty((1, 2, 3), np.float32); pn = np.empty((1, 2, 3), np.float32); kn = np.empty((1, 2), np.int32)
rows = []
for i in range(n):
    p = P[i].astype(np.float32)
    state = np.ascontiguousarray(p[L.p_com0:L.p_com0 + 9]); com_ref = np.ascontiguousarray(p[L.p_comref:L.p_comref + 3 * (cfg.N + 1)])
    h_ref = np.ascontiguousarray(p[L.p_href:L.p_href + 3 * (cfg.N + 1)])
    t0 = time.perf_counter()
    rc = lib.cmpc_set_state(h, state.ctypes.data, None); assert rc == 0
    rc = lib.cmpc_set_reference(h, com_ref.ctypes.data, h_ref.ctypes.data); assert rc == 0
    t1 = time.perf_counter()
    rc = lib.cmpc_advance(h); assert rc == 0, mpc._solver.last_error
    t2 = time.perf_counter()
    rc = lib.cmpc_get_output(h, f0.ctypes.data, p0.ctypes.data, pn.ctypes.data, kn.ctypes.data); assert rc == 0
    t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, float(lib.cmpc_last_solve_ms(h))))
r = np.array(rows[5:])
m = np.median(r, 0)
print(f"B=1 class surface, {n - 5} ticks (cold starts, config-3 problems): setters {m[0]:.4f} ms | cmpc_advance {m[1]:.4f} ms (its kernel {m[3]:.4f} ms: "
      f"{(m[1] - m[3]) * 1e3:.0f} us around the launch) | cmpc_get_output {m[2]:.4f} ms | whole tick p50 {np.median(r[:, :3].sum(1)):.4f} ms p99 {np.percentile(r[:, :3].sum(1), 99):.4f} ms")
