import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmpc_amd as cm
cm._capi.LIB_PATH = os.path.join(os.path.dirname(cm._capi.LIB_PATH), os.environ["DBG_LIB"])
N = int(os.environ.get("DBG_N", "13"))
cfg, P, X0 = cm.synthetic.config3_external_push(400, N=N, seed=44)
pb = int(os.environ.get("DBG_PROB", "0"))
s = cm.BatchSolver(cfg, 1, max_iterations=int(os.environ.get("DBG_IT", "40")), final_extrapolation=False)
X, info, rc = s.solve_host(P[pb:pb + 1].astype(np.float32), X0[pb:pb + 1].astype(np.float32))
print(os.environ["DBG_LIB"], "N", N, "info", info[0])
np.save(os.path.join("gpurun_out", "dbgX_" + os.environ["DBG_LIB"] + "_it" + os.environ.get("DBG_IT", "40") + ".npy"), X[0])
buf = (C.c_float * (48 * 1585))()
cm._capi.lib().cmpc_dbg_read(buf)
np.save(os.path.join("gpurun_out", "dbgP_" + os.environ["DBG_LIB"] + ".npy"), np.array(buf[:]).reshape(48, 1585))
