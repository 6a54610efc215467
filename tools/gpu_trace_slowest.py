"""Developer probe (GPU box): the slowest problem of a bench batch (library CMPC_LIB), re-solved alone with the -DCMPC_PROFILE build CMPC_PROF_LIB for its iteration trace.
   CMPC_LIB=libcmpc_hip_exp.so CMPC_PROF_LIB=libcmpc_hip_prof.so python tools/gpu_trace_slowest.py config5_footstep_candidates 8192"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm  # noqa: E402

gen, B = sys.argv[1], int(sys.argv[2])
cfg, P, X0 = getattr(cm.synthetic, gen)(B)
P32, X032 = P.astype(np.float32), X0.astype(np.float32)
if len(sys.argv) > 3:
    b = int(sys.argv[3])
    s1 = cm.BatchSolver(cfg, 1, factors="hbm")
    X1, info1, rc = s1.solve_host(P32[b:b + 1], X032[b:b + 1])
    tr = (C.c_float * 512)()
    cm._capi.lib().cmpc_trace_read(tr)
    tr = np.array(tr[:]).reshape(64, 8)
    print(f"problem {b} alone: iterations {int(info1[0, 0])} status {int(info1[0, 5])} safeguards {int(info1[0, 3])}")
    print("      mu_cur      ep       ec(max tz)  step     ap    ad    sigma    mu_t")
    for i in range(min(int(info1[0, 0]), 64)):
        print("  it %2d  %.3e %.3e %.3e %.3e %.4f %.4f %.3e %.3e" % ((i,) + tuple(tr[i])))
else:
    s = cm.BatchSolver(cfg, B)
    X, info, rc = s.solve_host(P32, X032)
    order = np.argsort(-info[:, 0])
    print("slowest:", [(int(b), int(info[b, 0]), int(info[b, 3])) for b in order[:5]], "mean %.3f" % info[:, 0].mean())
    env = dict(os.environ, CMPC_LIB=os.environ["CMPC_PROF_LIB"])
    subprocess.run([sys.executable, __file__, gen, str(B), str(int(order[0]))], env=env)
