"""Developer probe (GPU box): iteration trace (mu, residuals, step, step lengths, sigma) of ONE problem of a generator, solved alone (workgroup 0, resident variant) with the
-DCMPC_PROFILE build named by CMPC_LIB.   CMPC_LIB=libcmpc_hip_prof.so python tools/gpu_trace_one.py config3_external_push 32 44 8 [N]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm  # noqa: E402

gen, B, seed, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = {"N": int(sys.argv[5])} if len(sys.argv) > 5 else {}
cfg, P, X0 = getattr(cm.synthetic, gen)(B, seed=seed, **kw)
P32, X032 = P.astype(np.float32), X0.astype(np.float32)
s1 = cm.BatchSolver(cfg, 1)
X1, info1, rc = s1.solve_host(P32[b:b + 1], X032[b:b + 1])
tr = (C.c_float * 512)()
cm._capi.lib().cmpc_trace_read(tr)
tr = np.array(tr[:]).reshape(64, 8)
print(os.environ.get("CMPC_LIB"), f"problem {b}: iterations {int(info1[0, 0])} status {int(info1[0, 5])} safeguards {int(info1[0, 3])}")
print("      mu_cur      ep       ec(max tz)  step     ap    ad    sigma    mu_t")
for i in range(int(info1[0, 0])):
    print("  it %2d  %.3e %.3e %.3e %.3e %.4f %.4f %.3e %.3e" % ((i,) + tuple(tr[i])))
