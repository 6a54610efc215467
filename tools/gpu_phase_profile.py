"""Diagnostic (GPU box): per-phase shader-clock shares of the solver kernel, from the
-DCMPC_PROFILE build (libcmpc_hip_prof.so).  Shares only; never quote its run time."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (same HIP runtime)
import cmpc_amd as cm
cm._capi.LIB_PATH = os.path.join(os.path.dirname(cm._capi.LIB_PATH), os.environ.get("CMPC_PROF_LIB", "libcmpc_hip_prof.so"))
# (resident variants, streaming square-root stage: slots 1, 2, 4..8 are unused, slot 3 is wave 0's wait at the stage barrier, slot 9 its factorisation; the HBM-factor
#  variants and a -DCMPC_SQRT_BACKWARD=0 build fill the ph1..ph4 slots instead)
names = ["(ph0: under ph3 now)", "ph1 G", "ph2 Quu,panel,Pd,qu", "ph3 chol+solve | sq: barrier wait", "ph4 P update", "ph2/w0: values", "ph2/w0: stores", "ph2/w2: diag blocks", "ph2/w3: qu", "sq: factorisation (w0)",
         "residuals", "backward(total)", "fwd1", "steps+muaff", "delta", "fwd2", "steplen+costate", "update+conv"]
kw = {}
if os.environ.get("CMPC_PROBE") == "push":
    cfg, P, X0 = cm.synthetic.walking_push_n12("tmp")
elif os.environ.get("CMPC_PROBE") == "config3":    # HBM-factor variant, three workgroups on every CU
    cfg, P, X0 = cm.synthetic.config3_external_push(768)
    kw = dict(factors="hbm")
elif os.environ.get("CMPC_PROBE") == "config5":
    cfg, P, X0 = cm.synthetic.config5_footstep_candidates(768)
    kw = dict(factors="hbm")
else:
    cfg, P, X0 = cm.synthetic.config2_perturbed_com(256)
s = cm.BatchSolver(cfg, P.shape[0], **kw)
X, info, rc = s.solve_host(P, X0)
out = (C.c_longlong * 128)()
cm._capi.lib().cmpc_profile_read(out, 1)
v = np.array(out[:18], float)
tot = v[10:18].sum()
print("iters block0", info[0, 0], "total cycles (sum of phases) %.3g" % tot, "kernel cycles %.3g" % info[0, 6])
if out[31]:
    print("backward passes of workgroup 0: %d -> %.0f cycles per stage of a pass" % (out[31], out[63] / (out[31] * cfg.N)))
for _ in range(6):
    _, info2, _ = s.solve_host(P, X0)
print("batch (7th launch): iterations max %d, kernel %.4f ms; slowest block %.4g cycles -> %.0f MHz shader clock" % (info2[:, 0].max(), s.last_solve_ms(), info2[:, 6].max(), info2[:, 6].max() / s.last_solve_ms() / 1e3))
for n, x in zip(names, v):
    print("%-22s %12.0f  %5.1f%%  per stage-iter %8.0f" % (n, x, 100 * x / tot, x / (info[0, 0] * cfg.N)))

print("  ph4 (MFMA build): tiles on wave 0 %8.0f, value gradient on its wave %8.0f per stage-iter" % (out[18] / (info[0, 0] * cfg.N), out[19] / (info[0, 0] * cfg.N)))
sub = np.array(out[20:31], float)
for n, x in zip(["fwd: loop head", "fwd: y = lq + Ws ds + Wp dp", "fwd: du = -Linv^T y", "fwd: AB_step",
                 "delta: loop head", "delta: rhs g + B^T fp", "delta: dl = Linv dq", "delta: fp update", "ph3: load rows", "ph3: cholesky+solve", "ph3: store"], sub):
    print("  %-30s %12.0f  per stage-sweep %8.0f" % (n, x, x / (info[0, 0] * cfg.N)))
if any(out[32:62]):
    den = out[31] * cfg.N
    print("  consumer waves 1, 2, 3, 5, 6, 7, cycles from entry (per stage of a pass): entry | assembly | operand rows | blocks 0-7 | exit")
    for w in range(6):
        print("    wave %d: " % (w + 1 if w < 3 else w + 2) + " ".join("%6d" % (out[32 + 5 * w + i] / den) for i in range(5)))
if any(out[64:112]):
    den = out[31] * cfg.N
    print("  finer: Y written | Y wait over | float64 part | [assembly] | assembly-count wait over | blocks 8, 9 | [exit]")
    for w in range(6):
        print("    wave %d: " % (w + 1 if w < 3 else w + 2) + " ".join("%6d" % (out[64 + 8 * w + i] / den) for i in range(5)))
if any(out[112:120]):
    den = out[31] * cfg.N
    print("  wave 4: descriptors done | blocks 0-3 | blocks 4-7 | gradient column applied: " + " ".join("%6d" % (out[112 + i] / den) for i in range(4)))
tr = (C.c_float * 512)()
cm._capi.lib().cmpc_trace_read(tr)
tr = np.array(tr[:]).reshape(64, 8)
if os.environ.get("CMPC_DUMPW"):
    np.set_printoptions(linewidth=250, precision=5)
    print("published rows, sum |x| over blocks (first factorised stage, first pass):")
    print(tr[56:62].reshape(-1)[:46])
print("iteration trace of problem 0:  mu_cur      ep       ec(max tz)  step     ap    ad    sigma    mu_t")
for i in range(int(info[0, 0])):
    print("  it %2d  %.2e %.2e %.2e %.2e %.3f %.3f %.2e %.2e" % ((i,) + tuple(tr[i])))
