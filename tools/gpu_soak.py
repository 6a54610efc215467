"""Developer probe: many seeded batches of every config; counts problems that do not converge, reports the slowest problem (seed, index) and asserts that no wave of
the streaming stage ever gave up at a hand-off word.   python tools/gpu_soak.py [B] [seeds] [auto|lds|hbm]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import cmpc_amd as cm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seeds = range(100, 100 + (int(sys.argv[2]) if len(sys.argv) > 2 else 10))
factors = sys.argv[3] if len(sys.argv) > 3 else "auto"
print("# factors =", factors)
for name, gen in (("cfg2", cm.synthetic.config2_perturbed_com), ("cfg3", cm.synthetic.config3_external_push),
                  ("cfg4", cm.synthetic.config4_monte_carlo), ("cfg5", cm.synthetic.config5_footstep_candidates)):
    if factors == "lds" and name == "cfg5":
        continue      # (N = 30 does not fit LDS)
    tot = bad = fb = pol = rst = giveups = 0
    slow = (0, -1, -1)
    hist = np.zeros(64, int)
    itmax = 0
    its = []
    s = None
    for sd in seeds:
        cfg, P, X0 = gen(B, seed=sd)
        if s is None:
            s = cm.BatchSolver(cfg, B, factors=factors)
        X, info, rc = s.solve_host(P.astype(np.float32), X0.astype(np.float32))
        sgw = info[:, 3].astype(np.int64)     # safeguard word (include/cmpc.h): fallbacks + 100 re-centrings + 10000 cold restart + 100000 tail polished
        giveups += int((sgw // 1000000).sum()); sgw = sgw % 1000000
        b = int(np.argmax(info[:, 0]))
        if info[b, 0] > slow[0]:
            slow = (int(info[b, 0]), sd, b)
        fb += int((sgw % 10000 > 0).sum()); rst += int(((sgw // 10000) % 10 > 0).sum()); pol += int((sgw // 100000 > 0).sum()); hist += np.bincount(np.minimum(info[:, 0].astype(int), 63), minlength=64)
        tot += B; bad += int((info[:, 5] != 0).sum()); itmax = max(itmax, int(info[:, 0].max())); its.append(info[:, 0].mean())
        assert np.isfinite(X).all()
    print(name, "problems", tot, "not converged", bad, "iterations mean %.2f max %d" % (np.mean(its), itmax), "| problems with a Gauss-Newton fallback or a re-centring", fb, "cold restarts", rst, "tail polished", pol,
          "| iterations >= 14:", int(hist[14:].sum()), "histogram 3..20:", hist[3:21].tolist(), "| slowest: %d iterations, seed %d, problem %d" % slow, "| sync give-ups", giveups, flush=True)
    assert giveups == 0
    s.close()
