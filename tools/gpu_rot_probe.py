"""Developer probe (GPU box): (1) where the hardware places the four waves of a workgroup (HW_ID of every wave, written
over x[0..3] under CMPC_DEV2=1) and (2) batch time against the wave-role rotation mode CMPC_DEV1."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cmpc_amd as cm


def timed(cfg, P32, X032, reps=5):
    B = P32.shape[0]
    s = cm.BatchSolver(cfg, B)
    dP, dX0 = torch.from_numpy(P32).cuda(), torch.from_numpy(X032).cuda()
    dX, dI = s.solve_device(dP, dX0)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        s.solve_device(dP, dX0, dX, dI)
        torch.cuda.synchronize()
        ms.append(s.last_solve_ms())
    info = dI.cpu().numpy()
    X = dX.cpu().numpy()
    s.close()
    return float(np.median(ms)), info, X


if os.environ.get("PROBE_HWID", "1") == "1":
    os.environ["CMPC_DEV2"] = "1"
    cfg, P, X0 = cm.synthetic.config2_perturbed_com(1024)
    _, info, X = timed(cfg, P.astype(np.float32), X0.astype(np.float32), reps=1)
    hw = X[:, :4].astype(np.int64)
    simd = (hw >> 4) & 3
    slot = hw & 15
    cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
    print("block: SIMD of waves 0..3 | wave slot | cu/sh/se id")
    for b in list(range(12)) + list(range(512, 520)):
        print(b, simd[b].tolist(), slot[b].tolist(), cu[b].tolist())
    pat = {}
    for b in range(hw.shape[0]):
        pat[tuple(simd[b].tolist())] = pat.get(tuple(simd[b].tolist()), 0) + 1
    print("SIMD patterns:", pat)
    sl = {}
    for b in range(hw.shape[0]):
        sl[tuple(slot[b].tolist())] = sl.get(tuple(slot[b].tolist()), 0) + 1
    print("slot patterns:", sl)
    os.environ["CMPC_DEV2"] = "0"

for name, gen, B in (("cfg3", cm.synthetic.config3_external_push, 4096), ("cfg5", cm.synthetic.config5_footstep_candidates, 2048),
                     ("cfg2", cm.synthetic.config2_perturbed_com, 256)):
    cfg, P, X0 = gen(B)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    for mode in os.environ.get("PROBE_MODES", "0,1,2,3,4").split(","):
        os.environ["CMPC_DEV1"] = mode
        ms, info, _ = timed(cfg, P32, X032)
        print(f"{name} B={B} rotation mode {mode}: {ms:.3f} ms  {B / ms * 1e3:.0f} solves/s  iters mean {info[:, 0].mean():.2f} bad {(info[:, 5] != 0).sum()}", flush=True)
