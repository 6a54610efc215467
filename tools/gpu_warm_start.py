"""Developer probe: receding-horizon ticks through the class surface (state fed back from the plant), cold vs warm start."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cmpc_amd as cm
from cmpc_amd.synthetic import _standing_lists, _walking_lists
B = 64
for sched_name in ("standing", "walking"):
  for mode in ["cold"] + sys.argv[1:]:
    if mode != "cold": os.environ["CMPC_MU_WARM"] = mode
    cfg, P, X0 = (cm.synthetic.config2_perturbed_com if sched_name == "standing" else cm.synthetic.config3_external_push)(B)
    L = cm.Layout(cfg.N); N = cfg.N
    mpc = cm.CentroidalMPC(batch=B); assert mpc.initialize(cfg)
    solver = mpc._solver
    state = P[:, L.p_com0:L.p_com0 + 9].astype(np.float32).copy()
    its = []
    for tick in range(8):
        t0 = tick * cfg.sampling_time
        assert mpc.set_state(state[:, 0:3], state[:, 3:6], state[:, 6:9])
        assert mpc.set_reference_trajectory(P[:, L.p_comref:L.p_comref + 3 * (N + 1)], P[:, L.p_href:L.p_href + 3 * (N + 1)])
        lists = _standing_lists(cfg, 10.0) if sched_name == "standing" else _walking_lists(cfg, 6, 8)
        assert mpc.set_contact_phase_list(lists, t0=t0)
        if tick > 0 and mode != "cold": assert mpc.set_initial_guess(None, shift_previous=True)
        assert mpc.advance(), mpc.last_error
        X, info = mpc.get_solution(); its.append(info[:, 0].mean())
        # plant: one MPC period
        dX = torch.from_numpy(X).cuda(); dP = torch.from_numpy(np.zeros((B, L.np), np.float32)).cuda()
        Pcur = np.zeros((B, L.np), np.float32); Pcur[:] = P.astype(np.float32)   # R, gamma of knot 0 (schedule at t0 not shifted here: standing ok)
        dP = torch.from_numpy(Pcur).cuda()
        st, zmp = solver.plant_step_device(dX, dP, torch.from_numpy(state).cuda(), step=0.01, substeps=6)
        torch.cuda.synchronize(); state = st.cpu().numpy()
    print(sched_name, mode, "mean iterations per tick:", ["%.1f" % v for v in its], flush=True)
