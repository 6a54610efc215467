"""Synthetic problem batches for the five BASELINE.json configurations.

Every generator returns (cfg, P[B,n_p], X0[B,n_x]) in float64, in the reference's x/p layout
(layout.py).  Distributions, seeds and the robot mass constant are the ones SURVEY 8d states; the
URDF is not in the reference tree, so the mass used to normalise pushes is an explicit constant.
"""
from __future__ import annotations

import numpy as np

from . import config as _cfg
from .contacts import PlannedContact, pack_lists, sample_schedule, sample_schedule_batch
from .layout import Layout, cold_start, pack_parameters

ROBOT_MASS = 56.0  # kg, stated constant
FOOT_Y = 0.08


def _tile(d, B):
    return {k: np.broadcast_to(v, (B,) + v.shape).copy() for k, v in d.items()}


def _standing_lists(cfg, horizon_end):
    return {cfg.contacts[0].contact_name: [PlannedContact(-1.0, horizon_end + 1.0, (0.0, FOOT_Y, 0.0))],
            cfg.contacts[1].contact_name: [PlannedContact(-1.0, horizon_end + 1.0, (0.0, -FOOT_Y, 0.0))]}


def _perturbed_state(rng, B, com_nominal):
    com0 = np.asarray(com_nominal)[None, :] + rng.uniform(-0.02, 0.02, (B, 3))
    dcom0 = rng.uniform(-0.1, 0.1, (B, 3))
    h0 = rng.uniform(-0.05, 0.05, (B, 3))
    return com0, dcom0, h0


def _finish(cfg, sched, com0, dcom0, h0, com_ref, h_ref, f_ext=None):
    P = pack_parameters(cfg.N, sched["R"], sched["upper"], sched["lower"], sched["enabled"],
                        sched["nominal"], sched["current"], com0, dcom0, h0, com_ref, h_ref, f_ext)
    return cfg, P, cold_start(cfg.N, P)


def config1_plumbing():
    """Single iCub3 standing MPC, horizon=10, 2 contacts (CPU-runnable plumbing case)."""
    cfg = _cfg.icub_gazebo_v3(10, 0.1)
    B, N = 1, cfg.N
    sched = _tile(sample_schedule(cfg, _standing_lists(cfg, N * cfg.sampling_time)), B)
    com0 = np.array([[0.0, 0.0, 0.53]])
    ref = np.broadcast_to(np.array([0.0, 0.0, 0.53]), (B, N + 1, 3)).copy()
    return _finish(cfg, sched, com0, np.zeros((B, 3)), np.zeros((B, 3)), ref, np.zeros((B, N + 1, 3)))


def config2_perturbed_com(B=256, N=20, seed=0):
    """Batch of perturbed-CoM standing problems, ergoCubGazeboV1 parameters (the bench workload)."""
    cfg = _cfg.ergocub_gazebo_v1(N, 0.06)
    rng = np.random.default_rng(seed)
    sched = _tile(sample_schedule(cfg, _standing_lists(cfg, N * cfg.sampling_time)), B)
    com0, dcom0, h0 = _perturbed_state(rng, B, (0.0, 0.0, 0.7))
    ref = np.broadcast_to(np.array([0.0, 0.0, 0.7]), (B, N + 1, 3)).copy()
    return _finish(cfg, sched, com0, dcom0, h0, ref, np.zeros((B, N + 1, 3)))


def _walking_lists(cfg, swing_start, swing_len, step=0.1):
    dt = cfg.sampling_time
    t_end = cfg.N * dt
    left = [PlannedContact(-1.0, swing_start * dt, (0.0, FOOT_Y, 0.0)),
            PlannedContact((swing_start + swing_len) * dt, t_end + 1.0, (step, FOOT_Y, 0.0))]
    right = [PlannedContact(-1.0, t_end + 1.0, (0.0, -FOOT_Y, 0.0))]
    return {cfg.contacts[0].contact_name: left, cfg.contacts[1].contact_name: right}


def config3_external_push(B=4096, N=20, seed=1, shard=None):
    """Walking schedule with a swing phase inside the horizon (left Gamma = 1x6, 0x8, 1x6) so the
    step adjustment is active, plus an external push U(-50,50) N in x,y over the first 0.2 s.
    shard=(lo, hi): only problems [lo, hi) of the B-problem batch are built (the random draws are those
    of the whole batch, so a shard equals the same rows of the unsharded batch bit for bit)."""
    cfg = _cfg.ergocub_gazebo_v1(N, 0.06)
    rng = np.random.default_rng(seed)
    com0, dcom0, h0 = _perturbed_state(rng, B, (0.0, 0.0, 0.7))
    push = rng.uniform(-50.0, 50.0, (B, 2)) / ROBOT_MASS
    if shard is not None:
        lo, hi = shard
        com0, dcom0, h0, push = com0[lo:hi], dcom0[lo:hi], h0[lo:hi], push[lo:hi]
        B = hi - lo
    sched = _tile(sample_schedule(cfg, _walking_lists(cfg, 6, 8)), B)
    ref = np.broadcast_to(np.array([0.0, 0.0, 0.7]), (B, N + 1, 3)).copy()
    f_ext = np.zeros((B, N, 3))
    nk = int(np.ceil(0.2 / cfg.sampling_time))
    f_ext[:, :nk, :2] = push[:, None, :]
    return _finish(cfg, sched, com0, dcom0, h0, ref, np.zeros((B, N + 1, 3)), f_ext)


def config4_monte_carlo(B=65536, N=20, seed=2, shard=None):
    """Config-3 generator with seed 2; `shard=(lo, hi)` builds one rank's contiguous shard (8 x 8192 with
    distributed.shard_bounds at 8 GPUs)."""
    return config3_external_push(B, N, seed, shard)


def config5_footstep_candidates(B=8192, N=30, seed=3):
    """ergoCubGazeboV1, horizon 30.  The reference's MANN generator cannot run here (no onnxruntime,
    model blob missing), so candidate schedules are synthetic: step length U(0,0.15) m, width
    0.16+-0.02 m, yaw U(-0.2,0.2) rad, step duration 0.6-0.9 s, double support 0.12-0.24 s."""
    cfg = _cfg.ergocub_gazebo_v1(N, 0.06)
    rng = np.random.default_rng(seed)
    dt, t_end = cfg.sampling_time, N * cfg.sampling_time
    lists = []
    for _ in range(B):  # the random draws only (their order defines the batch); the sampling below is vectorised
        step_T = rng.uniform(0.6, 0.9)
        ds = rng.uniform(0.12, 0.24)
        swing_left = bool(rng.integers(2))
        feet = {True: [PlannedContact(-1.0, 0.0, (0.0, FOOT_Y, 0.0))],
                False: [PlannedContact(-1.0, 0.0, (0.0, -FOOT_Y, 0.0))]}
        t = ds
        while t < t_end + step_T:
            stance = feet[not swing_left][-1]
            land_t = t + (step_T - ds)
            width = rng.uniform(0.14, 0.18)
            yaw = rng.uniform(-0.2, 0.2)
            x = stance.position[0] + rng.uniform(0.0, 0.15)
            y = stance.position[1] + (width if swing_left else -width)
            feet[swing_left][-1].deactivation_time = t
            feet[swing_left].append(PlannedContact(land_t, land_t, (x, y, 0.0), yaw))
            t = land_t + ds
            swing_left = not swing_left
        for lst in feet.values():
            lst[-1].deactivation_time = t_end + 10.0
            for a, b in zip(lst[:-1], lst[1:]):
                if a.deactivation_time <= a.activation_time:
                    a.deactivation_time = b.activation_time - (step_T - ds)
        lists.append({cfg.contacts[0].contact_name: feet[True], cfg.contacts[1].contact_name: feet[False]})
    sched, _ = sample_schedule_batch(cfg, *pack_lists(cfg, lists))
    sched = {k: v.astype(np.float64) for k, v in sched.items()}
    com0, dcom0, h0 = _perturbed_state(rng, B, (0.0, 0.0, 0.7))
    # CoM reference: constant-velocity drift towards the mean of the last nominal foot positions
    goal = 0.5 * (sched["nominal"][:, 0, -1] + sched["nominal"][:, 1, -1])
    s = np.linspace(0.0, 1.0, N + 1)[None, :, None]
    ref = np.zeros((B, N + 1, 3))
    ref[:, :, :2] = s * goal[:, None, :2]
    ref[:, :, 2] = 0.7
    return _finish(cfg, sched, com0, dcom0, h0, ref, np.zeros((B, N + 1, 3)))


def standing_known_answer(N=12, dt=0.1, which="tmp"):
    """The standing problem of SURVEY 8c (iii): tmp.c weight set, N=12, dt=0.1 -> f* = 8.16487469."""
    cfg = _cfg.generated_code_weights(which, N, dt)
    sched = _tile(sample_schedule(cfg, _standing_lists(cfg, N * dt)), 1)
    com0 = np.array([[0.01, -0.005, 0.69]])
    dcom0 = np.array([[0.02, 0.0, 0.0]])
    ref = np.broadcast_to(np.array([0.0, 0.0, 0.7]), (1, N + 1, 3)).copy()
    return _finish(cfg, sched, com0, dcom0, np.zeros((1, 3)), ref, np.zeros((1, N + 1, 3)))


def walking_push_n12(which="tmp", B=16, seed=41):
    """N = 12, dt = 0.1 with the weights baked into the reference's generated code (`which`: tmp.c or
    jit_tmpComMiH.c): a swing phase inside the horizon (left Gamma = 1x3, 0x5, 1x4, next footstep +0.1 m) and an
    external push over the first two knots -- the step adjustment is active.  Small on purpose: these problems are
    solved on the reference's own compiled functions (oracle/_ref) to make golden vectors."""
    N, dt = 12, 0.1
    cfg = _cfg.generated_code_weights(which, N, dt)
    rng = np.random.default_rng(seed)
    sched = _tile(sample_schedule(cfg, _walking_lists(cfg, 3, 5)), B)
    com0, dcom0, h0 = _perturbed_state(rng, B, (0.0, 0.0, 0.7))
    ref = np.broadcast_to(np.array([0.0, 0.0, 0.7]), (B, N + 1, 3)).copy()
    f_ext = np.zeros((B, N, 3))
    f_ext[:, :2, :2] = (rng.uniform(-40.0, 40.0, (B, 2)) / ROBOT_MASS)[:, None, :]
    return _finish(cfg, sched, com0, dcom0, h0, ref, np.zeros((B, N + 1, 3)), f_ext)


def yawed_steps_n12(which="tmp", B=16, seed=42):
    """N = 12, dt = 0.1, generated-code weights: two yawed footsteps (R != I in the friction and bounding-box rows)."""
    N, dt = 12, 0.1
    cfg = _cfg.generated_code_weights(which, N, dt)
    rng = np.random.default_rng(seed)
    lists = []
    for _ in range(B):
        y1, y2 = rng.uniform(-0.25, 0.25, 2)
        left = [PlannedContact(-1.0, 0.2, (0.0, FOOT_Y, 0.0), rng.uniform(-0.1, 0.1)), PlannedContact(0.6, 10.0, (0.12, FOOT_Y + 0.02, 0.0), y1)]
        right = [PlannedContact(-1.0, 0.8, (0.0, -FOOT_Y, 0.0), rng.uniform(-0.1, 0.1)), PlannedContact(1.1, 10.0, (0.2, -FOOT_Y - 0.01, 0.0), y2)]
        lists.append({cfg.contacts[0].contact_name: left, cfg.contacts[1].contact_name: right})
    sched, _ = sample_schedule_batch(cfg, *pack_lists(cfg, lists))
    sched = {k: v.astype(np.float64) for k, v in sched.items()}
    com0, dcom0, h0 = _perturbed_state(rng, B, (0.0, 0.0, 0.7))
    s = np.linspace(0.0, 1.0, N + 1)[None, :, None]
    ref = np.zeros((B, N + 1, 3))
    ref[:, :, 0] = 0.1 * s[:, :, 0]
    ref[:, :, 2] = 0.7
    return _finish(cfg, sched, com0, dcom0, h0, ref, np.zeros((B, N + 1, 3)))


def push_recovery_n12(which="tmp", B=16, seed=43):
    """N = 12, dt = 0.1, generated-code weights: both feet on the ground and a horizontal push of 120-220 N (2.1-3.9 N/kg against a
    friction limit of mu g = 3.2 N/kg) over the first three knots -- friction rows are ACTIVE at the optimum (push recovery on the
    edge of the cone)."""
    N, dt = 12, 0.1
    cfg = _cfg.generated_code_weights(which, N, dt)
    rng = np.random.default_rng(seed)
    sched = _tile(sample_schedule(cfg, _standing_lists(cfg, N * dt)), B)
    com0, dcom0, h0 = _perturbed_state(rng, B, (0.0, 0.0, 0.7))
    ref = np.broadcast_to(np.array([0.0, 0.0, 0.7]), (B, N + 1, 3)).copy()
    ang, mag = rng.uniform(0.0, 2.0 * np.pi, B), rng.uniform(120.0, 220.0, B) / ROBOT_MASS
    f_ext = np.zeros((B, N, 3))
    f_ext[:, :3, 0] = (mag * np.cos(ang))[:, None]
    f_ext[:, :3, 1] = (mag * np.sin(ang))[:, None]
    return _finish(cfg, sched, com0, dcom0, h0, ref, np.zeros((B, N + 1, 3)), f_ext)


def single_support_end_n12(which="tmp", B=16, seed=44):
    """N = 12, dt = 0.1, generated-code weights: the left foot lifts at knot 7-9 and is still in the air at the end of the horizon
    (single support at the horizon end: the landing row constrains the last knot only, landing knot = N); small pushes."""
    N, dt = 12, 0.1
    cfg = _cfg.generated_code_weights(which, N, dt)
    rng = np.random.default_rng(seed)
    scheds = []
    for _ in range(B):
        lift = int(rng.integers(7, 10))
        scheds.append(sample_schedule(cfg, _walking_lists(cfg, lift, N, step=float(rng.uniform(0.05, 0.12)))))
    sched = {k: np.stack([sc[k] for sc in scheds]) for k in scheds[0]}
    com0, dcom0, h0 = _perturbed_state(rng, B, (0.0, -0.02, 0.7))
    ref = np.broadcast_to(np.array([0.0, -0.04, 0.7]), (B, N + 1, 3)).copy()      # weight shifting onto the stance foot
    f_ext = np.zeros((B, N, 3))
    f_ext[:, :2, :2] = (rng.uniform(-25.0, 25.0, (B, 2)) / ROBOT_MASS)[:, None, :]
    return _finish(cfg, sched, com0, dcom0, h0, ref, np.zeros((B, N + 1, 3)), f_ext)


def standing_n12(which="tmp", B=16, seed=45):
    """N = 12, dt = 0.1, generated-code weights: a batch of the standing problem of SURVEY 8c (iii) with perturbed initial states
    (config 2 at the horizon and weights of the reference's generated code)."""
    N, dt = 12, 0.1
    cfg = _cfg.generated_code_weights(which, N, dt)
    rng = np.random.default_rng(seed)
    sched = _tile(sample_schedule(cfg, _standing_lists(cfg, N * dt)), B)
    com0, dcom0, h0 = _perturbed_state(rng, B, (0.0, 0.0, 0.7))
    ref = np.broadcast_to(np.array([0.0, 0.0, 0.7]), (B, N + 1, 3)).copy()
    return _finish(cfg, sched, com0, dcom0, h0, ref, np.zeros((B, N + 1, 3)))
