"""TEST INFRASTRUCTURE ONLY -- maps a package config + parameter vector onto the reference NLP's
(cfg struct, lbg, ubg): what CasADi's Opti hands IPOPT as constraint bounds (SURVEY 8a-NLP:
`param == expr` / `lo <= expr <= hi` become parametric bounds outside the generated code)."""
import numpy as np

from . import oracle_lib as ol

INF = 1e20


def oracle_cfg(cfg):
    return ol.make_cfg(cfg.N, cfg.sampling_time, mu=cfg.static_friction_coefficient,
                       w_com=cfg.com_weight, w_h=cfg.angular_momentum_weight,
                       w_pos=cfg.contact_position_weight, w_rate=cfg.force_rate_of_change_weight,
                       w_sym=cfg.contact_force_symmetry_weight,
                       corners=[c.corners for c in cfg.contacts])


def bounds(cfg, p):
    """lbg, ubg for one parameter vector p (reference g-row order, SURVEY 8a-NLP 'Constraints')."""
    N = cfg.N
    ng = 53 * N + 15
    lb = np.zeros(ng)
    ub = np.zeros(ng)
    per = 19 * N + 6
    tail = 2 * per
    cur = [p[c * per + 19 * N + 3:c * per + 19 * N + 6] for c in range(2)]
    init = np.concatenate([p[tail:tail + 9], cur[0], cur[1]])
    lb[:15] = init
    ub[:15] = init
    o = 15 + 15 * N  # after com/dcom/h dynamics (9N) and both foot-position dynamics (6N)
    for c in range(2):
        up = p[c * per + 9 * N:c * per + 12 * N]
        lo = p[c * per + 12 * N:c * per + 15 * N]
        lb[o:o + 3 * N] = lo
        ub[o:o + 3 * N] = up
        o += 3 * N
        lb[o:o + 16 * N] = -INF
        ub[o:o + 16 * N] = 0.0
        o += 16 * N
    assert o == ng
    return lb, ub
