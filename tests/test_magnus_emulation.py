"""CPU: the algorithm behind the default layout's Magnus substep (gym_rotor_amd/csrc/qr_dynamics.h: integrate_magnus), as restated in
NumPy by tools/numerics_magnus.py, against the REFERENCE's one-step vectors (tests/golden/onestep_quad.npz: DOP853) — 4th-order
convergence in the substep count and the sign conventions of the Magnus series' commutator term.  The kernel itself is pinned on the GPU
(tests/test_gpu_parity.py: test_onestep_golden_magnus_substeps, the production-mode tests); this test keeps the derivation honest."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import quad_oracle as orc  # noqa: E402


def _one_step(nsub, dtype, d):
    from tools import numerics_magnus as nm
    st = d["state"].astype(np.float64)
    dv = orc.derive(d["params"].astype(np.float64))
    f, M = orc.action_map_batch("quad", d["action"].astype(np.float64), st, dv)
    q = nm.R_to_quat(st[:, 6:15])
    c = f / dv.m
    A1 = (dv.J1 - dv.J3) / dv.J1
    U = np.stack([M[:, 0] / dv.J1, M[:, 1] / dv.J1, M[:, 2] / dv.J3], 1)
    x, v, qn, W = nm.magnus_step_em(st[:, 0:3].copy(), st[:, 3:6].copy(), q, st[:, 15:18].copy(), orc.DT, nsub, c, A1, U, dtype, deg=5)
    qn = qn / np.linalg.norm(qn, axis=1, keepdims=True)
    return np.concatenate([x, v, nm.quat_to_R(qn), W], 1)


def test_magnus_substep_converges_at_fourth_order_to_the_reference_step():
    from conftest import grouped_rel_err
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", "onestep_quad.npz")))
    ref = d["next_state"]
    err = {n: grouped_rel_err(_one_step(n, np.float64, d), ref) for n in (1, 2, 4)}
    print("Magnus, float64, one env-step against the reference (DOP853):", {n: f"{e:.2e}" for n, e in err.items()})
    assert err[1] <= 2e-8 and err[2] <= 2e-9 and err[4] <= 2e-10          # the f64 layout's RK4 bars (test_gpu_parity.ONESTEP_TOL)
    assert 8.0 <= err[1] / err[2] <= 32.0                                  # 4th order: x 16 per halving of h
    # the float32 arithmetic of the kernel (x, v float32): the default layout's one-step bar
    assert grouped_rel_err(_one_step(2, np.float32, d), ref) <= 2e-7


def test_commutator_sign_matters():
    """With the Magnus series' second term reversed the scheme drops to second order: the sign in the kernel is the measured one."""
    from conftest import grouped_rel_err
    from tools import numerics_magnus as nm
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", "onestep_quad.npz")))
    good = grouped_rel_err(_one_step(2, np.float64, d), d["next_state"])
    nm.SIGMA = -1.0
    try:
        bad = grouped_rel_err(_one_step(2, np.float64, d), d["next_state"])
    finally:
        nm.SIGMA = 1.0
    assert bad > 20 * good
