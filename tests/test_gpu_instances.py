"""Every step_kernel instantiation the library holds (175: quadrotor_kernels.hip QR_INSTANCES x MAG) is reachable through the public API,
runs, and agrees with the float64 layout's instantiation of the same workload — and the process' launch counters (qr_launch_stats)
show that all of them were really launched ("nothing records which instantiations the GPU tests launch").
Also here: qr_touch, the do-nothing kernel bench.py prices a step against."""
import itertools

import numpy as np
import pytest
import torch

from conftest import grouped_rel_err

KINDS = ("quad", "coupled", "decoupled")
BIG = 64 * 1700 + 5      # 1701 tiles: beyond QR_HELP_REWARD_TILES / QR_HELP_ROWS_TILES, inside the helper-wave thresholds


def _make(kind, n, layout, goal_mode, auto_reset, w_adapt, helper=None, substeps=1):
    from gym_rotor_amd import QuadVecEnv
    return QuadVecEnv(kind, n, device="cuda", seed=4, layout=layout, goal_mode=goal_mode, auto_reset=auto_reset, w_adapt=w_adapt,
                      helper=helper, helper_rollout=helper, autotune=False, substeps=substeps)


def _recipes():
    """One way of reaching every instantiation: walk a grid of env configurations and workloads, ask the launcher's own decision
    function (launch_plan: host-side) which kernel each would run, keep the first recipe per kernel."""
    found = {}
    for layout, kind, goal_mode, auto_reset, w_adapt, n, sub in itertools.product(("mixed", "f64", "f32"), KINDS, (None, 0, 2), (True, False), (16.0, 8.0),
                                                                                  (640, BIG), (1, 2)):
        if sub == 2 and layout != "mixed":      # (the Magnus twins exist in the default layout only)
            continue
        env = _make(kind, n, layout, goal_mode, auto_reset, w_adapt, substeps=sub)
        for helper, (T, actor) in itertools.product((None, False), ((1, None), (3, None), (3, "ppo"), (3, "sac"))):
            if actor and kind == "quad":
                continue
            env.set_launch(helper, helper)
            key = env.launch_plan(T, actor)["key"]
            found.setdefault(key, dict(layout=layout, kind=kind, goal_mode=goal_mode, auto_reset=auto_reset, w_adapt=w_adapt, n=n, helper=helper, T=T, actor=actor,
                                       substeps=sub))
        del env
    return found


def _run(r, layout, helper):
    """The recipe's workload on `layout`: three env-steps from a seeded reset; returns (state, any-done over the steps)."""
    from gym_rotor_amd import random_actors
    env = _make(r["kind"], r["n"], layout, r["goal_mode"], r["auto_reset"], r["w_adapt"], helper, r["substeps"])
    env.reset("train")
    if r["kind"] != "quad":
        if r["goal_mode"] is not None:
            env.get_desired(store_goal=True)
        env.get_norm_error_state()
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    acts = torch.rand(3, r["n"], env.action_dim, device="cuda", generator=g) * 2 - 1
    if r["actor"]:
        actors = random_actors(r["kind"], "cuda", generator=torch.Generator(device="cuda").manual_seed(5), log_std=-1.0, algo=r["actor"])
        out = env.rollout_actor(actors, 3)
        done = out["terminated"].any(dim=0).any(dim=-1)
    elif r["T"] > 1:
        done = env.rollout(acts)["terminated"].any(dim=0).any(dim=-1)
    else:
        done = torch.zeros(r["n"], dtype=torch.bool, device="cuda")
        for t in range(3):
            done |= env.step(acts[t])[2].any(dim=-1)
    plan = env.launch_plan(r["T"], r["actor"])
    return env.get_current_state().cpu().numpy(), done.cpu().numpy(), plan


@pytest.mark.gpu
def test_every_instantiation_is_reachable_runs_and_agrees_with_the_float64_layout():
    from gym_rotor_amd import _lib
    table = _lib.instance_table()
    assert len(table) == len(set(table)) == 175
    recipes = _recipes()
    missing = [_lib.describe_key(k) for k in table if k not in recipes]
    assert not missing, f"no recipe reaches: {missing}"
    assert set(recipes) == set(table)                    # and the decision function never names a kernel outside the table
    _lib.launch_stats(reset=True)
    refs, worst = {}, {}
    for key in table:
        r = recipes[key]
        state, done, plan = _run(r, r["layout"], r["helper"])
        assert plan["key"] == key and np.isfinite(state).all(), (_lib.describe_key(key), r)
        rk = (r["kind"], r["n"], r["goal_mode"], r["auto_reset"], r["w_adapt"], r["T"], r["actor"], r["substeps"])
        if rk not in refs:
            refs[rk] = _run(r, "f64", None)[:2]
        ref_state, ref_done = refs[rk]
        keep = ~done & ~ref_done                         # (an env that ended was re-sampled; a termination decided within rounding may differ)
        assert keep.mean() > 0.8, (_lib.describe_key(key), keep.mean())
        err = grouped_rel_err(state[keep], ref_state[keep])
        # actor rollouts: the float32 observation feeds the actor, whose action feeds the dynamics — rounding is amplified by the policy
        tol = {"mixed": 2e-5 if r["actor"] else 1e-5, "f64": 1e-9, "f32": 2e-2 if r["actor"] else 5e-3}[r["layout"]]
        assert err <= tol, (_lib.describe_key(key), r, err)
        worst[r["layout"]] = max(worst.get(r["layout"], 0.0), err)
    launched = _lib.launch_stats()
    never = [_lib.describe_key(k) for k in table if launched.get(k, 0) == 0]
    assert not never, f"instantiations the launch counters never saw: {never}"
    print("worst grouped error vs the float64 layout per layout:", worst)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,layout", [("quad", "mixed"), ("coupled", "mixed"), ("decoupled", "mixed"), ("quad", "f64"), ("decoupled", "f32")])
def test_touch_moves_the_steps_bytes_and_changes_no_state(kind, layout):
    """qr_touch (bench.py's yardstick): reads and writes what a step does, leaves state / integrators / parameters bit-identical and
    zeroes the output rows; ragged tail included."""
    from gym_rotor_amd import QuadVecEnv
    n = 64 * 9 + 17
    env = QuadVecEnv(kind, n, device="cuda", seed=2, layout=layout, auto_reset=True, obs_rows=True)
    env.reset("train")
    if kind != "quad":
        env.get_norm_error_state()
    act = torch.rand(n, env.action_dim, device="cuda") * 2 - 1
    env.step(act)
    before = [t.clone() for t in (env._pos_vel, env._att_rate, env._params, env._integ, env._episode, env._reset_count) if t is not None]
    env.touch(act)
    torch.cuda.synchronize()
    after = [t for t in (env._pos_vel, env._att_rate, env._params, env._integ, env._episode, env._reset_count) if t is not None]
    for b, a in zip(before, after):
        assert torch.equal(a, b)
    assert float(env._reward.abs().max()) == 0.0 and not bool(env._done.any()) and float(env._obs0.abs().max()) == 0.0
    if env._obs1 is not None:
        assert float(env._obs1.abs().max()) == 0.0
    with pytest.raises(ValueError):
        env.touch(act[:-1])
