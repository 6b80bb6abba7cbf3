"""CPU: host-side logic, the C-ABI surface (loads, exports, struct layout, argument errors)
and the multi-process sharding path over gloo.  No compute kernel is launched here."""
import ctypes as C
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import quad_oracle as orc


def _lib():
    from gym_rotor_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L


def test_constants_match_reference(golden):
    from gym_rotor_amd import QuadConstants
    c, k = QuadConstants(), golden("kat_units")["constants"]
    assert (c.reward_min, c.reward_min_1, c.reward_min_2) == tuple(k[0:3])
    assert (c.dt, c.x_lim, c.v_lim, c.W_lim, c.euler_lim) == tuple(k[3:8])
    d = golden("kat_units")["act_quad_derived"][0]  # nominal params row
    assert np.allclose([c.hover_force, c.max_force, c.avrg_act, c.scale_act], d, rtol=1e-7)  # params row is f32-rounded
    assert np.array_equal(c.forces_to_fM, orc.forces_to_fM(c.d_nominal, c.c_tf_nominal))
    assert np.array_equal(c.nominal_params, orc.NOMINAL_PARAMS)


def test_box_space():
    from gym_rotor_amd import Box
    b = Box(-1.0, 1.0, shape=(5,), dtype=np.float32)
    b.seed(3); x = b.sample(); b.seed(3)
    assert x.dtype == np.float32 and x.shape == (5,) and np.array_equal(x, b.sample()) and b.contains(x)
    assert not b.contains(np.full(5, 2.0, np.float32))


def test_shard_range_properties():
    from hypothesis import given, settings, strategies as st
    from gym_rotor_amd import shard_range

    @settings(max_examples=200, deadline=None)
    @given(st.integers(0, 10 ** 7), st.integers(1, 64))
    def prop(n, w):
        rs = [shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        sizes = [e - s for s, e in rs]
        if n >= 64 * w:   # tile-aligned shards: every boundary a multiple of 64 envs (one wavefront), tiles dealt evenly
            assert all(s % 64 == 0 for s, _ in rs) and max(sizes) - min(sizes) <= 64 + 63
            assert sizes[:-1] == sorted(sizes[:-1], reverse=True)
        else:
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)

    prop()
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)
    assert shard_range(262144, 3, 8) == (98304, 131072)  # BASELINE.json configs[3]: 32 768 envs per GPU


def test_library_exports_every_declared_symbol():
    L = _lib()
    hdr = open(os.path.join(ROOT, "include", "quadrotor_hip.h")).read()
    declared = set(re.findall(r"\b(qr_[a-z_]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    lib = L.load()
    for s in declared:
        assert hasattr(lib, s)
    assert lib.qr_abi_version() == L.ABI_VERSION
    assert int(re.search(r"#define QR_ABI_VERSION (\d+)", hdr).group(1)) == L.ABI_VERSION


def test_ctypes_structs_mirror_the_header(tmp_path):
    """Compile a C program against include/quadrotor_hip.h and compare sizeof/offsetof."""
    L = _lib()
    fields = {n: [f[0] for f in getattr(L, n)._fields_] for n in ("QrEnv", "QrStepOut", "QrCoeffs", "QrActor", "QrPolicyRollout")}
    lines = []
    for sname, fl in fields.items():
        lines.append(f'printf("{sname} %zu\\n", sizeof({sname}));')
        lines += [f'printf("{sname}.{f} %zu\\n", offsetof({sname}, {f}));' for f in fl]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "quadrotor_hip.h"\nint main(void){' + "".join(lines) + "return 0;}")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT}/include", str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for sname, fl in fields.items():
        ct = getattr(L, sname)
        assert int(out[sname]) == C.sizeof(ct)
        for f in fl:
            assert int(out[f"{sname}.{f}"]) == getattr(ct, f).offset, (sname, f)


def test_plain_c_host_builds_against_the_header(tmp_path):
    """tests/cabi/host_demo.c — a C99 host of the C-ABI (no Python, no HIP compiler: gcc + the HIP runtime's C API) — compiles with
    -Wall -Werror against include/quadrotor_hip.h and links against the library.  (It RUNS, and matches the Python host bit for
    bit, in the GPU suite: test_plain_c_host_of_the_c_abi_matches_the_python_host.)"""
    _lib()
    exe = tmp_path / "host_demo"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT}/include",
           f"{ROOT}/tests/cabi/host_demo.c", f"-L{ROOT}/gym_rotor_amd", "-lquadrotor_hip", "-L/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{ROOT}/gym_rotor_amd", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0 and exe.exists(), r.stderr[-1500:]


def test_abi_argument_errors_without_gpu():
    L = _lib()
    lib = L.load()
    e, o = L.QrEnv(), L.QrStepOut()
    assert lib.qr_step(None, None, 1, None, None) == -1
    z = L.QrEnv()
    z.num_envs, z.pos_vel, z.att_rate = 4, 0x1000, 0x2000
    assert lib.qr_reset(C.byref(z), None, None) == -3                         # zero QrCoeffs: qr_default_coeffs not called
    lib.qr_default_coeffs(C.byref(e.coeffs))
    e.kind = 7
    assert lib.qr_step(C.byref(e), None, 1, C.byref(o), None) == -2
    e.kind, e.layout = 0, 5
    assert lib.qr_step(C.byref(e), None, 1, C.byref(o), None) == -2
    e.layout, e.num_envs = 0, -3
    assert lib.qr_step(C.byref(e), None, 1, C.byref(o), None) == -3
    e.num_envs = 4
    assert lib.qr_step(C.byref(e), None, 1, C.byref(o), None) == -1          # NULL state buffers
    e.pos_vel, e.att_rate = 0x1008, 0x2000
    assert lib.qr_reset(C.byref(e), None, None) == -4                         # misaligned
    e.pos_vel = 0x1000
    assert lib.qr_reset(C.byref(e), None, None) == -1                         # no episode counters
    assert lib.qr_step(C.byref(e), None, 1, C.byref(o), None) == -1          # NULL action
    assert lib.qr_step(C.byref(e), 0x3000, 0, C.byref(o), None) == -1        # outputs missing
    o.reward, o.done = 0x4000, 0x5000
    assert lib.qr_step(C.byref(e), 0x3000, 0, C.byref(o), None) == -3        # substeps < 1
    assert lib.qr_rollout(C.byref(e), 0x3000, 0, 1, C.byref(o), None) == -3  # n_steps < 1
    e.kind = 1
    assert lib.qr_step(C.byref(e), 0x3000, 1, C.byref(o), None) == -1        # wrapper without integ/obs
    assert lib.qr_error_obs(C.byref(e), None, None, None) == -1
    e.kind = 0
    assert lib.qr_error_obs(C.byref(e), 0x6000, None, None) == -2            # undefined for Quad-v0
    assert lib.qr_error_obs_format(C.byref(e), 1, 0x6000, None, None) == -2  # ... in either format (a bare QuadEnv raises in the reference too)
    e.kind = 1
    assert lib.qr_error_obs_format(C.byref(e), 0, 0x6000, None, None) == -2 and lib.qr_error_obs_format(C.byref(e), 3, 0x6000, None, None) == -2
    assert lib.qr_error_obs_format(C.byref(e), 2, 0x6000, None, None) == -1  # no integrator buffer; and MODUL needs obs1
    e.kind = 0
    assert lib.qr_get_state(C.byref(e), None, None) == -1 and lib.qr_set_state(C.byref(e), None, None, None, None) == -1
    # empty batch: a no-op that succeeds; oversize batch (32-bit buffer offsets): refused
    e2, o2 = L.QrEnv(), L.QrStepOut()
    lib.qr_default_coeffs(C.byref(e2.coeffs))
    e2.kind, e2.num_envs, e2.pos_vel, e2.att_rate = 0, 0, 0x1000, 0x2000
    o2.reward, o2.done = 0x4000, 0x5000
    assert lib.qr_step(C.byref(e2), 0x3000, 1, C.byref(o2), None) == 0
    e2.num_envs = 1 << 26
    assert lib.qr_step(C.byref(e2), 0x3000, 1, C.byref(o2), None) == -3
    e2.num_envs, e2.field_stride = 100, 50
    assert lib.qr_step(C.byref(e2), 0x3000, 1, C.byref(o2), None) == -3      # stride < N
    e2.field_stride, e2.goal_mode = 0, 2
    assert lib.qr_step(C.byref(e2), 0x3000, 1, C.byref(o2), None) == -1      # fused goals without traj buffer
    e2.goal_mode, e2.traj = 4, 0x7000
    assert lib.qr_get_desired(C.byref(e2), None, 0x8000, 0, None) == -1      # a stateful mode (take-off) without the goal buffer it persists in
    e2.goal_mode = 8
    assert lib.qr_get_desired(C.byref(e2), None, 0x8000, 0, None) == -2      # no such mode
    e2.goal_mode, e2.traj = 2, None
    # qr_rollout_actor: argument checks before any launch
    e3, o3, pol = L.QrEnv(), L.QrStepOut(), L.QrPolicyRollout()
    lib.qr_default_coeffs(C.byref(e3.coeffs))
    e3.kind, e3.num_envs, e3.pos_vel, e3.att_rate, e3.integ = 0, 64, 0x1000, 0x2000, 0x7000
    o3.reward, o3.done, o3.obs0, o3.obs1 = 0x4000, 0x5000, 0x8000, 0x9000
    assert lib.qr_rollout_actor(C.byref(e3), None, 1, 1, C.byref(o3), None) == -1
    assert lib.qr_rollout_actor(C.byref(e3), C.byref(pol), 1, 1, C.byref(o3), None) == -2     # Quad-v0 has no actor
    e3.kind = 1
    assert lib.qr_rollout_actor(C.byref(e3), C.byref(pol), 1, 1, C.byref(o3), None) == -1     # no actors / obs / outputs
    acts = (L.QrActor * 2)()
    pol.actors, pol.obs0_in, pol.obs1_in, pol.action_out, pol.max_action = acts, 0xa000, 0xb000, 0xc000, 1.0
    assert lib.qr_rollout_actor(C.byref(e3), C.byref(pol), 1, 1, C.byref(o3), None) == -3     # sizes are not 23 -> 16 -> 4
    acts[0].obs_dim, acts[0].hidden_dim, acts[0].action_dim = 23, 16, 4
    assert lib.qr_rollout_actor(C.byref(e3), C.byref(pol), 1, 1, C.byref(o3), None) == -1     # weight pointers missing
    for n in ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "mean_w", "mean_b", "log_std"):
        setattr(acts[0], n, 0xd000); setattr(acts[1], n, 0xd000)
    pol.action_out = 0xc004
    assert lib.qr_rollout_actor(C.byref(e3), C.byref(pol), 1, 1, C.byref(o3), None) == -4     # action rows misaligned
    pol.action_out, pol.max_action = 0xc000, 0.0
    assert lib.qr_rollout_actor(C.byref(e3), C.byref(pol), 1, 1, C.byref(o3), None) == -3     # max_action <= 0
    pol.max_action = 1.0
    assert lib.qr_rollout_actor(C.byref(e3), C.byref(pol), 0, 1, C.byref(o3), None) == -3     # n_steps < 1
    e3.kind = 2
    assert lib.qr_rollout_actor(C.byref(e3), C.byref(pol), 1, 1, C.byref(o3), None) == -3     # decoupled wants 15->16->4, 3->4->1
    with pytest.raises(ValueError):
        L.check(-3, "x")
    with pytest.raises(L.QuadrotorLibError):
        L.check(700, "x")
    co = L.default_coeffs()
    assert (co.Cx, co.CIx, co.Cv, co.Cb1, co.CIb1, co.CW, co.Cw12, co.CW3) == (6.0, 0.1, 0.4, 6.0, 0.1, 0.6, 0.6, 0.1)
    assert (co.alpha, co.beta, co.dt, co.W_lim) == (0.01, 0.05, 1 / 200, 2 * np.pi)
    assert (co.m_nominal, co.d_nominal, co.J1_nominal, co.J3_nominal, co.c_tf_nominal, co.c_tw_nominal, co.g, co.min_force) == \
           (2.15, 0.23, 0.022, 0.035, 0.0135, 2.2, 9.81, 0.5)                # quad.py:28-36
    # in-launch resets need the per-tile stream counters
    e4, o4 = L.QrEnv(), L.QrStepOut()
    lib.qr_default_coeffs(C.byref(e4.coeffs))
    e4.kind, e4.num_envs, e4.pos_vel, e4.att_rate, e4.episode, e4.flags = 0, 64, 0x1000, 0x2000, 0x3000, L.FLAG_AUTO_RESET
    o4.reward, o4.done = 0x4000, 0x5000
    assert lib.qr_step(C.byref(e4), 0x6000, 1, C.byref(o4), None) == -1      # reset_count missing


def test_abi_argument_errors_under_sanitizers(tmp_path):
    """The host-side launcher code (fill_env, fill_coeffs, do_rollout, the launch-geometry rule: everything a C-ABI call
    executes before it reaches a kernel) built with -fsanitize=address,undefined (`make host-sanitize`: the host pass of the
    single-source file only) and driven through every argument-error path by the two tests above, in a subprocess with the
    sanitizer runtime preloaded.  CPU box only; a report aborts the child (-fno-sanitize-recover, ASan's default)."""
    import glob
    import shutil
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("no ASan runtime in this ROCm install")
    out = tmp_path / "libquadrotor_hip_hostsan.so"
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "gym_rotor_amd", "csrc"), "host-sanitize", f"SAN_OUT={out}"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and out.exists(), r.stderr[-2000:]
    env = dict(os.environ, QR_LIB=str(out), LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", PYTHONPATH=ROOT,
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_host_logic.py"),
                        "-k", "test_abi_argument_errors_without_gpu or test_launch_geometry_rule_without_gpu or test_library_exports or test_launch_plan_names"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "4 passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_launch_geometry_rule_without_gpu():
    """qr_step_kernel_info (host-only) reports the launch the step launcher would use: one 64-lane wavefront per 64-env
    tile, plus a helper wavefront (128 threads per workgroup) exactly for: in-launch auto-reset, default layout, no rate
    adaptivity in reach, and grids of <= 3328 tiles (Quad-v0, one substep; 2560 with more) / <= 2560 (wrappers; 1664 with more substeps) / <= 1024 (rollouts); with the fused
    goal generator only for one-step launches."""
    L = _lib()
    lib = L.load()

    def info(kind, n, flags, layout=0, n_steps=1, goal_mode=None, w_adapt=None):
        e = L.QrEnv()
        lib.qr_default_coeffs(C.byref(e.coeffs))
        e.kind, e.layout, e.num_envs, e.pos_vel, e.att_rate, e.flags = kind, layout, n, 0x1000, 0x2000, flags
        if goal_mode is not None:
            e.goal_mode, e.traj = goal_mode, 0x7000
        if w_adapt is not None:
            e.coeffs.w_adapt = w_adapt
        g, b = C.c_int32(), C.c_int32()
        name = lib.qr_step_kernel_info(C.byref(e), n_steps, C.byref(g), C.byref(b))
        return name.decode(), g.value, b.value

    hdr = open(os.path.join(ROOT, "include", "quadrotor_hip.h")).read()
    AR = int(re.search(r"#define QR_FLAG_AUTO_RESET\s+(\S+)", hdr).group(1).rstrip("u"), 0)
    GOAL_EXTERNAL = int(re.search(r"#define QR_GOAL_EXTERNAL\s+(\d+)", hdr).group(1))
    assert info(0, 65536, AR) == ("qr::step_kernel<0,...>", 1024, 128)
    assert info(0, 65536 + 1, AR)[1:] == (1025, 128)                      # ragged tail: one more tile
    assert info(0, 212992, AR)[2] == 128 and info(0, 212992 + 64, AR)[2] == 64   # 3328 tiles (one substep)
    assert info(1, 163840, AR)[2] == 128 and info(1, 163840 + 64, AR)[2] == 64   # 2560 tiles (one substep)
    assert info(2, 32768, AR)[2] == 128 and info(2, 131072, AR)[2] == 128 and info(2, 262144, AR)[2] == 64
    assert info(0, 65536, 0)[2] == 64                                      # no in-launch reset: nothing for a helper to sample
    assert info(0, 65536, AR, layout=1)[2] == 64 and info(0, 65536, AR, layout=2)[2] == 64
    assert info(0, 65536, AR, w_adapt=3.0)[2] == 64                        # rate adaptivity within reach of |W| < W_lim: the adaptive kernel
    assert info(0, 65536, AR, n_steps=100)[2] == 128 and info(0, 65536 + 64, AR, n_steps=100)[2] == 64   # rollouts: up to 1024 tiles
    gm = int(re.search(r"#define QR_GOAL_MODE0\s+(\d+)", hdr).group(1))
    assert GOAL_EXTERNAL == 0 and info(1, 65536, AR, goal_mode=gm)[2] == 128 and info(1, 65536, AR, goal_mode=gm, n_steps=8)[2] == 64
    assert info(0, -5, AR)[0] == ""                                        # invalid descriptor
    # the override bits pin the choice per env, beyond the rule's limits in both directions; ignored where no helper instantiation exists
    flag = {n: int(re.search(r"#define QR_FLAG_%s\s+(\d+)u" % n, hdr).group(1))
            for n in ("FORCE_HELPER", "NO_HELPER", "FORCE_HELPER_ROLLOUT", "NO_HELPER_ROLLOUT", "CALLER_RESETS")}
    assert len(set(flag.values()) | {AR, 2, 4}) == 8                       # eight distinct bits
    assert info(0, 1 << 20, AR)[1:] == (16384, 64)
    assert info(0, 1 << 20, AR | flag["FORCE_HELPER"])[1:] == (16384, 128) and info(0, 65536, AR | flag["NO_HELPER"])[1:] == (1024, 64)
    assert info(2, 32768, AR | flag["NO_HELPER"])[2] == 64 and info(1, 1 << 19, AR | flag["FORCE_HELPER_ROLLOUT"], n_steps=8)[2] == 128
    # one bit pair per launch FAMILY: a choice timed on step() does not reach the rollouts, and the other way round
    assert info(1, 1 << 19, AR | flag["FORCE_HELPER"], n_steps=8)[2] == 64 and info(0, 65536, AR | flag["NO_HELPER"], n_steps=100)[2] == 128
    assert info(0, 65536, AR | flag["NO_HELPER_ROLLOUT"], n_steps=100)[2] == 64 and info(0, 65536, AR | flag["NO_HELPER_ROLLOUT"])[2] == 128
    assert info(0, 1 << 20, AR | flag["FORCE_HELPER_ROLLOUT"])[2] == 64
    # the thresholds the host-side autotuner reads
    thr = L.launch_thresholds()
    assert thr == {"step_quad": 3328, "step_wrappers": 2560, "rollout": 1024} or any(k in os.environ for k in ("QR_HELPER_GRID", "QR_HELPER_GRID_WRAP", "QR_HELPER_GRID_ROLLOUT"))
    assert info(0, 65536, AR | flag["FORCE_HELPER"], layout=1)[2] == 64 and info(0, 65536, AR | flag["FORCE_HELPER"], w_adapt=3.0)[2] == 64
    assert info(0, 65536, flag["FORCE_HELPER"])[2] == 64                   # no such instantiation without in-launch resets: ignored


def test_launch_plan_names_the_instantiation_without_gpu():
    """qr_launch_plan (host-only): the launcher's own decision with the env's substeps and, for qr_rollout_actor, the actor form —
    instantiation name as a rocprofv3 trace prints it, geometry, number of launches (chunked actor rollouts), counter key;
    qr_instance_table lists all 175 instantiations and every plan's key is one of them."""
    L = _lib()
    lib = L.load()
    AR = L.FLAG_AUTO_RESET

    def plan(kind, n, flags, n_steps=1, substeps=1, actor=0, layout=0, goal_mode=None, rc=0):
        e = L.QrEnv()
        lib.qr_default_coeffs(C.byref(e.coeffs))
        e.kind, e.layout, e.num_envs, e.pos_vel, e.att_rate, e.flags = kind, layout, n, 0x1000, 0x2000, flags
        if goal_mode is not None:
            e.goal_mode, e.traj, e.goal = goal_mode, 0x7000, 0x8000
        p = L.QrLaunchPlan()
        assert lib.qr_launch_plan(C.byref(e), n_steps, substeps, actor, C.byref(p)) == rc
        return p

    table = L.instance_table()
    assert len(table) == 175 and len(set(table)) == 175
    p = plan(0, 65536, AR)
    assert p.name == b"qr::step_kernel<0,float,double,64,0,0,0,1,1,1,0>" and (p.grid, p.block, p.launches) == (1024, 128, 1) and p.key in table
    assert L.describe_key(p.key) == "mixed/quad TRAJ=0 ADAPT=0 POLICY=0 SINGLE=1 HELP=1 HREW=1 MAG=0"
    # substeps move the thresholds (3328 -> 2560 tiles for Quad-v0, 2560 -> 2048 for the wrappers) and keep the reward on the helper wave
    assert plan(0, 64 * 3000, AR).help == 1 and plan(0, 64 * 3000, AR, substeps=2).help == 0
    assert plan(1, 64 * 2300, AR).help == 1 and plan(1, 64 * 2300, AR, substeps=4).help == 0
    assert plan(1, 64 * 1664, AR, substeps=4).help == 1 and plan(2, 64 * 1700, AR, substeps=2).help == 0   # (re-measured with the Magnus substep)
    assert plan(0, 64 * 1500, AR).hrew == 0 and plan(0, 64 * 1500, AR, substeps=10).hrew == 1 and plan(0, 64 * 1400, AR).hrew == 1
    assert plan(2, 64 * 1700, AR).hrew == 0 and plan(2, 64 * 1600, AR).hrew == 1
    # two or more substeps in the default layout: the Magnus-substep twin (MAG = 1) of the same instantiation — whatever the launch family,
    # never in the uniform layouts, never for the delta-form (free-run) kernels
    m = plan(0, 65536, AR, substeps=10)
    assert m.mag == 1 and m.name == b"qr::step_kernel<0,float,double,64,0,0,0,1,1,1,1>" and m.key in table and m.key == p.key | 0x1000
    assert L.describe_key(m.key) == "mixed/quad TRAJ=0 ADAPT=0 POLICY=0 SINGLE=1 HELP=1 HREW=1 MAG=1"
    assert plan(1, 65536, AR, n_steps=32, substeps=2, actor=1).mag == 1 and plan(2, 65536, AR, n_steps=8, substeps=3).mag == 1
    assert plan(0, 65536, AR, substeps=10, layout=1).mag == 0 and plan(0, 65536, 0, substeps=4).mag == 0 and plan(0, 65536, AR).mag == 0
    # free run (no resets) -> the rate-adaptive one-step kernel; the caller's reset promise -> the plain one; uniform layouts: no SINGLE
    assert (plan(0, 65536, 0).adapt, plan(0, 65536, 0).single) == (1, 1) and plan(0, 65536, L.FLAG_CALLER_RESETS).adapt == 0
    assert plan(0, 65536, AR, layout=1).name == b"qr::step_kernel<0,double,double,64,0,0,0,0,0,1,0>"
    # qr_rollout_actor: PPO / SAC forms, helper wave up to 1024 tiles, beyond it chunks of 1024 tiles (one launch after the other)
    a = plan(1, 65536, AR, n_steps=32, actor=1)
    assert (a.policy, a.help, a.block, a.launches, a.grid) == (1, 1, 128, 1, 1024)
    b = plan(2, 262144, AR, n_steps=32, actor=2)
    assert (b.policy, b.help, b.launches, b.grid) == (2, 1, 4, 1024) and b.key in table
    c = plan(1, 262144, AR | L.FLAG_NO_HELPER_ROLLOUT, n_steps=32, actor=1)
    assert (c.help, c.launches, c.grid, c.block) == (0, 1, 4096, 64)
    assert plan(1, 65536, AR, n_steps=8, actor=1, goal_mode=L.GOAL_MODE2).name == b"qr::step_kernel<1,float,double,64,2,1,2,0,0,1,0>"
    # argument errors
    plan(0, 65536, AR, actor=1, rc=-2)             # no actor rollouts for Quad-v0
    plan(1, 65536, AR, substeps=0, rc=-3)
    plan(1, 65536, AR, actor=3, rc=-3)
    assert lib.qr_launch_plan(None, 1, 1, 0, C.byref(L.QrLaunchPlan())) == -1
    # every key a plan can name is in the table (a sweep of the decision function's inputs)
    for kind in (0, 1, 2):
        for layout in (0, 1, 2):
            for flags in (0, AR, AR | L.FLAG_NO_HELPER | L.FLAG_NO_HELPER_ROLLOUT, L.FLAG_CALLER_RESETS):
                for n_steps, actor in ((1, 0), (5, 0), (5, 1), (5, 2)):
                    for gm in (None, L.GOAL_MODE0, L.GOAL_MODE5):
                        if actor and kind == 0:
                            continue
                        for sub in (1, 2):
                            assert plan(kind, 70000, flags, n_steps=n_steps, substeps=sub, actor=actor, layout=layout, goal_mode=gm).key in table
    assert L.launch_stats() == {} or all(k in table for k in L.launch_stats())    # (no launches on a box without a GPU)


def test_no_cpu_fallback():
    """Without a GPU the env refuses to exist; without the library the import of the binding raises."""
    _lib()
    from gym_rotor_amd import QuadVecEnv
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU execution path"):
            QuadVecEnv("coupled", 4)
    with pytest.raises(ValueError):
        QuadVecEnv("hexa", 4)
    with pytest.raises(ValueError):
        QuadVecEnv("quad", 4, layout="bf16")
    code = "import os; os.environ['QR_LIB']='/nonexistent/lib.so'; from gym_rotor_amd import _lib; _lib.load()"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gym_rotor_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no torch/NumPy", ""), fn


_GLOO_WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    from gym_rotor_amd import shard_range, all_gather_rows
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    N, T = 1001, 5                      # uneven shards: 501 + 500
    s, e = shard_range(N, rank, world)
    g = torch.arange(T * N, dtype=torch.float32).reshape(T, N)
    full = all_gather_rows(g[:, s:e].contiguous(), N)
    assert full.shape == (T, N) and torch.equal(full, g), rank
    # per-shard env-steps summed like bench.py does
    t = torch.tensor([float(e - s)]); dist.all_reduce(t); assert t.item() == N
    dist.barrier(); dist.destroy_process_group()
    print("ok", rank)
""")


def test_sharding_world_size_2_gloo(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER.format(root=ROOT, port=port))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err
        assert "ok" in out


def test_all_gather_rows_single_process_is_identity():
    from gym_rotor_amd import all_gather_rows
    x = torch.arange(12.).reshape(3, 4)
    assert all_gather_rows(x, 4) is x


def test_torch_custom_ops_are_registered():
    import gym_rotor_amd  # noqa: F401
    for name in ("qr_step", "qr_rollout", "qr_rollout_actor", "qr_error_obs", "qr_reset", "qr_traj_start", "qr_get_state", "qr_set_state", "qr_gae"):
        assert hasattr(torch.ops.gym_rotor_amd, name)
    with pytest.raises(RuntimeError, match="GPU only"):
        torch.ops.gym_rotor_amd.qr_gae(torch.zeros(2, 3), torch.zeros(2, 3, dtype=torch.bool), torch.zeros(3, 3), 0.9, 0.9,
                                       torch.zeros(2, 3), torch.zeros(2, 3))


def test_seed_range():
    """Seeds are 64-bit Philox keys carried as int64 by the torch ops: both paths accept 0 <= seed < 2^63 only."""
    from gym_rotor_amd import QuadVecEnv
    for bad in (-1, 2 ** 63, 2 ** 64):
        with pytest.raises(ValueError, match="seed"):
            QuadVecEnv._check_seed(bad)
    assert QuadVecEnv._check_seed(2 ** 63 - 1) == 2 ** 63 - 1


def test_gymnasium_is_optional():
    """The engine imports and works without gymnasium; the VectorEnv adapter asks for it explicitly."""
    import importlib.util
    import gym_rotor_amd
    if importlib.util.find_spec("gymnasium") is None:
        with pytest.raises(ImportError):
            gym_rotor_amd.as_gymnasium_vector_env(None)


def _integration_snippet():
    """The first python block of INTEGRATION.md §2 (the binding a maintainer would add)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes as C, torch\n.*?)```", text, re.S).group(1)
    return block


def test_integration_md_binding_matches_the_abi():
    """The ctypes structs printed in INTEGRATION.md are the ABI a reader will copy: they must be
    field-for-field the ones gym_rotor_amd/_lib.py (checked against the header above) uses."""
    from gym_rotor_amd import _lib
    block = _integration_snippet()
    assert f"qr_abi_version() == {_lib.ABI_VERSION}" in block
    decl = block[block.index("class QrCoeffs"):block.index("N = 65536")]
    ns = {"C": C}
    exec(decl, ns)
    for name in ("QrCoeffs", "QrEnv", "QrStepOut"):
        doc, real = ns[name], getattr(_lib, name)
        assert C.sizeof(doc) == C.sizeof(real), name
        assert [(n, getattr(doc, n).offset, getattr(doc, n).size) for n, *_ in doc._fields_] == \
               [(n, getattr(real, n).offset, getattr(real, n).size) for n, *_ in real._fields_], name


@pytest.mark.gpu
def test_integration_md_binding_runs():
    """Run that snippet as written (raw ctypes, no gym_rotor_amd import) on the GPU."""
    from gym_rotor_amd import _lib
    block = _integration_snippet().replace('C.CDLL("libquadrotor_hip.so")', f"C.CDLL({_lib.LIB_PATH!r})")
    ns = {}
    exec(block, ns)
    torch.cuda.synchronize()
    obs, rew, done = ns["obs"], ns["rew"], ns["done"]
    assert torch.isfinite(obs).all() and obs.abs().max() <= 1.5
    assert ((rew >= 0) & (rew <= 1) | (rew == -1)).all()
    assert 0 < done.float().mean() < 0.2 or done.sum() == 0


@pytest.mark.gpu
def test_bench_multi_rank_control_flow():
    """bench.py under torch.distributed.run with 2 ranks.  The box has one GPU, so the ranks share it
    through the QR_BENCH_BACKEND=gloo hook (NCCL/RCCL refuses two ranks on one device); everything
    else — env sharding by rank, barrier-bracketed timing, MAX over ranks, rank-0 JSON — is the
    path the 2/4/8-GPU runs take."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ, QR_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "200", "--warmup", "10",
                        "--extras", "0"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2 * d["config"]["envs_per_gpu"] and d["scaling"] == "weak"
    assert d["value"] > 1e9 and abs(d["value"] - d["config"]["global_envs"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d                           # N = 1 only


def test_bench_presets_and_scaling_flags_without_gpu(monkeypatch):
    """bench.py's argument layer (no GPU): --config k sets BASELINE.json configs[k]'s per-GPU shape, --scaling strong takes the
    preset's named total (or --global-envs, or --envs) as the fixed global batch."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")

    def parse(*argv):
        monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
        return bench.parse()

    a = parse()
    assert (a.kind, a.envs, a.substeps, a.scaling, a.config) == ("quad", 65536, 1, "weak", 0)
    a = parse("--config", "3")
    assert (a.kind, a.envs, a.substeps, a.scaling) == ("decoupled", 32768, 1, "weak")
    a = parse("--config", "4", "--scaling", "strong")
    assert (a.kind, a.envs, a.substeps, a.global_envs) == ("quad", 131072, 10, 1048576) and a.action_batches == 32
    a = parse("--config", "3", "--scaling", "strong", "--gpus", "8")
    assert a.global_envs == 262144
    assert parse("--scaling", "strong").global_envs == 65536 and parse("--scaling", "strong", "--envs", "4096").global_envs == 4096
    assert parse("--scaling", "strong", "--global-envs", "777").global_envs == 777
    assert parse("--config", "2").kind == "coupled" and parse("--config", "1").envs == 65536
    # what the default 1-GPU run measures beside the headline: every other BASELINE.json config in its per-GPU and one-GPU shape, the
    # fused rollout and the PPO collection loop — eight rows, each priced with SURVEY 8(d)'s bytes — then the float64 layout and
    # three more fused launches; all of it as FLAT scalar keys of `config` (the driver's record drops nested lists)
    shapes = [(c["kind"], c["envs"], c["substeps"], c["workload"], c["horizon"]) for c in bench.BASELINE_CONFIGS]
    assert shapes == [("coupled", 65536, 1, "step", 1), ("decoupled", 32768, 1, "step", 1), ("decoupled", 262144, 1, "step", 1),
                      ("quad", 131072, 10, "step", 1), ("quad", 1048576, 10, "step", 1), ("quad", 1048576, 1, "step", 1),
                      ("quad", 65536, 1, "rollout", 100), ("coupled", 65536, 1, "rollout_actor", 32)]
    assert all(c["steps"] % c["horizon"] == 0 and c["slabs"] * c["envs"] * 20 < 2 ** 31 for c in bench.BASELINE_CONFIGS + bench.OTHER_ROWS)
    assert [(c["kind"], c["envs"], c["workload"], c.get("layout")) for c in bench.OTHER_ROWS] == [
        ("quad", 65536, "step", "f64"), ("coupled", 65536, "step", "f64"), ("quad", 1048576, "step", "f64"),
        ("coupled", 65536, "rollout", None), ("decoupled", 65536, "rollout_actor", None), ("coupled", 262144, "rollout_actor", None)]
    keys = bench.secondary_keys()
    for k in ("sustained_us", "free_run_us", "c2_coupled65536_us", "c2_coupled65536_frac", "c3_share32768_us", "c3_share32768_frac", "c3_262144_us", "c3_262144_frac",
              "c4_share131072x10_us", "c4_share131072x10_frac", "c4_1Mx10_us", "c4_1Mx10_frac", "quad1Mx1_us", "quad1Mx1_frac", "quad1Mx1_noop_us",
              "rollout_T100_us_per_env_step", "ppo_collect_T32_us_per_env_step", "f64_quad65536_us", "f64_quad1Mx1_us", "f64_coupled65536_us"):
        assert k in keys, k
    # the whole JSON line must fit the driver's 8 KB record with room to spare: a worst-case line (every secondary key with an
    # 18-character value on top of ~2.6 KB of headline, roofline and cpu_baseline fields) stays under bench.MAX_LINE_BYTES
    import json
    assert bench.MAX_LINE_BYTES <= 6144 and len(json.dumps({k: 0.12345678901234567 for k in keys})) + 2700 <= bench.MAX_LINE_BYTES
    assert [bench.algo_bytes_per_env_step(k) for k in ("quad", "coupled", "decoupled")] == [189, 345, 334]
    assert abs(bench.algo_bytes_per_env_step("quad", "rollout", 100) - (21 + 168 / 100)) < 1e-12
    assert abs(bench.algo_bytes_per_env_step("coupled", "rollout_actor", 32) - (129 + 232 / 32)) < 1e-12
    a = parse()
    assert a.extras == 1 and a.extras_budget > 0 and a.actor == "ppo" and parse("--actor", "sac").actor == "sac" and 0 < a.sustain <= 10


def test_bench_gpus_n_spawns_its_own_ranks_without_gpu(monkeypatch):
    """`bench.py --gpus 4` with no WORLD_SIZE: the ranks are started as child processes through torch.distributed.run
    (same arguments, 127.0.0.1 rendezvous) and the children's exit code is the caller's; under a launcher nothing is spawned."""
    import importlib
    import subprocess as sp
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    calls = []

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return R()

    monkeypatch.setattr(sp, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher():
    """`python bench.py --gpus 2` called the way the driver calls the 1-GPU bench: it starts its two ranks itself
    (QR_BENCH_BACKEND=gloo lets them share the one GPU of this box) and relays rank 0's line."""
    import json
    env = dict(os.environ, QR_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "5", "--extras", "0"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2 * d["config"]["envs_per_gpu"] == 131072
    assert d["dtype"] == "mixed f32/f64" and "cpu_baseline" not in d


@pytest.mark.gpu
@pytest.mark.parametrize("flags,want", [
    (["--scaling", "strong"], dict(scaling="strong", kind="quad", global_envs=65536, envs_per_gpu=32768, substeps=1, baseline=None)),
    (["--config", "3"], dict(scaling="weak", kind="decoupled", global_envs=65536, envs_per_gpu=32768, substeps=1, baseline="configs[3]")),
    (["--config", "4", "--scaling", "strong"], dict(scaling="strong", kind="quad", global_envs=1048576, envs_per_gpu=524288, substeps=10, baseline="configs[4]")),
    (["--config", "2", "--scaling", "strong", "--global-envs", "100000"], dict(scaling="strong", kind="coupled", global_envs=100000, envs_per_gpu=50048, substeps=1, baseline="configs[2]")),
])
def test_bench_presets_and_strong_scaling_with_2_ranks(flags, want):
    """SURVEY.md 8(e)'s other runs, one flag each, through the same 2-rank path as above (the ranks share this box's GPU over gloo):
    `--config k` = BASELINE.json configs[k] in its per-GPU shape, `--scaling strong` = a fixed global batch cut into 64-aligned
    shards (shard_range), reported as such; `value` is always the GLOBAL batch over the max-over-ranks time."""
    import json
    env = dict(os.environ, QR_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "4", "--extras", "0"] + flags,
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == want["scaling"] and c["kind"] == want["kind"] and c["substeps"] == want["substeps"]
    assert c["global_envs"] == want["global_envs"] and c["envs_per_gpu"] == want["envs_per_gpu"]
    assert (c["baseline_config"] or "").startswith(want["baseline"] or "") and (want["baseline"] is not None or c["baseline_config"] is None)
    assert abs(d["value"] - want["global_envs"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] and c["state_finite"]


@pytest.mark.gpu
@pytest.mark.parametrize("workload,kind,horizon,per_step_bytes", [("rollout", "quad", 50, 21 + 168 / 50), ("rollout_actor", "coupled", 16, 129 + 232 / 16)])
def test_bench_rollout_workloads(workload, kind, horizon, per_step_bytes):
    """bench.py --workload rollout | rollout_actor: --steps counts env-steps (whole launches of `horizon`), the line is priced with
    the fused launch's OWN algorithmic bytes (per-step rows + 1/H of the working set) and one clock feeds every figure."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--kind", kind, "--horizon", str(horizon),
                        "--steps", str(4 * horizon + 1), "--warmup", "0", "--envs", "8192", "--cpu-seconds", "0", "--extras", "0"],
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["steps"] == 5 * horizon and d["config"]["env_steps_per_launch"] == horizon and d["config"]["workload_kind"] == workload
    assert abs(d["roofline"]["algorithmic_bytes_per_env_step"] - per_step_bytes) < 1e-9
    assert abs(d["roofline"]["avg_launch_us"] - d["ms_per_step"] * 1e3 * horizon) < 1e-9
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert abs(d["roofline"]["achieved"] - per_step_bytes * 8192 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["roofline"]["achieved"]


@pytest.mark.gpu
def test_readme_quick_start_runs():
    text = open(os.path.join(ROOT, "README.md")).read()
    block = re.search(r"```python\n(import torch\nfrom gym_rotor_amd import.*?)```", text, re.S).group(1)
    ns = {}
    exec(block, ns)
    torch.cuda.synchronize()
    assert ns["obs"].shape == (65536, 23) and ns["horizon"]["action"].shape == (32, 65536, 4)
    assert torch.isfinite(ns["horizon"]["obs0"]).all()


@pytest.mark.gpu
def test_bench_runs_over_rccl_with_one_rank():
    """The RCCL code path of the multi-GPU bench on a 1-GPU box: bench.py under torch.distributed.run with ONE rank and
    QR_BENCH_FORCE_DIST=1 initialises the `nccl` process group with device_id, takes the barriers and the device-side
    all_reduce(MAX) of the timed runs, and reports the world size RCCL saw."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, QR_BENCH_FORCE_DIST="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "5",
                        "--extras", "0", "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["n_ranks_rccl"] == 1 and d["n_gpus"] == 1 and d["value"] > 1e9
    assert abs(d["value"] - d["config"]["global_envs"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]   # one clock
    assert abs(d["roofline"]["avg_launch_us"] - d["ms_per_step"] * 1e3) < 1e-9


_NCCL_GATHER = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from gym_rotor_amd import all_gather_rows, QuadVecEnv, RolloutStorage
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(3 * 1000, dtype=torch.float32, device="cuda").reshape(3, 1000)
y = all_gather_rows(x, 1000)                      # RCCL all_gather, world of one
assert torch.equal(x, y) and y.data_ptr() != x.data_ptr()
adv = torch.randn(8, 1000, 2, device="cuda")
a64 = adv.double()
stats = torch.cat([torch.stack([a64.sum((0, 1)), (a64 * a64).sum((0, 1))], 1), torch.full((2, 1), 8000.0, dtype=torch.float64, device="cuda")], 1)
nrm = RolloutStorage.normalize(adv, stats)        # RCCL all_reduce of the 3 x n_agents doubles
ref = (adv - adv.mean((0, 1))) / (adv.reshape(-1, 2).std(0) + 1e-4)
assert torch.allclose(nrm, ref, atol=1e-5)
dist.barrier(); dist.destroy_process_group(); print("ok")
"""


@pytest.mark.gpu
def test_learner_side_collectives_over_rccl(tmp_path):
    """all_gather_rows and the normalisation-statistics all_reduce on the `nccl` (= RCCL) backend, world of one rank."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    script = tmp_path / "g.py"
    script.write_text(_NCCL_GATHER.format(root=ROOT, port=port))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_checkpoint_attitude_conversion():
    """A round-1 checkpoint stores att_rate as [7, N] (q, W); load_state_dict converts it to the smallest-three form the
    kernels read (three smaller components, dropped one positive, its index in the two low mantissa bits of the first)."""
    from gym_rotor_amd.vec_env import _pack_att_rate
    g = torch.Generator().manual_seed(0)
    for dt, ib, tol in ((torch.float64, torch.int64, 1e-15), (torch.float32, torch.int32, 5e-7)):
        q = torch.randn(4, 500, generator=g, dtype=dt); q /= q.norm(dim=0)
        q[:, 0] = torch.tensor([1, 0, 0, 0], dtype=dt); q[:, 1] = torch.tensor([0, 0, -1, 0], dtype=dt)
        W = torch.randn(3, 500, generator=g, dtype=dt)
        p = _pack_att_rate(torch.cat([q, W]))
        assert p.shape == (6, 500) and torch.equal(p[3:], W)
        bits = p[0].contiguous().view(ib)
        idx = bits & 3
        k = torch.stack([(bits & ~3).view(dt), p[1], p[2]])
        w = torch.sqrt(1 - (k * k).sum(0))
        assert (w >= 0.5 - 1e-6).all()                                  # the dropped component is the largest one
        for n in range(500):
            kk = list(k[:, n]); kk.insert(int(idx[n]), w[n])
            dec = torch.stack(kk)
            assert min((dec - q[:, n]).abs().max(), (dec + q[:, n]).abs().max()) <= tol   # q and -q are the same rotation
        assert int(idx[0]) == 0 and int(idx[1]) == 2 and float(p[:3, 0].abs().max()) == 0.0


def test_launch_cache_roundtrip_and_near_threshold(tmp_path, monkeypatch):
    """gym_rotor_amd.launch_cache (host logic of the default autotuner): which grids are worth timing, and the JSON file the choices
    persist in — merge on store, tolerant of a missing / corrupt file, QR_LAUNCH_CACHE moves or disables it."""
    sys.path.insert(0, ROOT)
    from gym_rotor_amd import launch_cache as lc
    assert lc.near_threshold(2560, 2560) and lc.near_threshold(1920, 2560) and lc.near_threshold(3200, 2560)
    assert not lc.near_threshold(1919, 2560) and not lc.near_threshold(3201, 2560) and not lc.near_threshold(1024, 2560)
    assert lc.near_threshold(3072, 2560) and not lc.near_threshold(2048, 0)       # Quad-v0 196 608 envs: timed; no threshold: never
    f = tmp_path / "sub" / "launch.json"
    monkeypatch.setenv("QR_LAUNCH_CACHE", str(f))
    k1 = lc.key("AMD Instinct MI355X", "abi13:2600000", "quad", 3072, "mixed", "external", "default")
    k2 = lc.key("AMD Instinct MI355X", "abi13:2600000", "coupled", 2048, "mixed", "external", "default")
    assert lc.lookup(k1) is None
    assert lc.store(k1, {"default": 9.0, "helper": 8.4, "no_helper": 9.0, "picked": "helper"})
    assert lc.store(k2, {"default": 9.5, "helper": 9.5, "no_helper": 9.2, "picked": "no_helper"})
    assert lc.lookup(k1) == {"picked": "helper", "us": {"default": 9.0, "helper": 8.4, "no_helper": 9.0}} and lc.lookup(k2)["picked"] == "no_helper"
    assert lc.lookup(k1.replace("abi13", "abi14")) is None                        # another library build: its own entries
    f.write_text("{ not json")
    assert lc.lookup(k1) is None and lc.store(k1, {"default": 1.0, "helper": 1.0, "no_helper": 1.0, "picked": "default"}) and lc.lookup(k1)["picked"] == "default"
    monkeypatch.setenv("QR_LAUNCH_CACHE", "off")
    assert lc.path() is None and lc.lookup(k1) is None and not lc.store(k1, {"picked": "helper"})
    monkeypatch.delenv("QR_LAUNCH_CACHE")
    assert lc.path().endswith(os.path.join(".cache", "gym_rotor_amd", "launch.json"))
