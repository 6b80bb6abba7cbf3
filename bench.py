#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched quadrotor step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python bench.py --gpus N ...          (no launcher: starts the N ranks itself as child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Quad-v0, 65 536 envs per GPU, per-env reset-distribution
states, +-10 % randomised parameters, U(-1,1) actions pre-generated on the device, float32
I/O.  A "step" is ONE env.step() launch over the whole batch (qr_step through the C-ABI).
Random actions terminate an episode within ~100 steps, so — like every loop of the reference
that steps an env (main.py:183-186 resets on done) — terminated envs are re-sampled, here
inside the same launch (QR_FLAG_AUTO_RESET, ~1.2 % of the envs per step).  The free-running
variant (--no-auto-reset: step() alone, the population tumbling far beyond termination, where
the rate-adaptive substep count adds work) is timed too and reported under
config.other_reset_mode.  Envs are sharded over ranks
with NO collective on the step path (weak scaling: 65 536 envs per GPU); `value` = all ranks'
env-steps / the max-over-ranks time of the K timed steps, inputs resident in HBM.

Timing.  The K steps are one hipGraph of K captured qr_step launches (default) or K eager calls
(--mode eager).  One timed REPETITION = barrier + synchronize, an untimed lead-in replay of
max(10, 300 - K) steps (so that the K timed launches are queued behind running work — no host
submission latency inside the timed region — and run at busy clocks, not on the ramp that
follows the barrier's idle gap), HIP event, the K steps, HIP event, synchronize + barrier.  For
K < 300 the timed graph holds ceil(300 / K) back-to-back copies of the K steps and the event
time is divided by that count: a graph launch costs a fixed ~9 us on the device, which would
otherwise weigh 8 % at K = 20 and 0.2 % at K = 1000.  Repetitions
are made until >= 20 of them AND >= 50 ms of timed work exist (at most 200); the MEDIAN
repetition is reported, after a MAX over ranks.  ONE clock feeds every figure of the line: the
HIP-event time of the K steps on the launch stream -> `ms_per_step`, `value` = envs x K / that
time, `roofline.achieved` = SURVEY.md 8(d) algorithmic bytes per launch / (that time / K) (the
same number again as `roofline.avg_launch_us`).  The host wall clock of the same repetitions
(which adds the graph submission and the synchronize round trip, ~20 us per repetition whatever
K is) is reported beside it as `wall_ms_per_step`; `--steps 20` and `--steps 1000` therefore
agree.  The default 1-GPU run then replays the headline graph back to back for --sustain seconds
(3) without synchronising in between: `config.sustained_us` / `sustained_steps`, the rate under sustained load.
`roofline.traffic` = HBM bytes per launch from the rocprofv3 PMC passes committed under
profiles/ (tools/profile.sh), when one exists for this configuration.

Secondary figures (rank 0, --extras 1), all through the same harness as the headline and all FLAT scalars under `config` (the
driver's record keeps scalars and drops nested lists; the whole line stays under 6 KB): `free_run_us` (the other reset mode), and
per BASELINE.json configuration `<tag>_us` (us per launch), `<tag>_frac` (SURVEY 8(d) algorithmic bytes / time / 8 TB/s) and
`<tag>_noop_us` (qr_touch over the same buffers: the do-nothing kernel that moves the step's bytes) for
c2_coupled65536, c3_share32768, c3_262144, c4_share131072x10, c4_1Mx10, quad1Mx1; the fused launches per env-step
(`rollout_T100_us_per_env_step`, `ppo_collect_T32_us_per_env_step`, ...); and the float64 layout beside the default one
(`f64_quad65536_us`, `f64_quad1Mx1_us`, `f64_coupled65536_us`).  What each row is and what binds it: DESIGN.md 5.

Other runs, one flag each: `--config 2|3|4` = BASELINE.json configs[2..4] in their per-GPU shape (CoupledWrapper
65 536; DecoupledWrapper 32 768 per GPU; Quad-v0 131 072 per GPU x 10 substeps), `--scaling strong` = a fixed global batch (the
preset's named total, or --global-envs) cut into 64-aligned shards over the ranks, reported as `"scaling": "strong"`.

`cpu_baseline` (rank 0, N=1 only) times the oracle's reference-style single-env path (NumPy RHS
+ scipy DOP853 + ensure_SO3, reset-on-done; oracle/quad_oracle.py) on one host core for ~12 s.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

TRI = {"auto": None, "on": True, "off": False}
# BASELINE.json configs[1..4] in their per-GPU shape (configs[0] is the reference's CPU run: `cpu_baseline`)
PRESETS = {1: dict(kind="quad", envs=65536, substeps=1, total=65536, slabs=64, name="configs[1]: Quad-v0 batched 65 536 envs, 1xMI355X"),
           2: dict(kind="coupled", envs=65536, substeps=1, total=65536, slabs=64, name="configs[2]: CoupledWrapper 65 536 envs + reward/done"),
           3: dict(kind="decoupled", envs=32768, substeps=1, total=262144, slabs=64, name="configs[3]: DecoupledWrapper two-agent, 262 144 envs over 8 GPUs = 32 768 per GPU"),
           4: dict(kind="quad", envs=131072, substeps=10, total=1048576, slabs=32, name="configs[4]: Quad-v0 1 048 576 envs over 8 GPUs = 131 072 per GPU, 10 substeps")}
# What the default 1-GPU run measures BESIDE the headline, each through the same run() harness — hipGraph, lead-in, >= 20
# repetitions, HIP events — as the headline itself: the other BASELINE.json configs in their per-GPU and one-GPU shapes, SURVEY.md
# 8(d)'s fused rollout, configs[2]'s PPO collection loop, and the float64 layout.  `steps` = env-steps per timed repetition;
# `tag` = the prefix of the row's flat keys in `config` (tag_us, tag_frac[, tag_noop_us]; fused launches: tag_us_per_env_step).
BASELINE_CONFIGS = [
    dict(tag="c2_coupled65536", kind="coupled", envs=65536, substeps=1, workload="step", horizon=1, steps=300, slabs=64),
    dict(tag="c3_share32768", kind="decoupled", envs=32768, substeps=1, workload="step", horizon=1, steps=300, slabs=64),
    dict(tag="c3_262144", kind="decoupled", envs=262144, substeps=1, workload="step", horizon=1, steps=300, slabs=32),
    dict(tag="c4_share131072x10", kind="quad", envs=131072, substeps=10, workload="step", horizon=1, steps=300, slabs=32),
    dict(tag="c4_1Mx10", kind="quad", envs=1048576, substeps=10, workload="step", horizon=1, steps=150, slabs=16),
    dict(tag="quad1Mx1", kind="quad", envs=1048576, substeps=1, workload="step", horizon=1, steps=300, slabs=16),
    dict(tag="rollout_T100", kind="quad", envs=65536, substeps=1, workload="rollout", horizon=100, steps=1000, slabs=4),
    dict(tag="ppo_collect_T32", kind="coupled", envs=65536, substeps=1, workload="rollout_actor", horizon=32, steps=960, slabs=4),
]
# the float64 layout beside the default one (what the precision choice costs: DESIGN.md 4), and the fused launches DESIGN.md quotes
OTHER_ROWS = [
    dict(tag="f64_quad65536", kind="quad", envs=65536, substeps=1, workload="step", horizon=1, steps=300, slabs=64, layout="f64"),
    dict(tag="f64_coupled65536", kind="coupled", envs=65536, substeps=1, workload="step", horizon=1, steps=300, slabs=64, layout="f64"),
    dict(tag="f64_quad1Mx1", kind="quad", envs=1048576, substeps=1, workload="step", horizon=1, steps=300, slabs=16, layout="f64"),
    dict(tag="rollout_T100_coupled", kind="coupled", envs=65536, substeps=1, workload="rollout", horizon=100, steps=1000, slabs=4),
    dict(tag="ppo_collect_T32_decoupled", kind="decoupled", envs=65536, substeps=1, workload="rollout_actor", horizon=32, steps=960, slabs=4),
    dict(tag="ppo_collect_T32_coupled262144", kind="coupled", envs=262144, substeps=1, workload="rollout_actor", horizon=32, steps=320, slabs=4),
]


def secondary_keys():
    """The flat `config` keys a full default run adds beside the headline (tests pin them; DESIGN.md 5 explains each)."""
    keys = ["sustained_us", "sustained_steps", "free_run_us"]
    for spec in BASELINE_CONFIGS + OTHER_ROWS:
        tag = spec["tag"]
        if spec["workload"] == "step":
            keys += [f"{tag}_us", f"{tag}_frac"] + ([] if "layout" in spec else [f"{tag}_noop_us"])
        else:
            keys += [f"{tag}_us_per_env_step", f"{tag}_frac"]
    return keys + ["extras_s"]


MAX_LINE_BYTES = 6144  # the driver keeps 8 KB of stdout: the JSON line must fit with room to spare (asserted before printing)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0  # that measured copy ceiling: SURVEY.md 8(d) asks for the fraction of both


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1000)   # K steps per timed repetition (>= 20 repetitions, >= 50 ms of timed work)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--envs", type=int, default=65536, help="envs PER GPU")
    p.add_argument("--kind", default="quad", choices=["quad", "coupled", "decoupled"])
    p.add_argument("--substeps", type=int, default=1)
    p.add_argument("--layout", default="mixed", choices=["mixed", "f64", "f32"])
    p.add_argument("--mode", default="graph", choices=["graph", "eager"])
    p.add_argument("--auto-reset", action=argparse.BooleanOptionalAction, default=True,
                   help="re-sample terminated envs inside the launch (what a loop that steps the reference's env does on done); "
                        "--no-auto-reset = T free-running random-action steps from one reset")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline leg (0 = skip)")
    p.add_argument("--action-batches", type=int, default=64, help="distinct pre-generated [N,A] action slabs cycled through (64 x 1 MiB > L2: every step streams its actions)")
    p.add_argument("--extras", type=int, default=1, help="0: only the headline measurement (used under rocprofv3); 1: + config.other_reset_mode and "
                                                         "config.baseline_configs within --extras-budget; 2: all of them whatever they take")
    p.add_argument("--sustain", type=float, default=3.0, help="seconds of back-to-back replays of the headline graph after the timed repetitions (config.sustained_us; 0 = skip; 1-GPU default run only)")
    p.add_argument("--extras-budget", type=float, default=60.0, help="seconds after which the remaining secondary rows are skipped (--extras 1)")
    p.add_argument("--actor", default="ppo", choices=["ppo", "sac"], help="--workload rollout_actor: the actor form (ppo: parameter log_std, "
                                                                            "tanh-of-mean rule; sac: state-dependent log_std head, tanh-of-sample rule = the POLICY=2 kernel)")
    p.add_argument("--workload", default="step", choices=["step", "rollout", "rollout_actor", "touch"],
                   help="step: one qr_step launch per env-step (the metric's configuration).  rollout: --horizon env-steps per qr_rollout "
                        "launch, state in registers (SURVEY.md 8(d) config 2).  rollout_actor: the PPO collection loop with the actor inside "
                        "the step kernel (qr_rollout_actor; BASELINE configs[2]; --kind coupled|decoupled).  touch: qr_touch, the do-nothing kernel that "
                        "moves one step's bytes (the yardstick of roofline.frac_of_noop_kernel).  --steps counts env-steps in all of them")
    p.add_argument("--horizon", type=int, default=0, help="env-steps per launch of the rollout workloads (default 100 / 32)")
    p.add_argument("--helper", default="auto", choices=["auto", "on", "off"], help="launch rule override: a helper wavefront per tile (QR_FLAG_FORCE_HELPER / QR_FLAG_NO_HELPER; their _ROLLOUT twins for the rollout workloads)")
    p.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4],
                   help="preset = BASELINE.json configs[k] in its per-GPU shape: 1 Quad-v0 65 536 envs (the default run); 2 CoupledWrapper 65 536; "
                        "3 DecoupledWrapper 32 768 per GPU (262 144 over 8 GPUs); 4 Quad-v0 131 072 per GPU x 10 substeps (1 048 576 over 8 GPUs). "
                        "Sets --kind / --envs / --substeps (and the named total for --scaling strong)")
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="weak: --envs per GPU, the global batch grows with --gpus (the driver's contract).  strong: a FIXED global batch — "
                        "--global-envs, default the preset's named total or --envs — cut into 64-aligned shards over the ranks (shard_range)")
    p.add_argument("--global-envs", type=int, default=0, help="global batch of --scaling strong")
    a = p.parse_args()
    if a.config:
        preset = PRESETS[a.config]
        a.kind, a.envs, a.substeps = preset["kind"], preset["envs"], preset["substeps"]
        if a.action_batches == 64:   # (default) keep the action slabs streaming from HBM without outgrowing it
            a.action_batches = preset["slabs"]
    if a.scaling == "strong" and a.global_envs <= 0:
        a.global_envs = PRESETS[a.config]["total"] if a.config else a.envs
    if a.workload == "rollout_actor" and a.kind == "quad":
        a.kind = "coupled"
    if a.horizon <= 0:
        a.horizon = {"step": 1, "touch": 1, "rollout": 100, "rollout_actor": 32}[a.workload]
    if a.workload in ("step", "touch"):
        a.horizon = 1
    a.steps = -(-a.steps // a.horizon) * a.horizon   # whole launches
    return a


def _cpu_single_env_loop(args):
    """`seconds` of the oracle's reference-style single-env step on the calling process' core."""
    kind, seconds, seed = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import quad_oracle as orc
    rng = np.random.default_rng(seed)
    env = orc.RefEnv(kind, orc.sample_params(rng, 1)[0])
    env.state = orc.sample_reset_state(rng, 1)[0]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            _, _, done, _, _ = env.step(rng.uniform(-1, 1, orc.ACTION_DIM[kind]).astype(np.float32))
            n += 1
            if any(done):  # reset-on-done, like the training loop (main.py:212-230)
                env.set_params(orc.sample_params(rng, 1)[0])
                env.state = orc.sample_reset_state(rng, 1)[0]
                env.zero_integrators()
    return n, time.perf_counter() - t0


def cpu_baseline(kind: str, seconds: float):
    """Oracle ('port' of the reference's single-env step): `value` on one host core for `seconds`;
    beside it one independent env per host core and the vectorised NumPy oracle."""
    import multiprocessing as mp
    import platform
    import scipy
    from oracle import quad_oracle as orc
    n, dt = _cpu_single_env_loop((kind, seconds, 0))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container's CPU quota, when it has one, is the honest core count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except Exception:
        pass
    procs = min(cores, 16)  # bounded so that the default bench run stays short; the figure scales linearly beyond
    all_cores = None
    if procs > 1:
        try:
            with mp.get_context("spawn").Pool(procs) as pool:
                res = pool.map(_cpu_single_env_loop, [(kind, min(4.0, seconds), 100 + k) for k in range(procs)])
            all_cores = sum(r[0] / r[1] for r in res)
        except Exception as e:  # a locked-down box: report the single-core figure only
            all_cores = f"unavailable ({type(e).__name__})"
    # best-effort vectorised NumPy line (same maths, all envs at once) for context
    rng = np.random.default_rng(1)
    nb = 65536
    st, pr = orc.sample_reset_state(rng, nb), orc.sample_params(rng, nb)
    ac = rng.uniform(-1, 1, (nb, orc.ACTION_DIM[kind]))
    t1 = time.perf_counter()
    orc.step_batch(kind, st, ac, pr, None, np.zeros((nb, 8)))
    vec = nb / (time.perf_counter() - t1)
    cpu = platform.processor() or ""
    try:
        cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} single-env {kind} steps of oracle.RefEnv (NumPy RHS + scipy DOP853, reset-on-done) in {dt:.1f}s, 1 core",
            "vectorised_numpy_value": vec, "multi_process_value": all_cores, "multi_process_procs": procs, "host_cores": cores,
            "cpu_model": cpu, "numpy": np.__version__, "scipy": scipy.__version__,
            # `value` is the ORACLE's port of the single-env step on this box.  The reference itself, timed in the survey
            # container: 0.5-0.7 k steps/s per core — its own Python overhead (ensure_SO3 through
            # np.isclose, scipy Rotation objects) is larger than the port's
            "reference_itself_steps_per_s_per_core": "500-700 (survey container, BASELINE.md 1)"}


def committed_traffic(kind, envs, layout, auto_reset, substeps=1, workload="step", horizon=1):
    """HBM bytes per launch from the PMC profile committed under profiles/ (None if absent); the newest file wins."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None
    for fn in sorted(os.listdir(pdir)):
        if fn.endswith("_traffic.json"):
            try:
                for rec in json.load(open(os.path.join(pdir, fn))):
                    if (rec["kind"], rec["envs"], rec["layout"], rec["auto_reset"], rec.get("substeps", 1), rec.get("workload", "step"),
                            rec.get("env_steps_per_launch", 1)) == (kind, envs, layout, auto_reset, substeps, workload, horizon):
                        best = rec
            except Exception:
                pass
    return best


def _init_dist():
    """(dist module or None, rank, local_rank, world, backend).  torch.distributed is initialised when
    WORLD_SIZE > 1 — or when QR_BENCH_FORCE_DIST=1 (a world of ONE rank under torch.distributed.run: lets
    a 1-GPU box execute the exact RCCL code path of the 2/4/8-GPU runs)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and os.environ.get("QR_BENCH_FORCE_DIST") != "1":
        return None, rank, local_rank, world, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    # QR_BENCH_BACKEND=gloo is a test hook: it lets N ranks share the GPUs that exist (rank -> device
    # local_rank % device_count) so that the multi-rank control flow can be exercised on a 1-GPU box
    backend = os.environ.get("QR_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
    else:
        local_rank = local_rank % torch.cuda.device_count()
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, rank, local_rank, world, backend


def _ensure_library(dist, local_rank):
    """A checkout without the (git-ignored) library: local rank 0 builds it (to a temporary name, renamed
    into place), everyone else waits.  Every rank takes the same decision and the same barrier."""
    lib = os.path.join(ROOT, "gym_rotor_amd", "libquadrotor_hip.so")
    if os.environ.get("QR_LIB"):
        return
    need = torch.tensor([0 if os.path.exists(lib) else 1], dtype=torch.int32)
    if dist is not None:  # rank 0's view decides for all (a later rank may already see the finished file)
        if dist.get_backend() == "nccl":
            need = need.cuda()
        dist.broadcast(need, src=0)
    if int(need.item()):
        if local_rank == 0:
            import subprocess
            tmp = lib + f".tmp{os.getpid()}"
            subprocess.run(["make", "-C", os.path.join(ROOT, "gym_rotor_amd", "csrc"), f"OUT={tmp}"], check=True, stdout=sys.stderr)
            os.replace(tmp, lib)
        if dist is not None:
            dist.barrier()


def _spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` called without a launcher (no WORLD_SIZE in the environment): start the N ranks
    as CHILD processes — `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` — relay
    their output (rank 0 prints the JSON line) and return their exit code.  Decided before anything in this
    process touches the GPU; nothing is exec'ed over a process that did."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def algo_bytes_per_env_step(kind: str, workload: str = "step", H: int = 1) -> float:
    """SURVEY.md 8(d) algorithmic bytes per env-step.  One-step launches: the per-kind figure + 24 B of per-env parameters
    (Quad-v0 165 + 24 = 189, Coupled 321 + 24, Decoupled 310 + 24).  Rollout launches: per step only what crosses memory EVERY
    step — action row in (or, with the actor in the kernel, action + log-prob rows out), reward, done, [observation rows] —
    plus the launch's once-per-horizon share of the working set (state r/w 144, params 24, integrators r/w 64)."""
    from gym_rotor_amd.constants import ALGO_BYTES, ALGO_BYTES_PARAMS
    if workload == "step":
        return ALGO_BYTES[kind] + ALGO_BYTES_PARAMS
    per_step_io = {"quad": 16 + 4 + 1, "coupled": 16 + 92 + 4 + 1, "decoupled": 20 + 72 + 8 + 2}[kind]
    if workload == "rollout_actor":
        per_step_io += {"coupled": 16, "decoupled": 20}[kind]          # action AND log-prob rows are written
    once = 144 + 24 + (64 if kind != "quad" else 0)
    return per_step_io + once / H


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(a.gpus))
    dist, rank, local_rank, world, backend = _init_dist()
    n_gpus = world
    if a.gpus != world and rank == 0:
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    _ensure_library(dist, local_rank)
    from gym_rotor_amd import QuadVecEnv, shard_range, _lib
    if a.scaling == "strong":      # a fixed global batch, this rank's 64-aligned shard of it (no collective on the step path either way)
        G = a.global_envs
        lo, hi = shard_range(G, rank, world)
        N, env_offset = hi - lo, lo
    else:                          # weak: --envs per GPU
        N, env_offset, G = a.envs, rank * a.envs, a.envs * world
    auto_reset = a.auto_reset
    on_dev = dist is None or backend == "nccl"

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run(w, ar: bool, timed: bool, sustain_s: float = 0.0):
        """`w` = the workload (kind, envs, env_offset, substeps, workload, horizon, steps, slabs, helper, actor).  W warm-up
        steps, then repetitions of exactly K = w.steps timed env-steps; returns the per-repetition (HIP-event ms, wall ms) lists
        and a few facts about the final state.  timed=False: no cross-rank barrier (rank-0-only secondary measurements).
        sustain_s > 0 (graph mode, in-launch resets): afterwards the timed graph is replayed back to back for about that many seconds
        with NO synchronisation in between — the rate the chip holds under sustained load (its clocks under power, not a 4 ms burst)."""
        H, K, n = w.horizon, w.steps, w.envs
        env = QuadVecEnv(w.kind, n, device=dev, seed=0, substeps=w.substeps, layout=getattr(w, "layout", a.layout), use_UDM=True,
                         auto_reset=ar, env_offset=w.env_offset, **{"helper" if w.workload in ("step", "touch") else "helper_rollout": TRI[w.helper]},
                         **({"obs_rows": True} if w.workload == "rollout_actor" else {}))

        def fresh():  # the timed steps start from reset-distribution states (configs[1])
            env.reset("train")
            if w.kind != "quad":
                env.get_norm_error_state()

        fresh()
        gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
        if w.workload in ("step", "touch"):
            acts = [torch.rand(n, env.action_dim, device=dev, generator=gen) * 2 - 1 for _ in range(w.slabs)]
            if w.workload == "step":
                launch = lambda i: env.step(acts[i % len(acts)])                  # noqa: E731
            else:   # the do-nothing kernel over the same buffers and action slabs: what moving the step's bytes costs on this box
                launch = lambda i: env.touch(acts[i % len(acts)])                 # noqa: E731
            last_done = lambda: env._done                                          # noqa: E731
        elif w.workload == "rollout":   # [H, N, A] action slabs (4 x 105 MB at H = 100: streamed from HBM) and preallocated outputs
            acts = [torch.rand(H, n, env.action_dim, device=dev, generator=gen) * 2 - 1 for _ in range(min(w.slabs, 4))]
            ro = env.rollout(acts[0])
            launch = lambda i: env.rollout(acts[i % len(acts)], out=ro)            # noqa: E731
            last_done = lambda: ro["terminated"][H - 1]                            # noqa: E731
        else:                           # the actor(s) inside the step kernel, exploration noise drawn in the kernel
            from gym_rotor_amd import random_actors
            actors = random_actors(w.kind, dev, generator=torch.Generator(device=dev).manual_seed(7), log_std=-0.5,
                                   **({"algo": w.actor} if w.actor != "ppo" else {}))
            po = env.rollout_actor(actors, H)
            pout = {k: v for k, v in po.items() if k != "obs"}
            launch = lambda i: env.rollout_actor(actors, H, out=pout)              # noqa: E731
            last_done = lambda: pout["terminated"][H - 1]                          # noqa: E731
        n_launch = K // H
        for i in range(-(-a.warmup // H)):
            launch(i)
        # untimed lead-in of every repetition: long enough (~1.5 ms) to bring the chip back to its busy clocks after the
        # barrier's idle gap — the first ~50 launches after an idle period run 5-20 % slow — and to keep the queue ahead
        n_lead = max(10, 300 - K) if H == 1 else max(1, 300 // H)     # (in launches)
        # a graph launch costs a fixed ~9 us on the device whatever it holds (measured: the 20-step graph ran 8 % slower per step
        # than the 1000-step one): for small K the timed graph holds `copies` back-to-back copies of the K steps
        copies = max(1, -(-300 // K)) if a.mode == "graph" else 1
        graph = lead = None
        side = torch.cuda.Stream(dev)
        if a.mode == "graph":
            torch.cuda.synchronize(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            graph, lead = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for i in range(n_launch * copies):
                        launch(i)
                with torch.cuda.graph(lead, stream=side):
                    for i in range(n_lead):
                        launch(n_launch + i)
            torch.cuda.current_stream(dev).wait_stream(side)
            # untimed: the first replay uploads the graph; keep replaying for ~50 ms so the timed
            # repetitions run at the clocks a training loop sees, not at the idle-to-busy ramp
            t_w = time.perf_counter()
            while True:
                graph.replay()
                torch.cuda.synchronize(dev)
                if time.perf_counter() - t_w > 0.05:
                    break

        def issue_lead():
            if lead is not None:
                lead.replay()
            else:
                for i in range(n_lead):
                    launch(n_launch + i)

        def issue_steps():
            if graph is not None:
                graph.replay()
            else:
                for i in range(n_launch):
                    launch(i)

        dev_ms, wall_ms = [], []
        reps_min, reps_max, budget_ms = 20, 200, 50.0
        while True:
            if not ar:
                fresh()  # free run: every repetition is the K steps that follow one reset
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier() if timed else torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            issue_lead()          # untimed lead-in: the timed launches queue up behind running work
            ev0.record()
            issue_steps()         # exactly K steps
            ev1.record()
            barrier() if timed else torch.cuda.synchronize(dev)
            wall_ms.append((time.perf_counter() - t0) * 1e3)
            dev_ms.append(ev0.elapsed_time(ev1) / copies)
            done_reps = len(dev_ms)
            stop = done_reps >= reps_max or (done_reps >= reps_min and sum(dev_ms) * copies >= budget_ms)
            if dist is not None and timed:  # all ranks stop together (the barrier count must match)
                flag = torch.tensor([1 if stop else 0], dtype=torch.int32, device=dev if on_dev else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                stop = bool(flag.item())
            if stop:
                break
        sustained = None
        if sustain_s > 0 and graph is not None and ar:
            per_replay_ms = float(np.median(dev_ms)) * copies
            n_rep = max(1, int(sustain_s * 1e3 / per_replay_ms))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            issue_lead()
            ev0.record()
            for _ in range(n_rep):
                graph.replay()
            ev1.record()
            torch.cuda.synchronize(dev)
            sustained = (ev0.elapsed_time(ev1) / (n_rep * copies), n_rep * copies * K)   # (ms per K steps, env-steps per env in the window)
        finite = bool(torch.isfinite(env.get_current_state()).all())
        done_rate = float(last_done().float().mean())
        tuned = env.autotune_report   # None unless a recorded choice of the launch cache applies (grids near a threshold of the launch rule)
        launch_rule = "compiled rule" if tuned is None else f"{tuned.get('picked')} ({tuned.get('source', 'timed')})"
        plan = env.launch_plan(H, actor=(w.actor if w.workload == "rollout_actor" else None))   # the launcher's own decision, this env's substeps
        if w.workload == "touch":
            plan = dict(plan, name=f"qr::touch_kernel<{_lib.KIND_ID[w.kind]},...>", block=64, launches=1)
        return dev_ms, wall_ms, finite, done_rate, (plan, launch_rule), n_lead * H, copies, sustained

    from types import SimpleNamespace as NS
    head = NS(kind=a.kind, envs=N, env_offset=env_offset, substeps=a.substeps, workload=a.workload, horizon=a.horizon, steps=a.steps,
              slabs=a.action_batches, helper=a.helper, actor=a.actor, layout=a.layout)
    sustain_s = a.sustain if (world == 1 and a.extras and a.workload == "step") else 0.0
    dev_ms, wall_ms, finite, done_rate, kinfo, n_lead, copies, sustained = run(head, auto_reset, True, sustain_s)
    reps = len(dev_ms)
    med_dev, med_wall = float(np.median(dev_ms)), float(np.median(wall_ms))
    tmax = torch.tensor([med_dev, med_wall], dtype=torch.float64, device=dev if on_dev else "cpu")
    n_ranks_rccl = None
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        if backend == "nccl":  # the world as RCCL sees it: every rank contributes 1
            ones = torch.ones(1, dtype=torch.int32, device=dev)
            dist.all_reduce(ones)
            n_ranks_rccl = int(ones.item())
    med_dev, med_wall = float(tmax[0]), float(tmax[1])

    if rank == 0:
        H = a.horizon
        ms_per_step = med_dev / a.steps              # THE clock of this line: HIP events around the K steps
        launch_us = ms_per_step * 1e3 * H            # duration of one launch (= H env-steps)
        wall_ms_per_step = med_wall / (a.steps * copies + n_lead)
        algo = algo_bytes_per_env_step(a.kind, "step" if a.workload == "touch" else a.workload, H)
        achieved = algo * N * H / (launch_us * 1e-6) / 1e9
        plan, launch_rule = kinfo
        traffic = committed_traffic(a.kind, N, a.layout, auto_reset, a.substeps, a.workload, H)
        # the committed profile's clock for this configuration: rocprofv3 --kernel-trace for kernels it does not inflate (>= ~6 us),
        # for the shorter ones the in-kernel real-time-clock span + launch gap of the QR_SPAN build (tools/span_timeline.py), both
        # in profiles/rNN_traffic.json (`profile_period_us`)
        prof_us = (traffic or {}).get("profile_period_us") or (traffic or {}).get("rocprofv3_kernel_mean_us")
        what = {"step": "one qr_step launch per env-step", "touch": "qr_touch: the step's bytes, no arithmetic",
                "rollout": f"fused rollout, {H} env-steps per qr_rollout launch",
                "rollout_actor": f"PPO collection, {H} env-steps per qr_rollout_actor launch, actor inside"}[a.workload]
        cfg = {"workload": (f"configs[1]: Quad-v0 {N} envs/GPU, random actions, fp32 I/O, "
                            + ("in-launch resets" if auto_reset else "free run") if a.kind == "quad" and not a.config else
                            (PRESETS[a.config]["name"] if a.config else f"{a.kind} wrapper, {N} envs/GPU"))[:118],
               "launch": what, "workload_kind": a.workload, "env_steps_per_launch": H,
               "baseline_config": (PRESETS[a.config]["name"][:118] if a.config else None), "kind": a.kind, "envs_per_gpu": N, "global_envs": G,
               "substeps": a.substeps, "state_layout": a.layout, "io_dtype": "f32", "auto_reset": auto_reset,
               "done_rate_last_step": round(done_rate, 5), "launch_mode": a.mode,
               "parallelism": f"env-shard x{n_gpus}, no collective" + (f", strong scaling of {G} envs" if a.scaling == "strong" else ""),
               "state_finite": finite, "n_ranks_rccl": n_ranks_rccl, "reps": reps, "lead_in_steps": n_lead,
               "wall_ms_per_step": round(wall_ms_per_step, 7)}
        out = {
            "metric": "quadrotor env-steps/sec at 65 536 envs; 1/2/4/8 MI355X + CPU ref",
            "value": G / (ms_per_step * 1e-3), "unit": "env-steps/s", "n_gpus": n_gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": {"mixed": "mixed f32/f64", "f64": "f64", "f32": "f32"}[a.layout], "data": "synthetic",
            "config": cfg,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": (traffic or {}).get("bytes_per_launch"),
                         "kernel": plan["name"], "grid": plan["grid"], "block": plan["block"], "launches_per_step": plan["launches"],
                         "launch_rule": launch_rule, "avg_launch_us": launch_us,
                         "algorithmic_bytes_per_env_step": algo,
                         "frac_of_copy_ceiling": round(achieved / HBM_COPY_GBS, 4), "copy_ceiling": HBM_COPY_GBS,
                         # the same bytes moved by a kernel that computes nothing (qr_touch), measured below on this box: the
                         # ceiling of ANY one-launch-per-step design at this size; null when extras are off
                         "noop_kernel_us": None, "frac_of_noop_kernel": None,
                         # the SAME algorithmic bytes on the committed profile's clock (profiles/): null without a committed profile
                         "profile_period_us": prof_us,
                         "frac_profile_clock": (round(algo * N * H / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if prof_us else None)},
        }
        if n_gpus == 1 and a.extras and a.workload == "step":
            t_extra = time.perf_counter()

            def us_per_launch(w, ar=True):
                dms = run(w, ar, False)[0]
                torch.cuda.empty_cache()
                return float(np.median(dms)) * 1e3 / w.steps * w.horizon

            def touch_of(w):   # the do-nothing kernel over the same configuration
                return NS(**{**vars(w), "workload": "touch", "horizon": 1, "steps": 300})

            if sustained is not None:   # the headline launch replayed back to back for ~a.sustain seconds, no synchronisation in between
                cfg["sustained_us"], cfg["sustained_steps"] = round(sustained[0] * 1e3 / a.steps * H, 4), int(sustained[1])
            noop = us_per_launch(touch_of(head))
            out["roofline"]["noop_kernel_us"] = round(noop, 4)
            out["roofline"]["frac_of_noop_kernel"] = round(noop / launch_us, 4)
            # the other reset mode of the headline workload (no reset inside step = the reference's own semantics: the population
            # flies on beyond termination and the rate-adaptive kernel is the one launched)
            cfg["free_run_us" if auto_reset else "auto_reset_us"] = round(us_per_launch(head, not auto_reset), 4)
            # every other BASELINE.json config (per-GPU and one-GPU shapes), the fused launches and the float64 layout
            skipped = []
            for spec in BASELINE_CONFIGS + OTHER_ROWS:
                tag = spec["tag"]
                if a.extras < 2 and time.perf_counter() - t_extra > a.extras_budget:
                    skipped.append(tag)
                    continue
                w = NS(kind=spec["kind"], envs=spec["envs"], env_offset=0, substeps=spec["substeps"], workload=spec["workload"],
                       horizon=spec["horizon"], steps=spec["steps"], slabs=spec["slabs"], helper="auto", actor="ppo",
                       layout=spec.get("layout", a.layout))
                us = us_per_launch(w)
                ab = algo_bytes_per_env_step(w.kind, w.workload, w.horizon)
                frac = ab * w.envs * w.horizon / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
                if w.workload == "step":
                    cfg[f"{tag}_us"], cfg[f"{tag}_frac"] = round(us, 3), round(frac, 4)
                    if "layout" not in spec:
                        cfg[f"{tag}_noop_us"] = round(us_per_launch(touch_of(w)), 3)
                else:
                    cfg[f"{tag}_us_per_env_step"], cfg[f"{tag}_frac"] = round(us / w.horizon, 4), round(frac, 4)
            if skipped:
                cfg["rows_skipped"] = ",".join(skipped)[:118]
            cfg["extras_s"] = round(time.perf_counter() - t_extra, 2)
        if n_gpus == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(a.kind, a.cpu_seconds)
        line = json.dumps(out)
        assert len(line) <= MAX_LINE_BYTES, f"bench line is {len(line)} bytes (> {MAX_LINE_BYTES}): the driver's record would lose its tail"
        print(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
