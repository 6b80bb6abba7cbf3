#!/bin/bash
# Run on the GPU box (via gpurun): everything DESIGN.md §5 cites, into gpurun_out/<round>/ (copy to profiles/<round>/).
#   tools/collect_evidence.sh r02
set -u
RND=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$RND
mkdir -p "$OUT"
cd "$ROOT"
# (1) rocprofv3 kernel stats + PMC traffic of bench.py per configuration
prof() { tag=$1; shift; bash tools/profile.sh $RND/$tag "$@" > "$OUT/${tag}_profile.log" 2>&1; cp "$OUT/$tag/summary.txt" "$OUT/${tag}_summary.txt"; cp "$OUT/$tag/kernel_stats.csv" "$OUT/${tag}_kernel_stats.csv" 2>/dev/null; tail -3 "$OUT/$tag/stats.log" > "$OUT/${tag}_stats_tail.log" 2>/dev/null; }
prof quad65536 --steps 500
prof quad65536_noreset --steps 500 --no-auto-reset
prof quad1M --envs 1048576 --action-batches 16 --steps 100
prof quad131072x10 --envs 131072 --substeps 10 --steps 300
prof coupled65536 --kind coupled --steps 500
prof decoupled32768 --kind decoupled --envs 32768 --steps 500
prof coupled1M --kind coupled --envs 1048576 --action-batches 16 --steps 60
prof decoupled1M --kind decoupled --envs 1048576 --action-batches 16 --steps 60
# (2) the bench lines themselves (un-profiled)
b() { tag=$1; shift; python3 bench.py --cpu-seconds 0 --extras 0 "$@" > "$OUT/bench_$tag.json" 2>> "$OUT/bench.err"; }
b quad65536 ; b quad65536_steps20 --steps 20 --warmup 5 ; b quad65536_noreset --no-auto-reset
b quad1M --envs 1048576 --action-batches 16 --steps 300 ; b quad131072x10 --envs 131072 --substeps 10
b quad1Mx10 --envs 1048576 --substeps 10 --action-batches 16 --steps 100
b coupled65536 --kind coupled ; b decoupled32768 --kind decoupled --envs 32768 ; b decoupled65536 --kind decoupled
b coupled1M --kind coupled --envs 1048576 --action-batches 16 --steps 100 ; b decoupled1M --kind decoupled --envs 1048576 --action-batches 16 --steps 100
python3 bench.py --cpu-seconds 12 > "$OUT/bench_full_line.json" 2>> "$OUT/bench.err"
# (3) build-time ablations (A/B of libraries), run-time A/B, timelines, instruction issue costs
L=gym_rotor_amd
QR_AB_JSON=$OUT/ab_quad_builds.json QR_AB_KINDS=quad QR_AB_SIZES=65536,1048576 python3 tools/ab_libs.py $L/libquadrotor_hip_q.so $L/libquadrotor_hip_q_norew.so $L/libquadrotor_hip_q_nohelp.so $L/libquadrotor_hip_q_floor.so $L/libquadrotor_hip_q_copy.so $L/libquadrotor_hip_q_spec.so $L/libquadrotor_hip_q_spec_noreset.so $L/libquadrotor_hip_q_pk.so $L/libquadrotor_hip_q_w4.so > "$OUT/ab_quad_builds.txt" 2>&1
tools/sweep_kinds.sh "$L/libquadrotor_hip_nohelp.so $L/libquadrotor_hip.so" "quad:65536 quad:98304 quad:131072 quad:196608 coupled:65536 coupled:98304 coupled:131072 decoupled:32768 decoupled:65536 decoupled:98304 decoupled:131072" > "$OUT/ab_helper_wave.txt" 2>&1
python3 tools/evidence.py > "$OUT/runtime_ab.json" 2> "$OUT/runtime_ab.err"
for ar in 1 0; do QR_LIB=$L/libquadrotor_hip_q_stamps.so python3 tools/stamp_timeline.py --auto-reset $ar --json "$OUT/stamps_quad65536_ar$ar.json" > "$OUT/stamps_quad65536_ar$ar.txt" 2>&1; done
QR_LIB=$L/libquadrotor_hip_q_stamps.so python3 tools/stamp_timeline.py --auto-reset 1 --envs 1048576 --json "$OUT/stamps_quad1M_ar1.json" > "$OUT/stamps_quad1M_ar1.txt" 2>&1
bash tools/rocprof_floor.sh > "$OUT/rocprof_dispatch_floor.txt" 2>&1
bash tools/pmc_insts.sh > "$OUT/pmc_insts.txt" 2>&1
./build/valu_mb > "$OUT/valu_microbench.json" 2> /dev/null
python3 tools/ppo_rollout_bench.py > "$OUT/ppo_rollout.json" 2> "$OUT/ppo_rollout.err"
ls "$OUT"
