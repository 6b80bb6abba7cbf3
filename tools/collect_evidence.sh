#!/bin/bash
# Run on the GPU box (via gpurun): everything DESIGN.md §5 cites, into gpurun_out/<round>/ (copy to profiles/<round>/).
#   tools/collect_evidence.sh r05 [profiles|bench|ab|tests|all]
# The measurement-only builds are NOT pushed with the repo (.gpurunignore: build/evidence/): they are built here first.
set -u
RND=${1:-r05}; WHAT=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$RND
EV=build/evidence
mkdir -p "$OUT"
cd "$ROOT"
# the product library is rebuilt HERE from the pushed sources, whatever .so rode along (round 4 once measured a stale one)
make -B -C gym_rotor_amd/csrc > "$OUT/product_build.log" 2>&1 || { echo "product build FAILED"; tail -5 "$OUT/product_build.log"; exit 1; }
if [ "$WHAT" = all ] || [ "$WHAT" = ab ]; then
  make -C gym_rotor_amd/csrc evidence-libs > "$OUT/evidence_libs_build.log" 2>&1 || { echo "evidence-libs build FAILED"; tail -5 "$OUT/evidence_libs_build.log"; }
fi
if [ "$WHAT" = all ] || [ "$WHAT" = profiles ]; then
# (1) rocprofv3 kernel stats + PMC traffic + instruction counters of bench.py per configuration
prof() { tag=$1; shift; bash tools/profile.sh $RND/$tag "$@" > "$OUT/${tag}_profile.log" 2>&1; cp "$OUT/$tag/summary.txt" "$OUT/${tag}_summary.txt"; cp "$OUT/$tag/kernel_stats.csv" "$OUT/${tag}_kernel_stats.csv" 2>/dev/null; tail -3 "$OUT/$tag/stats.log" > "$OUT/${tag}_stats_tail.log" 2>/dev/null; rm -rf "$OUT/$tag"; }
prof quad65536 --steps 500
prof quad65536_noreset --steps 500 --no-auto-reset
prof quad1M --envs 1048576 --action-batches 16 --steps 100
prof quad131072x10 --envs 131072 --substeps 10 --steps 300
prof quad1Mx10 --envs 1048576 --substeps 10 --action-batches 16 --steps 60
prof coupled65536 --kind coupled --steps 500
prof decoupled32768 --kind decoupled --envs 32768 --steps 500
prof decoupled262144 --kind decoupled --envs 262144 --action-batches 32 --steps 200
prof coupled1M --kind coupled --envs 1048576 --action-batches 16 --steps 60
prof decoupled1M --kind decoupled --envs 1048576 --action-batches 16 --steps 60
prof rollout_quad65536_T100 --workload rollout --horizon 100 --steps 2000
prof rollout_actor_coupled65536_T32 --workload rollout_actor --kind coupled --horizon 32 --steps 960
prof rollout_actor_sac_coupled65536_T32 --workload rollout_actor --kind coupled --horizon 32 --steps 960 --actor sac
prof rollout_coupled65536_T100 --workload rollout --kind coupled --horizon 100 --steps 1000
prof rollout_actor_coupled262144_T32 --workload rollout_actor --kind coupled --envs 262144 --horizon 32 --steps 320
# the do-nothing kernel (qr_touch) under the same tool: what rocprofv3 reports for a kernel that only moves the step's bytes
prof touch_quad65536 --workload touch --steps 500
prof touch_quad1M --workload touch --envs 1048576 --action-batches 16 --steps 100
# kernel span + launch gap on the device's own clock (the profile clock of the kernels rocprofv3 inflates): the light stamp build
make -C gym_rotor_amd/csrc span-lib > "$OUT/span_lib_build.log" 2>&1
SPAN_CFGS="quad:65536:1 coupled:65536:1 decoupled:32768:1 quad:131072:10 quad:131072:1 quad:1048576:1 decoupled:262144:1 quad:65536:1:rollout:100 coupled:65536:1:rollout_actor:32"
QR_LIB=$PWD/build/ab/libquadrotor_hip_span.so python3 tools/span_timeline.py --json "$OUT/span_timeline.json" $SPAN_CFGS > /dev/null 2> "$OUT/span_timeline.txt"
python3 tools/span_timeline.py --json "$OUT/span_product_hip_events.json" $SPAN_CFGS > /dev/null 2> "$OUT/span_product_hip_events.txt"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
# (2) the bench lines themselves (un-profiled)
b() { tag=$1; shift; python3 bench.py --cpu-seconds 0 --extras 0 "$@" > "$OUT/bench_$tag.json" 2>> "$OUT/bench.err"; }
b quad65536 ; b quad65536_steps20 --steps 20 --warmup 5 ; b quad65536_noreset --no-auto-reset
b quad1M --envs 1048576 --action-batches 16 --steps 300 ; b quad131072x10 --envs 131072 --substeps 10
b quad1Mx10 --envs 1048576 --substeps 10 --action-batches 16 --steps 100
b coupled65536 --kind coupled ; b decoupled32768 --kind decoupled --envs 32768 ; b decoupled65536 --kind decoupled
b decoupled262144 --kind decoupled --envs 262144 --action-batches 32 --steps 300
b coupled1M --kind coupled --envs 1048576 --action-batches 16 --steps 100 ; b decoupled1M --kind decoupled --envs 1048576 --action-batches 16 --steps 100
b rollout_quad65536_T100 --workload rollout --horizon 100 --steps 2000
b rollout_actor_coupled65536_T32 --workload rollout_actor --kind coupled --horizon 32 --steps 960
b rollout_actor_sac_coupled65536_T32 --workload rollout_actor --kind coupled --horizon 32 --steps 960 --actor sac
b rollout_actor_sac_decoupled65536_T32 --workload rollout_actor --kind decoupled --horizon 32 --steps 960 --actor sac
b rollout_actor_decoupled65536_T32 --workload rollout_actor --kind decoupled --horizon 32 --steps 960
b rollout_coupled65536_T100 --workload rollout --kind coupled --horizon 100 --steps 1000
b rollout_actor_coupled262144_T32 --workload rollout_actor --kind coupled --envs 262144 --horizon 32 --steps 320
b config2 --config 2 ; b config3 --config 3 ; b config4 --config 4
python3 bench.py > "$OUT/bench_full_line.json" 2>> "$OUT/bench.err"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_full_line_steps20.json" 2>> "$OUT/bench.err"
python3 tools/eager_cost.py > "$OUT/eager_cost.json" 2>> "$OUT/bench.err"
b f64_quad65536 --layout f64 ; b f64_coupled65536 --layout f64 --kind coupled ; b f64_quad1M --layout f64 --envs 1048576 --action-batches 16 --steps 300
python3 tools/noop_yardstick.py > "$OUT/noop_yardstick.json" 2> "$OUT/noop_yardstick.err"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = ab ]; then
# (3) build-time ablations (A/B of libraries), run-time A/B, timelines, microbenchmarks
QR_AB_JSON=$OUT/ab_quad_builds.json QR_AB_KINDS=quad QR_AB_SIZES=65536,1048576 python3 tools/ab_libs.py $EV/libquadrotor_hip_q.so $EV/libquadrotor_hip_q_aux0.so $EV/libquadrotor_hip_q_norew.so $EV/libquadrotor_hip_q_nohelp.so $EV/libquadrotor_hip_q_floor.so $EV/libquadrotor_hip_q_copy.so > "$OUT/ab_quad_builds.txt" 2>&1
tools/sweep_kinds.sh "$EV/libquadrotor_hip_nohelp.so $EV/libquadrotor_hip_aux0.so gym_rotor_amd/libquadrotor_hip.so" "quad:65536 quad:98304 quad:131072 quad:163840 quad:196608 coupled:65536 coupled:131072 coupled:262144 decoupled:32768 decoupled:65536 decoupled:131072 decoupled:262144" > "$OUT/ab_helper_wave.txt" 2>&1
python3 tools/evidence.py > "$OUT/runtime_ab.json" 2> "$OUT/runtime_ab.err"
for ar in 1 0; do QR_LIB=$PWD/$EV/libquadrotor_hip_q_stamps.so python3 tools/stamp_timeline.py --auto-reset $ar --json "$OUT/stamps_quad65536_ar$ar.json" > "$OUT/stamps_quad65536_ar$ar.txt" 2>&1; done
bash tools/rocprof_floor.sh > "$OUT/rocprof_dispatch_floor.txt" 2>&1
$EV/valu_mb > "$OUT/valu_microbench.json" 2> "$OUT/valu_microbench.err"
$EV/vmem_mb > "$OUT/vmem_width_microbench.json" 2> "$OUT/vmem_width_microbench.err"
$EV/first_load_mb > "$OUT/first_load_microbench.json" 2> "$OUT/first_load_microbench.err"
timeout 120 $EV/resident_mb > "$OUT/resident_pacing_microbench.json" 2> "$OUT/resident_pacing_microbench.err"
python3 tools/ppo_rollout_bench.py > "$OUT/ppo_rollout.json" 2> "$OUT/ppo_rollout.err"
python3 tools/autotune_table.py > "$OUT/autotune_table.txt" 2> "$OUT/autotune_table.err"
bash tools/resource_usage.sh > "$OUT/resource_usage.txt" 2>/dev/null
fi
if [ "$WHAT" = all ] || [ "$WHAT" = tests ]; then
# (4) the parity figures the GPU tests print, and the soak run of the final build
python3 -m pytest tests -q -m gpu -s 2>&1 | grep -E "[0-9]e-[0-9]|passed|failed" | cut -c1-400 > "$OUT/parity_summary.txt"
python3 tools/soak.py 100000 > "$OUT/soak.json" 2> "$OUT/soak.err"
python3 tools/make_digest.py "$OUT/digest_gfx950.json" > /dev/null 2> "$OUT/digest.err"
cp gpurun_out/launch_stats_*.json "$OUT/" 2>/dev/null
fi
ls "$OUT"
