set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r04; mkdir -p $OUT
python3 -m pytest tests -q -m gpu -x -k "helper_wave or auto_reset or final_observation or rollout_storage or launch_rule or headline or production_mode or fuzz or reset_on_done or set_state or legacy or in_launch" 2>&1 | tail -5 > $OUT/run3_tests.txt
for k in coupled decoupled; do QR_AB_KIND=$k python3 tools/ab_equal.py build/ab/all_noobs.so build/ab/all_obs.so; done > $OUT/ab_help_obs_equal.txt 2>&1
QR_AB_KINDS=coupled,decoupled QR_AB_SIZES=32768,65536,131072 QR_AB_AR=1 QR_AB_REPS=3 python3 tools/ab_libs.py build/ab/all_noobs.so build/ab/all_obs.so > $OUT/ab_help_obs.txt 2>&1
cat $OUT/run3_tests.txt $OUT/ab_help_obs_equal.txt $OUT/ab_help_obs.txt
