bash tools/exp_step_power.sh 2>&1 | tail -5
