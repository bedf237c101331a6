cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02 2>&1 | tail -12
bash tools/measure_extras.sh r02 2>&1 | tail -16
