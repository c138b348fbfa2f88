cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03a > gpurun_out/r03a_profile_round.log 2>&1
tail -30 gpurun_out/r03a_profile_round.log
