cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03z > gpurun_out/r03z_profile_round.log 2>&1
tail -30 gpurun_out/r03z_profile_round.log
