cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
tail -60 gpurun_out/collect_r06.log
