bash tools/profile_round.sh r02 C4 > gpurun_out/r02_profile_C4.log 2>&1; tail -30 gpurun_out/r02_profile_C4.log
