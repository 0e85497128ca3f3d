#!/bin/bash
set -o pipefail
bash tools/profile_round.sh r04 C4 > gpurun_out/r04_prof_C4.log 2>&1; echo "C4 rc=$?"; tail -3 gpurun_out/r04_prof_C4.log
bash tools/profile_round.sh r04 C2 > gpurun_out/r04_prof_C2.log 2>&1; echo "C2 rc=$?"
bash tools/profile_round.sh r04 C4 u8 > gpurun_out/r04_prof_C4u8.log 2>&1; echo "C4u8 rc=$?"
bash tools/profile_methods.sh r04 > gpurun_out/r04_prof_methods.log 2>&1; echo "methods rc=$?"; tail -5 gpurun_out/r04_prof_methods.log
ls gpurun_out/profiles | grep r04
