#!/bin/bash
cd /root/repo
bash tools/profile_round.sh r02 > gpurun_out/profile_round.log 2>&1
bash tools/measure_extras.sh r02 > gpurun_out/extras.log 2>&1
tail -5 gpurun_out/profile_round.log; tail -30 gpurun_out/extras.log
