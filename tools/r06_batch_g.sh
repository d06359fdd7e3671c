#!/bin/bash
# round-6 batch G (final sources): profiles, then -- in a second call, once they are copied into profiles/ -- the bench line and the suite
bash tools/profile_round.sh r06g > /dev/null 2>&1
ls gpurun_out/prof_r06g | head -20
