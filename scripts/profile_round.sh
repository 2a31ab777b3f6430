#!/bin/bash
# Profiles the headline frame of bench.py at 3840x2160 and 1920x1080 (strict kernel) and at 3840x2160 with the tolerance
# kernel: kernel trace + counter passes each (scripts/profile_bench.sh), under gpurun_out/prof_<tag>_*.
set -eo pipefail
TAG=${1:-r02}
bash $GRAFT_REPO_ROOT/scripts/profile_bench.sh ${TAG}_4k --width 3840 --height 2160
bash $GRAFT_REPO_ROOT/scripts/profile_bench.sh ${TAG}_1080p --width 1920 --height 1080
bash $GRAFT_REPO_ROOT/scripts/profile_bench.sh ${TAG}_4k_fast --width 3840 --height 2160 --kernel 4
