#!/bin/bash
# Canary of the trial-finish ticket protocol (k_backsub_residual): the litmus test on a library whose ba.hip was compiled at -O1
# (a different instruction schedule around the agent-scope stores / s_waitcnt / ticket).  Run through gpurun from the repo root
# AFTER building the variant in the build container:  python tools/build_variant.py O1 ba.hip -O1   ->  ssvio_amd/libssx.so.O1
set -u
SSX_LIB=$PWD/ssvio_amd/libssx.so.O1 SSX_LITMUS_REPS=${1:-16} python -m pytest tests/test_ba_gpu.py -q -x -k "trial_finish_litmus or is_deterministic" 2>&1 | tail -5
