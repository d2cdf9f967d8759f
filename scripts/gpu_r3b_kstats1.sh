#!/bin/bash
# per-kernel durations of the windowed inverse with ONE stream group (kernels one after the other)
export NEGF_GJ_SPLIT_MAX=0
SIZES="${SIZES:-500}" bash scripts/gpu_kstats.sh
