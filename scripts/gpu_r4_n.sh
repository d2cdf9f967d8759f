#!/bin/bash
mkdir -p gpurun_out
echo "== single-workgroup kernel (default)"; timeout -k 10 200 python scripts/time_smallbatch.py 200x2 200x12 200x54 200x108 128x2 128x54 256x2 256x54 2>&1 | grep "^n="
echo "== windowed path from n = 97 (NEGF_GJ_LARGE_MIN=97)"; NEGF_GJ_LARGE_MIN=97 timeout -k 10 200 python scripts/time_smallbatch.py 200x2 200x12 200x54 200x108 128x2 128x54 256x2 256x54 2>&1 | grep "^n="
