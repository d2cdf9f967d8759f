#!/bin/bash
# end of round: the fuzzers on the final build, fresh seeds
mkdir -p gpurun_out
timeout -k 10 280 python scripts/fuzz_products.py 41 2>&1 | tail -3 || exit 1
timeout -k 10 280 python scripts/fuzz_inverse.py 42 2>&1 | tail -3 || exit 1
timeout -k 10 200 python scripts/fuzz_small.py 43 60 2>&1 | tail -2 || exit 1
timeout -k 10 200 python scripts/fuzz_chain_rr.py 44 60 2>&1 | tail -2 || exit 1
