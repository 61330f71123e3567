#!/bin/bash
# round-4 GPU session K: the full GPU suite
set -u
O=gpurun_out; mkdir -p $O
(timeout 2400 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -30) > $O/r4k_tests.log
tail -5 $O/r4k_tests.log
