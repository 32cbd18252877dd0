#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r02_tree_tests.log 2>&1; tail -4 gpurun_out/r02_tree_tests.log
timeout 300 python scripts/tree_repro.py 2>&1 | tail -6 | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
