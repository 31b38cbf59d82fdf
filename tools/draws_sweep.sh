#!/bin/bash
# k_mega: draws per wave the grab size is chosen for (one rank's share of the bench frame for N = 1, 2, 4, 8, alone on one GPU)
cd $GRAFT_REPO_ROOT
for d in 4 8 16 32; do echo "== draws per wave: $d"; LJ_TUNE_MEGA_DRAWS=$d timeout -k 10 100 python3 tools/shard_time.py 2>/dev/null; done
