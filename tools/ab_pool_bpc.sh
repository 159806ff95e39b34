#!/bin/bash
# lab: register budget of downscale_pool_kernel (launch bounds: blocks per CU), same box, alternating builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for v in 4 3 2; do
    PD_POOL_BPC=$v python -m physdock_amd.build pool.hip > /dev/null 2>&1
    echo "== PD_POOL_BPC=$v"
    python tools/pool_bench.py 2>&1 | grep "^downscale"
  done
done
python -m physdock_amd.build pool.hip > /dev/null 2>&1
