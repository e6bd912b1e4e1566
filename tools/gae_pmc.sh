#!/bin/bash
# HBM traffic of the GAE kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), summarised into
# gpurun_out/prof/<tag>_gae_pmc.csv (MI355X_MICROARCH.md, HBM / rocprofv3 section)
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 $R/tools/gae_pmc_driver.py > /tmp/pmc_f.log 2>&1; echo "fetch pass rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 $R/tools/gae_pmc_driver.py > /tmp/pmc_w.log 2>&1; echo "write pass rc=$?"
( python3 $R/tools/summarize_pmc.py /tmp/pmc_f; python3 $R/tools/summarize_pmc.py /tmp/pmc_w ) > $R/gpurun_out/prof/${TAG}_gae_pmc.csv
cat $R/gpurun_out/prof/${TAG}_gae_pmc.csv
