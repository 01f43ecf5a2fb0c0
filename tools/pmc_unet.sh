#!/bin/bash
# PMC traffic of one UNet forward (tools/time_unet.py):  bash tools/pmc_unet.sh
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf /tmp/pw
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o b --output-format csv -- python $R/tools/time_unet.py --iters 1 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o b --output-format csv -- python $R/tools/time_unet.py --iters 1 > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw
timeout -k 10 200 python $R/tools/time_unet.py 2>&1 | tail -2
