export PYTHONPATH=$GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout -k 10 200 python -m pytest $R/tests/test_gpu_kernels.py -m gpu -q -x -k dcn 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o b --output-format csv -- python $R/tools/time_unet.py --iters 1 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o b --output-format csv -- python $R/tools/time_unet.py --iters 1 > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw | head -12
timeout -k 10 200 python $R/tools/time_unet.py 2>&1 | tail -2
