# HBM traffic of the dominant kernel's launch set (tools/pmc_dominant.py) with the two tile orders of the temporal convolutions: bash tools/pmc_tfast.sh <bench line json> <out txt>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
export PYTHONPATH=$R
LINE=$R/$1; OUT=$R/$2
: > $OUT
for v in 0 1; do
  export FLAIR_DMA_TFAST=$v
  rm -rf /tmp/pf_$v /tmp/pw_$v
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$v -o b --output-format csv -- python $R/tools/pmc_dominant.py $LINE > /dev/null 2>&1
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$v -o b --output-format csv -- python $R/tools/pmc_dominant.py $LINE > /dev/null 2>&1
  echo "## FLAIR_DMA_TFAST=$v" >> $OUT
  python $R/tools/pmc_traffic.py /tmp/pf_$v /tmp/pw_$v /tmp/t_$v.json | grep -v "elementwise" >> $OUT
done
cat $OUT
