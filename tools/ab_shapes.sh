# Same-box A/B per launch shape: bench.py's roofline leg (every distinct launch shape replayed alone) with the previous and the new library.
export PYTHONPATH=$PWD
OUT=${1:-gpurun_out/abs}; mkdir -p $OUT; export ABS_OUT=$OUT
for lib in prev new; do
  if [ $lib = prev ]; then export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_prev.so; else unset FLAIR_HIP_LIB; fi
  timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>>$OUT/err.log | tail -1 > $OUT/$lib.json
done
python - <<'PY'
import json
a=json.load(open(''+__import__("os").environ.get("ABS_OUT","gpurun_out/abs")+'/prev.json')); b=json.load(open(''+__import__("os").environ.get("ABS_OUT","gpurun_out/abs")+'/new.json'))
print('ms/step', a['ms_per_step'], b['ms_per_step'])
def fam(j):
    out={}
    for f in j['roofline']['families']:
        out[f.get('family', f.get('name'))]=f
    return out
fa,fb=fam(a),fam(b)
for k in fa:
    if k in fb:
        x,y=fa[k],fb[k]
        print(k, {kk:(round(x[kk],2),round(y[kk],2)) for kk in x if isinstance(x[kk],(int,float)) and kk in y and ('ms' in kk or 'us' in kk)})
def key(s): return json.dumps(s['shape'],sort_keys=True)
sa={key(s):s for s in a['roofline']['by_shape']}; sb={key(s):s for s in b['roofline']['by_shape']}
for k in sa:
    if k in sb:
        print(k, sa[k]['launches'], round(sa[k]['us_per_launch'],1), round(sb[k]['us_per_launch'],1))
PY
