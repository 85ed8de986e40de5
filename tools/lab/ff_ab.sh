A="--targets 512 --steps 2 --warmup 1 --e2e-targets 0 --frames-targets 0 --psf-targets 0 --linpsf-drift 0 --frame 0 --cpu-sample 0 --fullframe-frames 16"
python bench.py $A > gpurun_out/b_ff.json 2> gpurun_out/b_ff.err; tail -2 gpurun_out/b_ff.err
TP_MEDIAN_PLAIN=1 python bench.py $A > gpurun_out/b_ff0.json 2> gpurun_out/b_ff0.err
python - <<'PY'
import json
for f in ('b_ff','b_ff0'):
    r=json.load(open(f'gpurun_out/{f}.json'))['fit_background_frames']
    print(f, {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if 'ms' in kk or 'per' in kk}) for k,v in r.items() if k!='what'})
PY
