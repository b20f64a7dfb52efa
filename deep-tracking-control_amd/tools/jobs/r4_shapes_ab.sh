for im in 0 1; do DTC_IMAGES=$im DTC_PROF_SHAPES=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null > gpurun_out/shapes_im$im.json; done
python - <<'PY'
import json
a=json.load(open('gpurun_out/shapes_im0.json'))['kernel_classes']; b=json.load(open('gpurun_out/shapes_im1.json'))['kernel_classes']
keys=sorted(set(a)|set(b), key=lambda k:-(a.get(k,{}).get('ms',0)+b.get(k,{}).get('ms',0)))
for k in keys[:40]:
    x,y=a.get(k,{}),b.get(k,{})
    print(f"{k:34s} off {x.get('ms',0):7.3f} ms {x.get('launches',0):4d} x {1e3*x.get('ms',0)/max(1,x.get('launches',0)):6.1f} us | on {y.get('ms',0):7.3f} ms {y.get('launches',0):4d} x {1e3*y.get('ms',0)/max(1,y.get('launches',0)):6.1f} us")
PY
