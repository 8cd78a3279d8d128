# A/B of environment tunables inside ONE gpurun call: bash tools/ab.sh "A=0" "RVC_X=1" ...   (value, one clip alone, conv_x3s ms per clip of the roofline pass)
run() { env $@ timeout 300 python bench.py --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
o={k[0]:k[1] for k in r['others']}
print('$*', d['value'], d['config']['one_clip_alone_ms'], 'x3q', r['kernel_ms_per_clip'], r['frac'], 'x3s', o.get('conv_x3s'), 'all', r['all_conv_kernels_ms_per_clip'])"; }
for v in "$@"; do run $v; done
