cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( time timeout 900 python bench.py --no-cpu-baseline > $O/r2_call49_bench.json 2> $O/r2_call49_bench.err ) 2>&1 | grep real
python - <<PY
import json
d=json.loads(open('$O/r2_call49_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
e=d['extra']
for k,v in e.items():
    if k.startswith('images_per_s') or k.startswith('vqgan_encode_decode_images') or k.startswith('taming') or k.startswith('config4'): print(k, v)
PY
