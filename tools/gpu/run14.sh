set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MVS_CONV_WINO=4 python tests/variant_check.py 2>&1 | grep -v amdgpu.ids | tail -2
MVS_CONV_WINO=4 python tests/layer_check.py 16 16 32 f32 2>&1 | grep -v amdgpu.ids | tail -2
MVS_CONV_WINO=4 python tests/layer_check.py 24 24 40 f32 2>&1 | grep -v amdgpu.ids | tail -2
for wz in 2 4 2 4; do echo "CONV_WINO=$wz"; MVS_CONV_WINO=$wz python bench.py --streams 1 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k:v['ms'] for k,v in d['stages'].items() if k in ('conv2','conv4')})"; done
