# the fused tail with split bf16 operands (conv11_prob_split_kernel) against its fp32-MFMA form (MVS_TAIL_SPLIT=0)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "conv11 or fused or maps or optin" 2>&1 | tail -3 &&
for r in 1 2; do
python tools/time_stage.py tail 300
for a in $(ls $C | sed -n 's/libmvs_hip_ablate\([0-9]*\).so/\1/p'); do MVS_LIB_PATH=$C/libmvs_hip_ablate$a.so python tools/time_stage.py tail 300; done
MVS_TAIL_SPLIT=0 python tools/time_stage.py tail 300
done 2>&1 | grep -v amdgpu.ids
for v in 1 0; do
  MVS_TAIL_SPLIT=$v python bench.py --streams 1 --steps 10 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-live-traffic > /tmp/b$v.json 2>/dev/null &&
  python -c "
import json; d=json.load(open('/tmp/b$v.json')); print('MVS_TAIL_SPLIT=$v', d['value'], d['stages']['conv11_prob']['ms'], {k:(v['value'], v['stages_ms']['conv11_prob']) for k,v in d['other_configs'].items()})"
done
