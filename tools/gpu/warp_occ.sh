# warp + variance: resident waves (launch bounds) and the packed 16-bit tap cache (MVS_WARP_PACKED=1), kernel + transpose
# times through the bench's per-kernel pass
cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/scene_3dreconstruction_mvsnet_amd/csrc
run() {  # label, lib, packed
  for cfg in cfg2 cfg5 cfg3; do
    MVS_LIB_PATH=$2 MVS_WARP_PACKED=$3 python bench.py --config $cfg --steps 10 --streams 1 --prewarm-ms 100 --no-cpu-baseline --no-e2e --no-other-configs --no-live-traffic > /tmp/w.json 2>/dev/null
    python -c "
import json; d=json.load(open('/tmp/w.json')); print('$1', '$cfg', 'warp', d['stages']['warp_variance']['ms'], 'value', d['value'])"
  done
}
run default $C/libmvs_hip.so 0
run packed $C/libmvs_hip.so 1
run lb4 $C/libmvs_hip_ablate61.so 0
run lb4+packed $C/libmvs_hip_ablate61.so 1
run lb5+packed $C/libmvs_hip_ablate62.so 1
# bit-identity of the packed cache against the default kernel (variance volume, cfg5-like small problem)
python - <<'PY'
import os, subprocess, sys
code = '''
import numpy as np, torch, sys, hashlib
sys.path.insert(0, ".")
from scene_3dreconstruction_mvsnet_amd import _lib, synthetic
for st in ("f16", "bf16"):
    dt = _lib.dtype_code(st)
    N, h, w, D = 5, 40, 56, 48
    f = torch.from_numpy(synthetic.random_features(N, 32, h, w, seed=3)).cuda()
    p = torch.from_numpy(synthetic.cameras(N, h, w, yaw_deg=1.0)).cuda()
    dv = torch.from_numpy(synthetic.depth_values(D)).cuda()
    ws = _lib.alloc_workspace(N, 32, D, h, w, "cuda:0", dt)
    v = _lib.warp_variance(f, _lib.relative_proj(p), dv, ws, dtype=dt)
    print(st, hashlib.sha1(v.cpu().view(torch.int16).numpy().tobytes()).hexdigest())
'''
outs = [subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MVS_WARP_PACKED=pk), capture_output=True, text=True).stdout for pk in ("0", "1")]
print("packed cache bit-identical:", outs[0] == outs[1] and len(outs[0]) > 10); print(outs[0])
PY
