# soak: whole path on two streams from two host threads, every map compared bit for bit with the single-stream map
cd $GRAFT_REPO_ROOT
for st in f32 bf16 f16; do
python tools/probes/path_two_streams.py 192 128 160 3 $st 5 2>&1 | grep -v amdgpu.ids
python tools/probes/path_two_streams.py 48 32 40 4 $st 3 2>&1 | grep -v amdgpu.ids | awk -v st=$st '{s+=$6} END {print "48x32x40", st, s, "of 320 mismatching"}'
python tools/probes/path_two_streams.py 96 64 80 3 $st 4 2>&1 | grep -v amdgpu.ids | awk -v st=$st '{s+=$6} END {print "96x64x80", st, s, "of 240 mismatching"}'
done
