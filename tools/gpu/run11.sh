set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for args in "--streams 2" "--streams 2 --cu-split" "--streams 4 --cu-split" "--streams 2" "--streams 2 --cu-split" "--streams 3 --cu-split"; do
echo "== $args"; python bench.py $args --no-cpu-baseline --no-e2e --staged-steps 0 --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
