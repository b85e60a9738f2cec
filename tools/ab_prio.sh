cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
run() { env $1 python bench.py --no-cpu-baseline --no-kernels --steps 300 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for i in 1 2; do
  echo "default $(run X=1)"
  echo "SIDE_PRIORITY=1 $(run VLP3D_SIDE_PRIORITY=1)"
  echo "SIDE_PRIORITY=-1 $(run VLP3D_SIDE_PRIORITY=-1)"
done
