cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --no-cpu-baseline --no-kernels --steps 300 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])"; }
for i in 1 2; do
  echo "default $(run X=1)"
  echo "COMPACT_MIN_S=17 $(run VLP3D_SA_COMPACT_MIN_S=17)"
  echo "COMPACT_MIN_S=33 $(run VLP3D_SA_COMPACT_MIN_S=33)"
  echo "CHAIN_TILE_ROWS=64 $(run VLP3D_CHAIN_TILE_ROWS=64)"
  echo "RELBIAS_BLOCKS=512 $(run VLP3D_RELBIAS_BLOCKS=512)"
  echo "RELBIAS_BLOCKS=128 $(run VLP3D_RELBIAS_BLOCKS=128)"
  echo "DEFERRED_LAST=0 $(run VLP3D_DEFERRED_LAST=0)"
  echo "LINEAR_TILE_ROWS=64 $(run VLP3D_LINEAR_TILE_ROWS=64)"
done
