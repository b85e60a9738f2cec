# step time under a few scheduling knobs (python bench.py --no-kernels): run on the GPU box, e.g. after a kernel got faster
run() { echo -n "$* : "; env "$@" python bench.py --no-kernels --no-cpu-baseline --steps 80 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
run A=1
run VLP3D_RELBIAS_BLOCKS=192
run VLP3D_RELBIAS_BLOCKS=384
run VLP3D_DEFERRED_LAST=0
run VLP3D_SA_LAST_DGRAD_MIN_ROWS=200000
run VLP3D_SA_LAST_WGRAD_MIN_ROWS=200000
run VLP3D_SA_LAST_WBLOCKS=256
run VLP3D_SA_LAST_WBLOCKS=768
run VLP3D_WGRAD_SLAB_MB=8
run VLP3D_WGRAD_SLAB_MB=24
run VLP3D_WGRAD_TILES=8
run VLP3D_WGRAD_BLOCKS=1024
run VLP3D_ROWS_WGRAD_BLOCKS=32
run VLP3D_LINEAR_TILE_ROWS=32
run VLP3D_SDPA_DKV_WPB=2
run A=2
