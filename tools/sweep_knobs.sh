run() { echo -n "$* : "; env "$@" python bench.py --no-kernels --no-cpu-baseline --steps 80 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
run A=1
run VLP3D_RELBIAS_BLOCKS=512
run VLP3D_RELBIAS_BLOCKS=128
run VLP3D_DEFERRED_LAST=0
run VLP3D_DEFERRED_LAST=0 VLP3D_RELBIAS_BLOCKS=512
run VLP3D_WGRAD_SLAB_MB=32
run VLP3D_WGRAD_SLAB_MB=8
run VLP3D_WGRAD_BLOCKS=1024
run VLP3D_WGRAD_TILES=2
run VLP3D_WGRAD_TILES=8
run VLP3D_ROWS_WGRAD_BLOCKS=128
run VLP3D_ROWS_WGRAD_BLOCKS=32
run A=2
