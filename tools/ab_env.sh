# A/B of environment switches on ONE box: bash tools/ab_env.sh "VAR1=0 VAR2=0" [rounds]  -> ms_per_step of `python bench.py
# --no-cpu-baseline --no-kernels --steps 300` with and without the assignments, alternating (boxes differ by ~1.5 %).
set -e
cd $GRAFT_REPO_ROOT
rounds=${2:-3}
for i in $(seq 1 $rounds); do
  a=$(env $1 python bench.py --no-cpu-baseline --no-kernels --steps 300 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  b=$(python bench.py --no-cpu-baseline --no-kernels --steps 300 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "round $i: [$1] $a ms   [default] $b ms"
done
