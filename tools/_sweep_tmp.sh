cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1
mkdir -p /tmp/pt/x && cp /tmp/prof_t/t_kernel_trace.csv /tmp/pt/x/
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py /tmp/pt > gpurun_out/timeline_chain.txt 2>&1
grep -n "^== queue" gpurun_out/timeline_chain.txt
