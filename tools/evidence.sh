# Round evidence: PMC traffic of the roofline kernels, bench JSON lines, rocprofv3 kernel stats + one steady-state step.
# Usage (on the GPU box): bash tools/evidence.sh r04_e
set -e
tag=$1
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
for m in self cross; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$m -- python3 $GRAFT_REPO_ROOT/tools/roofline_kernels.py $m > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$m -- python3 $GRAFT_REPO_ROOT/tools/roofline_kernels.py $m > /dev/null 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pmc_f_self /tmp/pmc_w_self $out/${tag}_pmc_traffic.json /tmp/pmc_f_cross /tmp/pmc_w_cross > $out/${tag}_pmc_traffic.txt
cp /tmp/pmc_f_self/*/*counter_collection.csv $out/${tag}_pmc_fetch_size_counter_collection.csv
cp /tmp/pmc_w_self/*/*counter_collection.csv $out/${tag}_pmc_write_size_counter_collection.csv
cp $out/${tag}_pmc_traffic.json $GRAFT_REPO_ROOT/profiles/r04_pmc_traffic.json   # bench.py reads the table of THIS run
cd $GRAFT_REPO_ROOT
python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
cd /tmp
# (1) the default command (what the driver runs, minus the host-side CPU baseline): per-kernel statistics
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o g -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1
cp /tmp/prof_g/g_kernel_stats.csv $out/${tag}_bench_default_kernel_stats.csv
# (2) the timed steps alone (no per-kernel section, no other configurations): statistics + one steady-state step
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o h -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernels > /dev/null 2>&1
cp /tmp/prof_h/h_kernel_stats.csv $out/${tag}_bench_steps_only_kernel_stats.csv
mkdir -p /tmp/pg/x && cp /tmp/prof_h/h_kernel_trace.csv /tmp/pg/x/
python3 $GRAFT_REPO_ROOT/tools/trace_last_step.py /tmp/pg 400 $out/${tag}_step_launches.txt > $out/${tag}_steady_state_step.txt
head -3 $out/${tag}_steady_state_step.txt
