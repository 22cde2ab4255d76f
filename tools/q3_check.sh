# Q3 on one GPU: the pipeline's tests, then per-kernel timelines and the bench under the switches given as arguments
# (each argument an environment assignment or "-" for none), all on one box
mkdir -p gpurun_out/r03
timeout -k 10 700 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 -k "q3 or join_groupby or join_pipeline" > gpurun_out/r03/q3_tests.log 2>&1; tail -3 gpurun_out/r03/q3_tests.log
ROOT=$GRAFT_REPO_ROOT
for sw in "$@"; do
  [ "$sw" = "-" ] && sw="LLKV_NONE=1"
  echo "== $sw"
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_q3db
  env $sw python3 -c "pass"  # (rocprofv3 must start the interpreter itself: the switch goes through the environment of this shell)
  export $sw
  timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_q3db -o q3 -- python3 $ROOT/tools/q3_bench.py sf10 > $ROOT/gpurun_out/r03/q3_tl.log 2>&1
  db="$(find /tmp/prof_q3db -name "*_results.db" | head -1)"; python3 $ROOT/tools/rocprof_timeline.py "$db" hj_fill_zero_ranges_kernel | tee $ROOT/gpurun_out/r03/q3_timeline_${sw%%=*}.txt
  cd $ROOT && python tools/q3_bench.py sf10 2>/dev/null | cut -c1-330
  unset ${sw%%=*}
done
