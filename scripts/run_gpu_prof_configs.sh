# rocprofv3 kernel-trace summaries of the c3 and c5 configuration scripts (evidence for DESIGN.md section 7)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in c3 c5; do
  rm -rf gpurun_out/prof_$c
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$c -o $c -- python3 scripts/run_gpu_$c.py > gpurun_out/prof_$c.log 2>&1; echo "$c rc=$?"
  tail -1 gpurun_out/prof_$c.log | cut -c1-300
done
