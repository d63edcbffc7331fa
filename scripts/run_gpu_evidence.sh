# Round evidence at the current sources, written under gpurun_out/ev/ (copy into profiles/ afterwards):
# rocprofv3 stats + PMC of the bench in its three moist modes, the other configurations, the L2-side fetch counters,
# the level / grid-size sweeps.    usage: run_gpu_evidence.sh <round tag, e.g. r02_f>
tag=${1:-rXX}
mkdir -p gpurun_out/ev
for m in family exact table; do
  name=${tag}_$([ $m = exact ] && echo rk4 || echo $m)
  bash scripts/run_gpu_prof.sh $name --moist $m > gpurun_out/ev/${name}_prof.log 2>&1 || exit 1
  cp gpurun_out/${name}_stats.txt gpurun_out/${name}_pmc.json gpurun_out/ev/ && rm -rf gpurun_out/prof_$name gpurun_out/pmc_$name
  echo "done $name"
done
python3 scripts/run_gpu_configs.py > gpurun_out/ev/${tag}_configs.jsonl 2> gpurun_out/ev/${tag}_configs.err && echo "done configs"
bash scripts/run_gpu_fetch.sh ${tag}_c5 c5 family > /dev/null 2>&1 && cp gpurun_out/${tag}_c5_fetch.txt gpurun_out/ev/ && echo "done fetch c5"
bash scripts/run_gpu_fetch.sh ${tag}_c2 c2 family > /dev/null 2>&1 && cp gpurun_out/${tag}_c2_fetch.txt gpurun_out/ev/ && echo "done fetch c2"
python3 scripts/run_gpu_levels.py 2 4 8 16 24 32 48 64 96 128 > gpurun_out/ev/${tag}_levels.json 2>/dev/null && echo "done levels"
python3 scripts/run_gpu_persist.py > gpurun_out/ev/${tag}_persist.txt 2>/dev/null && echo "done persist"
python3 bench.py --steps 30 --warmup 5 --no-cpu --no-table-leg --no-config-legs --data smooth 2>/dev/null | tail -1 > gpurun_out/ev/${tag}_smooth_bench.json && echo "done smooth"
