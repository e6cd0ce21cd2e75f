# round-5 profiles: run on the GPU box through gpurun from the repo root; results land under gpurun_out/r05prof/
# (the counter passes and the matrix-core probe are tools/profile_r05_pmc.sh: together they do not fit one 1200 s call)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05prof; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
# a counter pass writes nothing for minutes: a heartbeat under gpurun_out/ keeps the run from being taken for hung
( while sleep 45; do date >> $O/heartbeat.txt; done ) & HB=$!
python bench.py > $O/bench.json 2> $O/bench.err && tail -c 300 $O/bench.json && echo
python bench.py --steps 20 > $O/bench_20_steps.json 2> $O/bench20.err && echo 20 ok
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/trace.err && echo trace ok
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cut -c1-260 $f > $O/bench_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
rm -rf $O/trace
du -sh $O; ls $O
kill $HB; rm -f $O/heartbeat.txt
