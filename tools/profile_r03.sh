# round-3 profiles: run on the GPU box through gpurun from the repo root; results land under gpurun_out/r03prof/
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03prof; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench.json 2> $O/bench.err && tail -c 300 $O/bench.json && echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/trace.err && echo trace ok
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 0 --schedule pipeline --no-cpu-baseline --no-timing --no-extras > /dev/null 2> $O/pmc_fetch.err && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 0 --schedule pipeline --no-cpu-baseline --no-timing --no-extras > /dev/null 2> $O/pmc_write.err && echo write ok
python tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "python3 bench.py --steps 6 --warmup 0 --schedule pipeline --no-cpu-baseline --no-timing --no-extras"
python bench.py --workload fem > $O/fem.json 2> $O/fem.err && tail -c 300 $O/fem.json && echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fem_trace -- python3 bench.py --workload fem --no-cpu-baseline > $O/fem_under_rocprof.json 2> $O/fem_trace.err && echo fem trace ok
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cut -c1-260 $f > $O/bench_kernel_stats.csv
f=$(find $O/fem_trace -name "*kernel_stats.csv" | head -1); cut -c1-260 $f > $O/fem_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
rm -rf $O/trace $O/fem_trace $O/pmc_fetch $O/pmc_write
du -sh $O; ls $O
