# round-2 profiles: run on the GPU box through gpurun from the repo root; results land under gpurun_out/r02prof/
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02prof; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench.json 2> $O/bench.err && tail -c 300 $O/bench.json && echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.err && echo trace ok
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-timing --schedule batch > /dev/null 2> $O/pmc_fetch.err && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-timing --schedule batch > /dev/null 2> $O/pmc_write.err && echo write ok
python tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-timing --schedule batch"
python bench.py --workload fem > $O/fem.json 2> $O/fem.err && tail -c 300 $O/fem.json && echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fem_trace -- python3 bench.py --workload fem --no-cpu-baseline > $O/fem_under_rocprof.json 2> $O/fem_trace.err && echo fem trace ok
timeout -k 10 300 python tools/bench_slfmm_box.py 1.0 0.05 > $O/slfmm_box.json 2> $O/slfmm.err; tail -c 600 $O/slfmm_box.json; echo
timeout -k 10 300 python tools/bench_amg_fem.py 96 4 > $O/amg_fem.json 2> $O/amg.err; tail -c 900 $O/amg_fem.json; echo
find $O -name "*kernel_stats.csv" | head; du -sh $O
# keep only the summaries (the raw traces are large)
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
