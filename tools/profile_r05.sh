# round-5 profiles: run on the GPU box through gpurun from the repo root; results land under gpurun_out/r05prof/
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05prof; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
# a counter pass writes nothing for minutes: a heartbeat under gpurun_out/ keeps the run from being taken for hung
( while sleep 45; do date >> $O/heartbeat.txt; done ) & HB=$!
python bench.py > $O/bench.json 2> $O/bench.err && tail -c 300 $O/bench.json && echo
python bench.py --steps 20 > $O/bench_20_steps.json 2> $O/bench20.err && echo 20 ok
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/trace.err && echo trace ok
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-timing --no-extras --no-check > /dev/null 2> $O/pmc_fetch.err && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-timing --no-extras --no-check > /dev/null 2> $O/pmc_write.err && echo write ok
python tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "python3 bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-timing --no-extras --no-check"
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cut -c1-260 $f > $O/bench_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
PYTHONPATH=. python tools/mfma_peak_probe.py > $O/mfma_peak_probe.txt 2>&1
du -sh $O; ls $O
kill $HB; rm -f $O/heartbeat.txt
