# round-5 PMC passes + the matrix-core probe alone (the kernel trace and the bench lines come from tools/profile_r05.sh); run through gpurun from the repo root
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05prof; mkdir -p $O
export TMPDIR=/tmp
( while sleep 45; do date >> $O/heartbeat.txt; done ) & HB=$!
PYTHONPATH=. python tools/mfma_peak_probe.py > $O/mfma_peak_probe.txt 2>&1; tail -4 $O/mfma_peak_probe.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-timing --no-extras --no-check > /dev/null 2> $O/pmc_fetch.err && echo fetch ok
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-timing --no-extras --no-check > /dev/null 2> $O/pmc_write.err && echo write ok
python tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "python3 bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-timing --no-extras --no-check"
rm -rf $O/pmc_fetch $O/pmc_write
kill $HB; rm -f $O/heartbeat.txt
ls $O
