cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/pmc_instmix.sh r02 > gpurun_out/instmix.log 2>&1; tail -3 gpurun_out/instmix.log
cp profiles/r02_pmc_instmix.json gpurun_out/r02_pmc_instmix.json
rm -rf gpurun_out/ks gpurun_out/ksl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0 > gpurun_out/ks_bench.log 2>&1; tail -1 gpurun_out/ks_bench.log | cut -c1-200
find gpurun_out/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_kernel_stats_bench4096.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ksl -- python3 tools/latency.py > gpurun_out/ksl.log 2>&1; tail -3 gpurun_out/ksl.log
find gpurun_out/ksl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_latency_kernel_stats.csv
rm -rf gpurun_out/ks gpurun_out/ksl gpurun_out/im_*
python3 bench.py > gpurun_out/r02_bench_4096.json 2> gpurun_out/bench_err.log; tail -c 600 gpurun_out/r02_bench_4096.json
python3 bench.py --size 3840x2160 --frames 1024 --chunk 1024 > gpurun_out/r02_bench_4k_1024.json 2>> gpurun_out/bench_err.log; tail -c 300 gpurun_out/r02_bench_4k_1024.json
