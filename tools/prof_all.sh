# usage: tools/prof_all.sh r06   (the round's tag): every profile the round commits, into gpurun_out/ (copy to profiles/ by hand)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/pmc_instmix.sh ${TAG} > gpurun_out/instmix.log 2>&1; tail -3 gpurun_out/instmix.log
cp profiles/${TAG}_pmc_instmix.json gpurun_out/${TAG}_pmc_instmix.json
rm -rf gpurun_out/ks gpurun_out/ksl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --streams 1 --pipelined-steps 0 --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0 > gpurun_out/ks_bench.log 2>&1; tail -1 gpurun_out/ks_bench.log | cut -c1-200
find gpurun_out/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats_bench4096.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ksl -- python3 tools/latency.py > gpurun_out/ksl.log 2>&1; tail -3 gpurun_out/ksl.log
find gpurun_out/ksl -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_latency_kernel_stats.csv
rm -rf gpurun_out/ks gpurun_out/ksl gpurun_out/im_*
# (the traffic passes first: the bench lines below replay profiles/${TAG}*_pmc_traffic.json and say whether it belongs to this tree)
bash tools/pmc_traffic.sh ${TAG} > gpurun_out/traffic.log 2>&1; tail -4 gpurun_out/traffic.log
bash tools/pmc_traffic.sh ${TAG}_4k 3840x2160 256 > gpurun_out/traffic4k.log 2>&1; tail -4 gpurun_out/traffic4k.log
python3 bench.py > gpurun_out/${TAG}_bench_4096.json 2> gpurun_out/bench_err.log; tail -c 600 gpurun_out/${TAG}_bench_4096.json
python3 bench.py --size 3840x2160 --frames 1024 --chunk 1024 > gpurun_out/${TAG}_bench_4k_1024.json 2>> gpurun_out/bench_err.log; tail -c 300 gpurun_out/${TAG}_bench_4k_1024.json
# the reference's own frame size (test.bmp: 1920x1200) through the run-time-band build of the fused sweep (round 6)
python3 bench.py --size 1920x1200 --frames 2048 --chunk 2048 --host-frames 0 --pose-frames 0 --latency-calls 0 > gpurun_out/${TAG}_bench_1920x1200.json 2>> gpurun_out/bench_err.log; tail -c 300 gpurun_out/${TAG}_bench_1920x1200.json
rm -rf gpurun_out/ks4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks4 -- python3 bench.py --size 3840x2160 --frames 1024 --chunk 1024 --streams 1 --pipelined-steps 0 --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0 > gpurun_out/ks4_bench.log 2>&1; tail -1 gpurun_out/ks4_bench.log | cut -c1-200
find gpurun_out/ks4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats_bench4k_1024.csv
rm -rf gpurun_out/ks4
