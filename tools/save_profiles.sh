# Copies the summaries tools/gpu_round2.sh left under gpurun_out/ into profiles/ under this round's names.
set -e
R=${1:-r02}
cp gpurun_out/bench.json profiles/${R}_bench.json
cp gpurun_out/bench_stats.csv profiles/${R}_bench_kernel_stats.csv
cp gpurun_out/stereo_stats.csv profiles/${R}_stereo_kernel_stats_100k.csv
cp gpurun_out/stereo10k_stats.csv profiles/${R}_stereo_kernel_stats_10k.csv
cp gpurun_out/stereo_gaps.txt profiles/${R}_stereo_gaps_100k.txt
cp gpurun_out/bench_pmc.json profiles/${R}_bench_pmc.json
[ -f gpurun_out/pmc_stereo.json ] && cp gpurun_out/pmc_stereo.json profiles/${R}_stereo_pmc.json
ls -la profiles/${R}_*
