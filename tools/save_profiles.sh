# Copies the summaries tools/gpu_round3.sh left under gpurun_out/ into profiles/ under the given prefix (default r03).
set -e
R=${1:-r03}
cp gpurun_out/bench.json profiles/${R}_bench.json
cp gpurun_out/bench_stats.csv profiles/${R}_bench_kernel_stats.csv
cp gpurun_out/stereo_stats_100000.csv profiles/${R}_stereo_kernel_stats_100k.csv
cp gpurun_out/stereo_stats_10000.csv profiles/${R}_stereo_kernel_stats_10k.csv
cp gpurun_out/stereo_gaps_100000.txt profiles/${R}_stereo_gaps_100k.txt
cp gpurun_out/stereo_gaps_10000.txt profiles/${R}_stereo_gaps_10k.txt
cp gpurun_out/bench_pmc.json profiles/${R}_bench_pmc.json
cp gpurun_out/pmc_stereo.json profiles/${R}_stereo_pmc.json
python3 - "$R" <<'PY'
import json, sys
R = sys.argv[1]
json.dump({"this build": json.load(open("gpurun_out/sq_r03.json")), "build of commit a4f74dc": json.load(open("gpurun_out/sq_r02.json")),
           "note": "rocprofv3 --pmc over tools/kernel_bench.py 100000 20 (the mono fused Jacobian kernel launched back to back), "
                   "one pass each; per-kernel means over the launches"}, open("profiles/%s_sq_active_before_after.json" % R, "w"), indent=1)
PY
cp gpurun_out/wholecall.json profiles/${R}_wholecall.json
grep -v "^$" gpurun_out/wholecall_phases.txt | tail -24 > profiles/${R}_wholecall_phases.txt
ls -la profiles/${R}_*
