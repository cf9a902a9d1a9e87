set -e
timeout -k 10 400 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
