set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/t.log
for cfg in "" "CCALIB_NO_CHAIN=1" "CCALIB_NO_FOLD=1" "CCALIB_NO_CHAIN=1 CCALIB_NO_FOLD=1"; do
env $cfg timeout -k 10 300 python bench.py --no-cpu > gpurun_out/bench_x.json 2> gpurun_out/bench.err
python3 -c "import json;d=json.load(open('gpurun_out/bench_x.json'));print('$cfg'.ljust(36),round(d['ms_per_step'],4),'%.3e'%d['value'],d['roofline']['launch_ms'],d['jac_passes'],d['err_passes'])"
done
