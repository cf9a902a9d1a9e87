# rehearsal of bench.py's multi-rank path on a one-GPU box: two ranks share the GPU, sums go through gloo
set -e
CCALIB_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 10 --warmup 2 --views 50000 2>&1 | tail -5
