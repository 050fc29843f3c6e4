#!/bin/bash
# rehearsal of the N=2 launch line on one GPU with the gloo backend (both ranks use cuda:0)
export SR_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --workload goldilocks_d65536_b16384 --batch 2048
