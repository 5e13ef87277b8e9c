#!/bin/bash
# Rehearse the driver's multi-GPU launch line on the 1-GPU box: torch.distributed.run + RCCL process group with 1 rank,
# then 2 ranks over gloo on the CPU control plane is covered by tests/test_dp_gloo.py.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python - <<'PY'
import torch, torch.distributed as dist, os
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier(device_ids=[0]); print("nccl 1-rank ok", t.tolist())
dist.destroy_process_group()
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -2 | cut -c1-300
