#!/bin/bash
# Experiment: the mixed lanes=4 batch while another process keeps the machine busy (is the slow tail a clock effect?)
cd $GRAFT_REPO_ROOT
python -c "
import torch, time
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16); b = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
torch.cuda.synchronize(); print('busy start', flush=True)
t0 = time.time()
while time.time() - t0 < 40:
    for _ in range(20): c = a @ b
    torch.cuda.synchronize()
print('busy end', flush=True)
" &
BUSY=$!
sleep 15
timeout -k 10 200 python tools/exp_replicas.py 800 4 4,1
kill $BUSY 2>/dev/null; wait $BUSY 2>/dev/null
echo "--- alone"
timeout -k 10 200 python tools/exp_replicas.py 800 4 4,1
