#!/bin/bash
# Rehearsal of the N > 1 bench path on ONE GPU (gloo group, every rank on cuda:0): the peer-store child group alone, then the whole
# two-rank line with the child group started by rank 0.  Not a measurement: the ranks share one GPU and gloo stages through the host.
O=gpurun_out/r4p; mkdir -p $O
export FP8MI_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=3 --master-addr 127.0.0.1 --master-port 29611 bench.py --peer-child --gpus 3 --steps 3 --warmup 1 > $O/child.json 2> $O/child.err
echo "child rc=$?"; tail -c 3000 $O/child.json
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 3 --warmup 1 > $O/line.json 2> $O/line.err
echo "line rc=$?"; python - <<'PY'
import json
for ln in open("gpurun_out/r4p/line.json"):
    if ln.startswith('{"metric"'):
        d = json.loads(ln)
        print("value", d["value"], d["unit"], "n_gpus", d["n_gpus"], "ms_per_step", d["ms_per_step"])
        print("peer_allgather", json.dumps(d.get("peer_allgather"))[:3000])
PY
tail -5 $O/line.err
# the headline step itself on the peer-store gather (on a real node bench.py chooses it when the child group measured it faster; here it is forced)
FP8MI_BENCH_GATHER=peer timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29615 bench.py --gpus 2 --steps 3 --warmup 1 --no-peer-store > $O/line_peer.json 2> $O/line_peer.err
echo "line (peer headline) rc=$?"; python - <<'PY'
import json
for ln in open("gpurun_out/r4p/line_peer.json"):
    if ln.startswith('{"metric"'):
        d = json.loads(ln)
        print("value", d["value"], d["unit"], "n_gpus", d["n_gpus"], "ms_per_step", d["ms_per_step"], "hip_graph", d["config"]["hip_graph"])
        print(d["config"]["parallelism"]); print("allgather_only", d.get("allgather_only"), "status", d.get("peer_timeout_status"))
PY
tail -3 $O/line_peer.err
