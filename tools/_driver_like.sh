#!/bin/bash
# what the driver runs at round end, on a fresh box, with wall times
cd /root/repo
t0=$SECONDS; echo "== pytest -m gpu -x -q"; timeout 1800 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2; echo "   $((SECONDS-t0)) s"
t0=$SECONDS; echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300; echo "   $((SECONDS-t0)) s"
t0=$SECONDS; echo "== bench (default)"; python bench.py 2>/tmp/bench.err | cut -c1-400; tail -1 /tmp/bench.err; echo "   $((SECONDS-t0)) s"
t0=$SECONDS; echo "== bench under torchrun, 1 rank"; python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 4 --warmup 1 2>/tmp/bench2.err | cut -c1-300; tail -1 /tmp/bench2.err; echo "   $((SECONDS-t0)) s"
