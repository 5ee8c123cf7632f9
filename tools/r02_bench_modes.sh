#!/bin/bash
# The non-default bench configurations on one GPU, and a two-rank rehearsal (gloo, both ranks on GPU 0).
set -o pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/modes"; mkdir -p "$O"
cd "$R" || exit 1
for c in c3 c4 c5; do
  timeout -k 10 400 python bench.py --config $c > "$O/$c.json" 2> "$O/$c.err" || { echo "$c failed"; tail -5 "$O/$c.err"; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$O/$c.json')); print('$c', d['metric'], d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['config'].get('timed_regions'))"
done
export JEA_DIST_BACKEND=gloo
for c in c2 c3 c4 c5; do
  extra=""; [ $c = c2 ] && extra="--quick --steps 200 --warmup 20"; [ $c = c3 -o $c = c5 ] && extra="--frames 128 --steps 3 --warmup 1"; [ $c = c4 ] && extra="--c4-size 4096 --steps 3 --warmup 1"
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --config $c $extra > "$O/${c}_2ranks.json" 2> "$O/${c}_2ranks.err" || { echo "$c 2 ranks failed"; tail -8 "$O/${c}_2ranks.err"; exit 1; }
  python3 -c "import json,sys; d=json.loads([l for l in open('$O/${c}_2ranks.json') if l.startswith('{')][-1]); print('$c x2', d['value'], d['ms_per_step'], d['n_gpus'])"
done
