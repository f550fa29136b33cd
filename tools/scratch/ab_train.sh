#!/bin/bash
# A/B of variant builds on the training step: for every suffix runs tools/bench_train.py with TACO_LIB=csrc/libtaco_hip<suffix>.so.   bash tools/scratch/ab_train.sh "" _db3
for v in "$@"; do
  TACO_LIB=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip$v.so python tools/bench_train.py --steps 12 --warmup 3 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v] %.4f ms per step; phases %s' % (d['ms_per_step'], d['phase_ms']))"
done
