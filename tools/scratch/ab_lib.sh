#!/bin/bash
# A/B of variant builds of the library on one box, same call: for every suffix given (e.g. "" _fpd4 _fpd8) runs bench.py with
# TACO_LIB=csrc/libtaco_hip<suffix>.so and prints ms per forward and the decoder / post-net stage times.   bash tools/scratch/ab_lib.sh "" _fpd4
for v in "$@"; do
  TACO_LIB=$GRAFT_REPO_ROOT/multi-speaker-tacotron-tensorflow_amd/csrc/libtaco_hip$v.so python bench.py --no-cpu-baseline --no-companions --steps 30 --warmup 5 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages']; print('variant [$v] %.4f ms per forward; decoder alone %.4f, post-net alone %.4f, encoder %.4f' % (d['ms_per_step'], s['decoder']['ms_alone_eager'], s['postnet']['ms_alone_eager'], s['encoder']['ms_alone_eager']))"
done
