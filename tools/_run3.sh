cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) 2>&1 | tail -12
BENCH_ARGS="--steps 10 --warmup 4" bash tools/env_ab.sh 2 BK_NO_FUSE=1 BK_ITEM_CAPS=64,32 BK_ITEM_CAPS=64,24 BK_ITEM_CAPS=40,24 2>&1 | tail -12
