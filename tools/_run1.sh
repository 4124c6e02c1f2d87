cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hpv or regional or two_words or batched or split or hot_bin or amplicon or forked or planes_are_clean or config2 or counter_planes or sharded_finalize_on_one_device" 2>&1 | tail -15
BENCH_ARGS="--steps 10 --warmup 4" bash tools/env_ab.sh 2 BK_NO_FUSE=1 BK_ITEM_CAPS=64,32 BK_ITEM_CAPS=64,24 BK_ITEM_CAPS=56,32 2>&1 | tail -20
