# the scan's share of the CUs with four samples in flight (testing build, BK_ITEM_SHARE in sixteenths), after Level 2's planned grid
cd "$GRAFT_REPO_ROOT" || exit 1
export BRONKO_HIP_LIB=$PWD/bronko_amd/libbronko_hip_testing.so
for i in 1 2; do
  for v in "" BK_ITEM_SHARE=9 BK_ITEM_SHARE=10 BK_ITEM_SHARE=11 BK_ITEM_SHARE=12 BK_ITEM_SHARE=13; do
    env $v timeout 120 python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 4 --experiment 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-(none)}'.ljust(28), '%.4g' % d['value'], '%.4f' % d['ms_per_sample'], '%.4f' % d['serial_ms_per_sample'])"
  done
done
