# level2_kernel's grid cap (quarters of the CU count; _ab/libbronko_hip.so reads BK_X_L2_CAP1) away from the sweet spot, four in flight
cd $GRAFT_REPO_ROOT
export BRONKO_HIP_LIB=$PWD/_ab/libbronko_hip.so
for c in 8 4 3 2; do echo "cap1 $c/4"; BK_X_L2_CAP1=$c timeout 200 python3 tools/stress_probe.py release "0.5 %,5 %,random" 4 2>&1 | grep " bp"; done
