cd $GRAFT_REPO_ROOT
timeout 400 python3 tools/call_throughput.py 64 4 2>&1 | grep -v amdgpu.ids | tail -5
timeout 300 python3 tools/call_throughput.py 64 8 2>&1 | grep -v amdgpu.ids | tail -3
