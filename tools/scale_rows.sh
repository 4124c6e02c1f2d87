cd $GRAFT_REPO_ROOT
{ for n in 100 250 500; do timeout 600 python3 tools/scale_probe.py $n 4 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/r06_b_scale.txt
cat gpurun_out/r06_b_scale.txt
