# FETCH_SIZE / WRITE_SIZE per true byte for this repo's access patterns (tools/micro/fetch_calibrate.hip) -> gpurun_out/<tag>_fetch_calibration.json
#   gpurun -- bash tools/fetch_calibrate.sh r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r04}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gpurun_out/fetch_calibrate tools/micro/fetch_calibrate.hip || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/cal_f -- ./gpurun_out/fetch_calibrate > gpurun_out/cal_true.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/cal_w -- ./gpurun_out/fetch_calibrate > /dev/null
python3 tools/fetch_calibrate.py gpurun_out/cal_true.txt $(find gpurun_out/cal_f gpurun_out/cal_w -name "*counter_collection.csv") > gpurun_out/${T}_fetch_calibration.json
cat gpurun_out/${T}_fetch_calibration.json
rm -rf gpurun_out/cal_f gpurun_out/cal_w gpurun_out/fetch_calibrate
