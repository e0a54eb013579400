set -o pipefail
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_planar.py -x -q -m gpu > gpurun_out/r04/gpu_planar_1.log 2>&1; echo "planar rc=$?"; tail -25 gpurun_out/r04/gpu_planar_1.log
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_planar.py > gpurun_out/r04/gpu_tests_2.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r04/gpu_tests_2.log
