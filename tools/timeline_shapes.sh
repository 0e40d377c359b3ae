set -e
export HWOCR_GEMM_ABLATE=10 HWOCR_GEMM256=2   # the eight-wave kernel carries the stamps
python tools/bench_gemm_timeline.py 41472 5120 1280 > gpurun_out/tl_fc1.txt 2>&1
python tools/bench_gemm_timeline.py 41472 1280 5120 > gpurun_out/tl_fc2.txt 2>&1
python tools/bench_gemm_timeline.py 21504 17920 1536 > gpurun_out/tl_gateup.txt 2>&1
python tools/bench_gemm_timeline.py 21504 1536 8960 > gpurun_out/tl_down.txt 2>&1
cat gpurun_out/tl_fc1.txt gpurun_out/tl_fc2.txt gpurun_out/tl_gateup.txt gpurun_out/tl_down.txt
