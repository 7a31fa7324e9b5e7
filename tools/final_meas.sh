set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/bench_modes.sh gpurun_out/r02_bench_modes.jsonl > gpurun_out/r02_bench_modes.txt 2>&1
cat gpurun_out/r02_bench_modes.txt
( echo "# tools/area_scales.sh 4 (BGRA) then 3 (BGR): INTER_AREA across shrink factors, uniform resident batches"; bash tools/area_scales.sh 4; bash tools/area_scales.sh 3 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_area_scales.txt
python bench.py --mixed 4096 --steps 10 > gpurun_out/r02_mixed_bgra.json
python bench.py --mixed 4096 --steps 10 --channels 3 > gpurun_out/r02_mixed_bgr.json
bash tools/stream_scaling.sh gpurun_out/r02_stream_scaling.txt > /dev/null 2>&1
python tools/perf_survey.py > gpurun_out/r02_operator_survey.txt 2>&1
python bench.py > gpurun_out/r02_bench.json 2>/dev/null
tail -1 gpurun_out/r02_bench.json | cut -c1-300
