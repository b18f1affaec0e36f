#!/bin/bash
# Collects on a GPU box what profiles/ holds: rocprofv3 per-kernel statistics of the bench command (eager launches:
# rocprofv3 7.2 crashes inside hipGraphLaunch of the frame graph), the two HBM counter passes, the bench lines of both
# engines, the streaming latency probe, the encode throughput and the full-size parity notes.  Every GPU step has its own
# time limit and the script stops at the first failure.  Output: gpurun_out/profiles_new/.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out/profiles_new"
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
POCKET_TTS_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --latency-runs 0 > "$OUT/stats_run.log" 2>&1
POCKET_TTS_NO_GRAPH=1 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --frames 16 --no-cpu-baseline --latency-runs 0 > "$OUT/pmc_fetch.log" 2>&1
POCKET_TTS_NO_GRAPH=1 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --frames 16 --no-cpu-baseline --latency-runs 0 > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
F=$(ls "$OUT"/pmc_fetch/*/*counter_collection.csv | head -1)
W=$(ls "$OUT"/pmc_write/*/*counter_collection.csv | head -1)
python3 pocket-tts.cpp_amd/tools/pmc_traffic.py "$F" "$W" "$OUT/r01_pmc_traffic.json" > "$OUT/pmc_top.txt"
cp "$OUT/r01_pmc_traffic.json" profiles/r01_pmc_traffic.json      # bench.py reads the traffic figure from here
cp "$(ls "$OUT"/stats/*/*kernel_stats.csv | head -1)" "$OUT/r01_kernel_stats.csv"
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/stats"
timeout -k 10 600 python3 bench.py > "$OUT/r01_bench.json"
timeout -k 10 600 python3 bench.py --precision int8 --no-cpu-baseline > "$OUT/r01_bench_int8.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/latency_probe.py --runs 20 > "$OUT/r01_latency_fp32.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/latency_probe.py --runs 20 --precision bf16 > "$OUT/r01_latency_bf16.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/latency_probe.py --runs 20 --precision int8 > "$OUT/r01_latency_int8.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/encode_bench.py > "$OUT/r01_encode.json"
rm -f gpurun_out/fullsize_parity.txt
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_fullsize.py -x -q --timeout 200 --timeout-method thread > "$OUT/fullsize_run.log" 2>&1
cp gpurun_out/fullsize_parity.txt "$OUT/r01_fullsize_parity.txt"
echo done
