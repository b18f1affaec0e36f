#!/bin/bash
# Collects on a GPU box what profiles/ holds for one round (argument: the prefix, e.g. r02): rocprofv3 per-kernel
# statistics of the bench command on its default path (one hipGraph per frame), the two HBM counter passes, the bench
# lines (default workload with its extra legs, and --workload c4 with two engines, one engine, and bf16), per-kernel statistics of the int8 engine with eager
# launches, the streaming latency probe, the encode throughput and the parity notes of
# the full-size tests.  Every GPU step has its own time limit and the script stops at the first failure.
# Under rocprofv3 7.2 the first hipGraphLaunch of a frame graph segfaults now and then inside the profiler's queue hook
# (a memcpy that runs off the end of a mapped region; logs kept under profiles/).  The profiled runs therefore switch
# off the HIP runtime's pre-recorded packet batch for graphs (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0): the graph is still
# captured and launched with hipGraphLaunch, its kernels reach the queue one by one.
# Output: gpurun_out/profiles_new/.
set -euo pipefail
R="${1:-r02}"
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out/profiles_new"
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/${R}_stats_run.log" 2>&1
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --frames 16 --no-cpu-baseline --no-extras > "$OUT/pmc_fetch.log" 2>&1
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- \
    python3 "$ROOT/bench.py" --steps 1 --warmup 0 --frames 16 --no-cpu-baseline --no-extras > "$OUT/pmc_write.log" 2>&1
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_i8" -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --precision int8 > "$OUT/stats_i8_run.log" 2>&1
cd "$ROOT"
F=$(ls "$OUT"/pmc_fetch/*/*counter_collection.csv | head -1)
W=$(ls "$OUT"/pmc_write/*/*counter_collection.csv | head -1)
python3 pocket-tts.cpp_amd/tools/pmc_traffic.py "$F" "$W" "$OUT/${R}_pmc_traffic.json" > "$OUT/${R}_pmc_top.txt"
cp "$OUT/${R}_pmc_traffic.json" "profiles/${R}_pmc_traffic.json"      # bench.py reads the traffic figure from here
cp "$(ls "$OUT"/stats/*/*kernel_stats.csv | head -1)" "$OUT/${R}_kernel_stats.csv"
cp "$(ls "$OUT"/stats_i8/*/*kernel_stats.csv | head -1)" "$OUT/${R}_kernel_stats_int8.csv"
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/stats" "$OUT/stats_i8" "$OUT/pmc_fetch.log" "$OUT/pmc_write.log" "$OUT/stats_i8_run.log"
timeout -k 10 600 python3 bench.py > "$OUT/${R}_bench.json"
timeout -k 10 600 python3 bench.py --workload c4 --no-cpu-baseline > "$OUT/${R}_bench_c4.json"
timeout -k 10 600 python3 bench.py --workload c4 --no-cpu-baseline --no-extras --engines-per-gpu 1 > "$OUT/${R}_bench_c4_one_engine.json"
timeout -k 10 600 python3 bench.py --workload c4 --no-cpu-baseline --no-extras --precision bf16 > "$OUT/${R}_bench_c4_bf16.json"
timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-extras --precision fp32 > "$OUT/${R}_bench_fp32.json"
timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-extras --frames 500 > "$OUT/${R}_bench_500_frames.json"
if [ -x pocket-tts.cpp_amd/xcd_probe ]; then timeout -k 5 60 pocket-tts.cpp_amd/xcd_probe > "$OUT/${R}_xcd_probe.txt"; fi
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/latency_probe.py --runs 20 > "$OUT/${R}_latency_fp32.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/latency_probe.py --runs 20 --precision bf16 > "$OUT/${R}_latency_bf16.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/latency_probe.py --runs 20 --precision int8 > "$OUT/${R}_latency_int8.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/encode_bench.py > "$OUT/${R}_encode.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/encode_bench.py --wavs 1024 > "$OUT/${R}_encode_wavs.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/encode_bench.py --wavs 256 --wav-rate 48000 >> "$OUT/${R}_encode_wavs.json"
timeout -k 10 300 python3 pocket-tts.cpp_amd/tools/encode_bench.py --wavs 64 --wav-rate 44100 >> "$OUT/${R}_encode_wavs.json"
rm -f gpurun_out/fullsize_parity.txt gpurun_out/table_stats.txt
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_table.py -x -q --timeout 300 --timeout-method thread > "$OUT/fullsize_run.log" 2>&1
cp gpurun_out/fullsize_parity.txt "$OUT/${R}_fullsize_parity.txt"
cp gpurun_out/table_stats.txt "$OUT/${R}_table_stats.txt"
echo done
