# rows-stationary adaLN product against skinny_gemm_kernel: same bytes out, time of a launch
# POCKET_TTS_WIDE_ROWS: 0 = skinny_gemm_kernel, 1 = 16 rows per wave, two workgroups per CU, 2 or unset = 32 rows per wave, one per CU (whole 128-row slabs)
set -e
cd "$(dirname "$0")/.."
for M in 640 1280 600; do
  KB_DUMP_M=$M KB_DUMP=/tmp/ada_wide.bin POCKET_TTS_WIDE_ROWS=1 timeout -k 10 60 ./kbench ada_dump
  KB_DUMP_M=$M KB_DUMP=/tmp/ada_skinny.bin POCKET_TTS_WIDE_ROWS=0 timeout -k 10 60 ./kbench ada_dump
  cmp /tmp/ada_wide.bin /tmp/ada_skinny.bin && echo "M=$M: identical bytes"
done
for rep in 1 2; do
  for m in 0 1 2; do echo "== POCKET_TTS_WIDE_ROWS=$m"; POCKET_TTS_WIDE_ROWS=$m timeout -k 10 60 ./kbench ada; done
done
