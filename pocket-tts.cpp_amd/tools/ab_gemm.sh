set -e
cd "$(dirname "$0")/.."
CASES="gemm_ada gemm_pre_qkv gemm_pre_ff2 gemm_dec_qkv gemm_dec_ff2 gemm_conv0 gemm_dec_up1 gemm_dec_up2 gemm_dec_c1_2 gemm_dec_c2_2"
for rep in 1 2; do
for m in 0 1; do
  echo "== POCKET_TTS_GEMM_XCD_MAP=$m"
  POCKET_TTS_GEMM_XCD_MAP=$m timeout -k 10 120 ./kbench $CASES
done
done
