# A/B of an environment switch on the bench in one call on one box: [AB_VALUES="0 2"] ab_bench.sh VAR [bench args]
set -e
cd "$(dirname "$0")/../.."
VAR=$1; shift
for rep in 1 2; do
for m in ${AB_VALUES:-0 1}; do
  echo "== $VAR=$m"
  env $VAR=$m timeout -k 10 200 python bench.py --no-extras "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('phase_ms_per_call'))"
done
done
