# A/B two builds of the library in one GPU session: bash tools/ab_bench.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    OAKE_LIB=$L python bench.py --no-cpu-baseline --no-profile --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], d['value'])"
  done
done
