# same-box A/B of environment switches: bash tools/gpu/ab_env.sh "<configs>" "VAR=a" "VAR=b" ...   (each twice, interleaved)
CFGS=$1; shift
mkdir -p gpurun_out/r5
for rep in 1 2; do
  for kv in "$@"; do
    for c in $CFGS; do
      env $kv python bench.py --config $c --no-other-configs --cpu-budget 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', '$c', d['value'], d['ms_per_step'])"
    done
  done
done 2>&1 | tee gpurun_out/r5/ab_env.log
