# same-box A/B of library builds: neosr_amd/lib/var0.so, var1.so, ... (NEOSR_AMD_LIB picks one), each twice, interleaved
#   gpurun -- 'bash tools/gpu/ab_libs.sh "bench_hat_l_otf_gan bench_swinir_medium" var0 var1'
CFGS=$1; shift
mkdir -p gpurun_out/r5
for rep in 1 2; do
  for v in "$@"; do
    export NEOSR_AMD_LIB=$PWD/neosr_amd/lib/$v.so
    for c in $CFGS; do
      python bench.py --config $c --no-other-configs --cpu-budget 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$c', d['value'], d['ms_per_step'])"
    done
  done
done 2>&1 | tee gpurun_out/r5/ab_libs.log
