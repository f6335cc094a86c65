# rocprofv3 profile of the round on the GPU box, one BASELINE config per call (a config takes 2-6 minutes):
#   gpurun --timeout 1500 -- 'bash tools/gpu/profile_all.sh r05 bench_esrgan'
# -> gpurun_out/prof_<round>_<config>/{kernel_stats.csv, kernel_stats_chains.csv, pmc_summary.json, sq_summary.json}
# (scratch: copy what should be judged into profiles/<round>_<config>_*; bench.py cites the newest committed summaries).
# An optional third argument is exported as NEOSR_AMD_FAST_MATMUL (1 = profile the fast_matmul tier).
ROUND=${1:-r05}; CFG=${2:-bench_esrgan}
[ -n "$3" ] && export NEOSR_AMD_FAST_MATMUL=$3
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_cfg.sh $ROUND $CFG
bash tools/profile_sq.sh $ROUND $CFG
ls -la gpurun_out/prof_${ROUND}_${CFG}
