# what list X costs the finishing launch of the fused form: the bench without exception bytes (DCRX_BENCH_N_RATE=0: list X empty) beside the default
R=$GRAFT_REPO_ROOT
export DCRX_DEBUG_FLAGS=1 DCRX_DEBUG_FUSE_E=1 CHECK=0
echo "== default (0.05 % of the reads carry an exception byte)"; bash $R/tools/r06_experiments/r06_ab.sh r06_e22_default 3 help | tail -n 1
echo "== no exception bytes"; DCRX_BENCH_N_RATE=0 bash $R/tools/r06_experiments/r06_ab.sh r06_e22_no_exc 3 help | tail -n 1
export DCRX_DEBUG_FUSE_E=0
echo "== role form, default"; bash $R/tools/r06_experiments/r06_ab.sh r06_e22_role_default 3 help | tail -n 1
echo "== role form, no exception bytes"; DCRX_BENCH_N_RATE=0 bash $R/tools/r06_experiments/r06_ab.sh r06_e22_role_no_exc 3 help | tail -n 1
