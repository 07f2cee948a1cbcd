# scanning waves that have left their loop help to empty the rings (DCRX_V2_HELP_DRAIN): both forms, forced, interleaved on one box
R=$GRAFT_REPO_ROOT
export DCRX_DEBUG_FLAGS=1
DCRX_DEBUG_FUSE_E=1 bash $R/tools/r06_experiments/r06_ab.sh r06_e21_fused 6 help nohelp
DCRX_DEBUG_FUSE_E=0 CHECK=0 bash $R/tools/r06_experiments/r06_ab.sh r06_e21_role 6 help nohelp
