# FUSE_E: what is left of the finishing launch (lists C and X, the left list) by its knobs: blocks per region on list X, waves on list C
export DCRX_DEBUG_FLAGS=1
R=$GRAFT_REPO_ROOT; cd /tmp
O=$R/gpurun_out/r06_e15; mkdir -p $O
export DCRX_LIB_PATH=$R/tools/variants/libdcrx_fuse_e.so
run() { n=$1; shift
  python3 $R/bench.py --no-cpu-baseline --steps 60 --warmup 10 "$@" 2>$O/err.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $n', d['ms_per_step'], d.get('ms_per_step_steady'), d['roofline']['dominant_kernel_ms_avg'], round(d['roofline']['step_device_ms_avg']-d['roofline']['dominant_kernel_ms_avg'],4))" || tail -3 $O/err.log
}
export DCRX_DEBUG_TAIL_WAVES=2 DCRX_DEBUG_FUSE_E_WAVES=3
(
for rep in 1 2; do
DCRX_DEBUG_FUSE_E=0 DCRX_DEBUG_TAIL_WAVES=0 run "role"
run "fused"
for b in 2 4 8; do DCRX_DEBUG_X_BSPLIT=$b run "fused_x_bsplit_$b"; done
for c in 256 1024 2048; do DCRX_DEBUG_RESCUE_WAVES_C=$c run "fused_c_waves_$c"; done
DCRX_DEBUG_X_BSPLIT=4 DCRX_DEBUG_RESCUE_WAVES_C=1024 run "fused_x4_c1024"
DCRX_DEBUG_X_BSPLIT=8 DCRX_DEBUG_RESCUE_WAVES_C=512 run "fused_x8_c512"
done
) 2>&1 | tee $O/raw.log
