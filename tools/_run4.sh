cd $GRAFT_REPO_ROOT
L=/root/repo/build_variants/diag/libxvector_hip.so
echo "== dgrad3 (beside wgrad3)"; XV_LIB=$L XV_DIAG_M=24576 XV_DIAG_N=512 XV_DIAG_K=3584 timeout 300 python3 tools/step_clock.py 2>&1 | grep -v amdgpu.ids
echo "== fwd3 (alone)"; XV_LIB=$L XV_DIAG_M=23808 XV_DIAG_N=512 XV_DIAG_K=3584 timeout 300 python3 tools/step_clock.py 2>&1 | grep -v amdgpu.ids
echo "== dgrad5 (beside wgrad5)"; XV_LIB=$L XV_DIAG_M=23808 XV_DIAG_N=512 XV_DIAG_K=1500 timeout 300 python3 tools/step_clock.py 2>&1 | grep -v amdgpu.ids
