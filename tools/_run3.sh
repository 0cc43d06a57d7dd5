cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_prio; mkdir -p $O
L=/root/repo/build_variants
tools/ab_env.sh $O/ab.txt 3 "|--chunks 64 --frames 200:400" "prio0:XV_LIB=$L/prio0/libxvector_hip.so" "prio3:XV_B=0" "prio1:XV_LIB=$L/prio1/libxvector_hip.so" "prio3s3:XV_GEMM_SLOTS=3" "prio3t768:XV_TN_TARGET=768" > /dev/null
tools/step_timeline.sh $O/tl_prio3 -- > /dev/null 2>&1
XV_GEMM_SLOTS=3 tools/step_timeline.sh $O/tl_prio3s3 -- > /dev/null 2>&1
cat $O/ab.txt
