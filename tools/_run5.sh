cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_prio; mkdir -p $O
L=/root/repo/build_variants
tools/ab_env.sh $O/ab2.txt 2 "--frames 400|--attention|--extended --frames 400|" "prio0:XV_LIB=$L/prio0/libxvector_hip.so" "prio0s3b:XV_LIB=$L/prio0/libxvector_hip.so XV_GEMM_SLOTS=3b" "prio3s3b:XV_GEMM_SLOTS=3b" "prio3:XV_B=0" > /dev/null
cat $O/ab2.txt
