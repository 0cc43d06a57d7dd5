cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_slots; mkdir -p $O
tools/ab_env.sh $O/ab.txt 2 "|--chunks 64 --frames 200:400" "base:XV_B=0" "s3:XV_GEMM_SLOTS=3" "s3b:XV_GEMM_SLOTS=3b" "t768:XV_TN_TARGET=768" "t1536:XV_TN_TARGET=1536" "t2048:XV_TN_TARGET=2048" "s3t1536:XV_GEMM_SLOTS=3 XV_TN_TARGET=1536" > /dev/null
XV_GEMM_SLOTS=3 tools/step_timeline.sh $O/tl_s3 -- > /dev/null 2>&1
XV_GEMM_SLOTS=3b tools/step_timeline.sh $O/tl_s3b -- > /dev/null 2>&1
cat $O/ab.txt
