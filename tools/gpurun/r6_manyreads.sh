# round 6: more than 4.9 M reads in one context (the counting pass in windows of groups) - tools/huge_run.py
cd $GRAFT_REPO_ROOT
timeout 2400 python tools/huge_run.py ${R6_READS:-6000000} ${R6_GENOME:-1500000000} 8 gpurun_out/r06_beyond_4_9M_reads.json 2>&1 | grep -v "^{" | tail -14
