#!/bin/bash
O=gpurun_out/r3_run3; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 tools/bin/probe_coissue 60 32 > $O/probe_coissue.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES -d $R/$O/pmc1 -o p1 --output-format csv -- $R/tools/bin/probe_coissue 8 4 pmc > $R/$O/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES -d $R/$O/pmc2 -o p2 --output-format csv -- $R/tools/bin/probe_coissue 8 4 pmc > $R/$O/pmc2.log 2>&1
cd $R
find $O -name "*counter_collection.csv" | head; 
timeout 1500 python3 -m pytest tests/test_gpu_gpv_scale.py -q -m gpu -x 2>&1 | tail -30 > $O/gpv_scale.log
cat $O/gpv_scale.log; tail -45 $O/probe_coissue.log
