#!/bin/bash
O=gpurun_out/r3_run2; mkdir -p $O
timeout 900 tools/bin/probe_coissue 60 32 > $O/probe_coissue.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$O/list_avail.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -i "mfma\|coexec\|VALU" $O/list_avail.txt | head -80 > $O/counters_mfma.txt
tail -60 $O/probe_coissue.log
