#!/bin/bash
# The A/Bs of round 6's small-batch kernels at C3 in one pass (experiments build for the comparison arms; rows compared bit for bit inside every tool):
#   product: one-wave tasks / LDS-shared tiles per batch size; gadget walk: every form; recombination: tiled + K splits / k_recombine_wg; rocprofv3 trace of one call of 64 and of 16
# usage: tools/small_batch_ab.sh <tag>    -> gpurun_out/<tag>_small_batch_*.log
tag=${1:-r06}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 300 python3 tools/stream_wg_ab.py c3 11 17 24 32 33 48 64 128 256 > $O/${tag}_small_batch_product.log 2>&1
timeout 300 python3 tools/gadget_mid_ab.py 1 4 8 16 32 64 128 256 > $O/${tag}_small_batch_gadget.log 2>&1
timeout 300 python3 tools/tail_ab.py k_gadget 4,8,16,24,32,64 default: row:PSF_GADGET_ROW=100000000,PSF_GADGET_WAVE=0 quad:PSF_GADGET_WAVE=0,PSF_GADGET_ROW=0 >> $O/${tag}_small_batch_gadget.log 2>&1
timeout 300 python3 tools/tail_ab.py k_recombine 5,16,32,48,64 tiled:PSF_RECOMBINE_STREAM=0 wg: > $O/${tag}_small_batch_recombine.log 2>&1
timeout 300 bash tools/prof_single_call.sh ${tag}b64 64 > /dev/null 2>&1; cp $O/${tag}b64_trace_single.txt $O/${tag}_small_batch_trace_b64.txt
timeout 300 bash tools/prof_single_call.sh ${tag}b16 16 > /dev/null 2>&1; cp $O/${tag}b16_trace_single.txt $O/${tag}_small_batch_trace_b16.txt
tail -n 12 $O/${tag}_small_batch_product.log $O/${tag}_small_batch_recombine.log; tail -n 16 $O/${tag}_small_batch_gadget.log; cat $O/${tag}_small_batch_trace_b64.txt
