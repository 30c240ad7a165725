#!/bin/bash
# Timings + two PMC passes of scripts/r4/tlb_probe.py (program directly after `--`; --pmc alone: gpurun refuses mixed modes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4/tlb
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 $R/scripts/r4/tlb_probe.py > $OUT/plain.txt 2> $OUT/plain.err && \
timeout -k 10 600 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $OUT/pmc1 -- python3 $R/scripts/r4/tlb_probe.py > $OUT/pmc1.txt 2> $OUT/pmc1.err && \
timeout -k 10 600 rocprofv3 --pmc TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum --output-format csv -d $OUT/pmc2 -- python3 $R/scripts/r4/tlb_probe.py > $OUT/pmc2.txt 2> $OUT/pmc2.err && \
timeout -k 10 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $OUT/pmc3 -- python3 $R/scripts/r4/tlb_probe.py > $OUT/pmc3.txt 2> $OUT/pmc3.err
echo rc=$?
for d in pmc1 pmc2 pmc3; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Kernel_Name|k_spmv|k_stream_read" $f > $OUT/${d}_counters.csv; rm -rf $OUT/$d; done
ls -la $OUT; cat $OUT/plain.txt
