#!/bin/bash
# Runs on the GPU box: bench.py plain, under rocprofv3 --kernel-trace --stats, and under separate --pmc passes (one run
# per pass: rocprofv3 does not split counters over passes by itself; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Everything lands in gpurun_out/prof/; tools/profile_digest.py turns it into the files kept under profiles/.
# usage: [PROF_DIR=name] tools/profile_bench.sh [bench.py args]      (the second workload of bench.py is switched off: one workload per profile)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${PROF_DIR:-prof}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --no-north-star "$@" > $O/bench.json || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --no-north-star --no-cpu-baseline "$@" > $O/bench_under_rocprof.json || exit 1
find $O/stats -name '*kernel_trace.csv' -delete      # (tens of MB per run; the summaries are made from *kernel_stats.csv)
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $O/pmc$i -o p -- python3 $R/bench.py --no-north-star --no-cpu-baseline --steps 4 --warmup 1 "$@" > /dev/null 2>$O/pmc$i.err || exit 1
done
# occupancy as rocprofv3 derives it (needs the accumulating read of SQ_LEVEL_WAVES); optional: a failure here keeps the rest
timeout -s KILL 120 rocprofv3 --pmc MeanOccupancyPerCU MeanOccupancyPerActiveCU --output-format csv -d $O/pmc_occ -o p -- python3 $R/bench.py --no-north-star --no-cpu-baseline --steps 4 --warmup 1 "$@" > /dev/null 2>$O/pmc_occ.err || echo "occupancy pass failed" >&2
echo done
