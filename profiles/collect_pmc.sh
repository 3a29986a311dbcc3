#!/bin/bash
# GPU box: bench line + kernel-trace stats + PMC passes of the same command, each pass a run of its own
# (rocprofv3 does not split counters over passes; <= 8 SQ, 4 TCC, 2 GRBM counters per pass).
#   bash profiles/collect_pmc.sh <tag> [extra bench args]
# Results land under gpurun_out/<tag>/; profiles/pmc_issue.py and profiles/pmc_traffic.py turn them into the
# summaries that are committed under profiles/.
set -o pipefail
tag=${1:-r02}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline $*"
timeout -k 10 300 python3 bench.py $* > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $BENCH > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
pass() {  # name, counters...
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d $out/pmc_$name -o p --output-format csv -- $BENCH > $out/pmc_$name.log 2>&1 || { tail -5 $out/pmc_$name.log; exit 1; }
}
pass sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU
pass tcc TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
pass rd FETCH_SIZE
pass wr WRITE_SIZE
find $out -name "*.csv" -size +40M -delete   # keep the merge-back under 64 MiB
python3 scratch/pmc_summ.py $out/pmc_sq1 $out/pmc_sq2 $out/pmc_tcc > $out/pmc_summary.txt
cp $out/trace/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
python3 profiles/pmc_traffic.py $out/pmc_rd $out/pmc_wr > $out/traffic.json
cat $out/bench.json
