#!/bin/bash
# round 3: why is k_level_mt slow?  SQ / LDS / TCC counters per dispatch (K = 64 target, one iteration) + T / row-block sweeps
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
run_pmc() {  # name counters...
  local name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OLDPWD/$O/pmc_$name -- python $OLDPWD/tools/probe.py --iters 1 --targets 10 --stats 0 > $OLDPWD/$O/pmc_$name.log 2>&1 )
  python tools/pmc_summary.py $O/pmc_$name --seq k_level_ > $O/pmc_${name}_summary.txt 2>&1
}
run_pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR
run_pmc tcc TCC_HIT_sum TCC_MISS_sum
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
for n in sq1 sq2 tcc fetch write; do echo "== $n"; grep -A400 "# per dispatch" $O/pmc_${n}_summary.txt | grep -E "k_level_mt|k_level_root" | tail -64; done > $O/pmc_per_dispatch.txt
tail -120 $O/pmc_per_dispatch.txt | cut -c1-160
echo "== sweeps"
for T in 1 2 8 64; do echo "MT_TREES=$T"; RGBM_MT_TREES=$T timeout 300 python tools/probe.py --iters 3 --targets 10 2>&1 | grep "^target" | tail -1; done | tee $O/sweep_trees.log
for B in 8 16 32 64; do echo "MT_TREES=4 MT_BLOCKS=$B"; RGBM_MT_TREES=4 RGBM_MT_BLOCKS=$B timeout 300 python tools/probe.py --iters 3 --targets 10 2>&1 | grep "^target" | tail -1; done | tee $O/sweep_blocks.log
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
