#!/bin/bash
# tools/gpu_round_profile.sh [quick] — run on the GPU box through gpurun: tests, bench line, rocprofv3 kernel stats, the two HBM PMC passes and the FP64
# instruction-count passes.  Everything lands under gpurun_out/; the summaries worth keeping are copied into profiles/ by tools/digest_round_profile.sh.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python qm_control_amd/record_model.py > gpurun_out/kernel_source_hash.txt      # the sources these counters belong to (bench.py: roofline `stale` on mismatch)
if [ "$1" != "noprof" ]; then
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
fi
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 1500 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
if [ "$1" == "quick" ]; then rocprofv3 -L > gpurun_out/counters.txt 2>&1; grep -i "F64\|MFMA" gpurun_out/counters.txt | head -40; exit 0; fi
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_flops_a gpurun_out/pmc_flops_b
B="python bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -- $B --steps 5 --warmup 2 > gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -- $B --steps 3 --warmup 1 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -- $B --steps 3 --warmup 1 > gpurun_out/pmc_write.log 2>&1
# FP64 instructions the kernels issue (wave-level counts): matrix-core ops in 512-flop units, vector FMA / MUL / ADD / transcendental
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --kernel-trace -d gpurun_out/pmc_flops_a -- $B --steps 3 --warmup 1 > gpurun_out/pmc_flops_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_MFMA SQ_WAVES --kernel-trace -d gpurun_out/pmc_flops_b -- $B --steps 3 --warmup 1 > gpurun_out/pmc_flops_b.log 2>&1
ls -R gpurun_out | head -60
rm -rf gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d gpurun_out/pmc_sq -- $B --steps 3 --warmup 1 > gpurun_out/pmc_sq.log 2>&1
tail -3 gpurun_out/pmc_sq.log
# LDS pipe and instruction mix (round 6: K1b turned out to be sensitive to LDS instructions and vector instructions, not to scalar ones — profiles/r06_ab_lq_regions.log)
rm -rf gpurun_out/pmc_lds
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pmc_lds -- $B --steps 3 --warmup 1 > gpurun_out/pmc_lds.log 2>&1
tail -3 gpurun_out/pmc_lds.log
