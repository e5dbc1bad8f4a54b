#!/bin/bash
# PMC passes for the bench kernel (separate runs per counter set; no tracing flags mixed in).
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc; rm -rf /tmp/pmcq; mkdir -p /tmp/pmcq
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmcq/p$i -o pmc -- python $ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline ${BENCH_ARGS:-} > /tmp/pmcq/log$i.txt 2>&1)
  f=$(find /tmp/pmcq/p$i -name "*counter_collection*" | head -1)
  [ -n "$f" ] && python $ROOT/tools/pmc_summary.py "$f" jpeg_coeffs | tee gpurun_out/pmc/pass$i.txt
done
