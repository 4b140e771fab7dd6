#!/bin/bash
# usage (GPU box): tools/pmc_sweep.sh <tag> [lib]  -- a wider SQ counter sweep (several --pmc passes) of the bench step's kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/pmc_sw_${1:?tag}"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
if [ -n "$2" ]; then export ENSLAM_LIB=$R/$2 ENSLAM_LIB_ALLOW_MISSING=1; fi
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --steps 10 --warmup 3 --eager --no-secondary --no-cpu-baseline --no-kernel-events > $O/p$i.log 2>&1
done
python3 - $O <<'PY'
import csv, sys, glob, collections, re, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + '/p*/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(render_fwd_ring_kernel|decoder_bwd_split_kernel|decoder_chain_kernel|decoder_dw_kernel|step_kernel|sample_kernel|composite_fwd_kernel)", r["Kernel_Name"])
        if not m: continue
        acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); n[m.group(1)][r["Counter_Name"]] += 1
out = {k: {c: round(v / n[k][c]) for c, v in sorted(acc[k].items())} for k in acc}
json.dump(out, open(sys.argv[1] + '/sweep.json', 'w'), indent=1)
for k in ('render_fwd_ring_kernel', 'decoder_bwd_split_kernel', 'decoder_chain_kernel', 'decoder_dw_kernel'):
    print(k)
    for c, v in out.get(k, {}).items(): print(f"   {c:34s} {v:>14,d}")
PY
