# cache / fabric counters of the three attention kernels (small separate passes: the TCC block takes few counters at once), folded by tools/pmc_fold.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_attn; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/attn_only.py > $O/p$i.log 2>&1
  echo "pass $i ($set): rc $?"
done
