cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for ko in 0; do
  rm -rf /tmp/pe_$ko
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d /tmp/pe_$ko -o pe -- $ROOT/tools/probe/emb_probe > /tmp/pe_$ko.log 2>&1
  echo "== KO $ko"; tail -1 /tmp/pe_$ko.log
  python $ROOT/tools/rocpd_summary.py $(find /tmp/pe_$ko -name "*.db" | head -1) 2>&1 | tail -15
done
