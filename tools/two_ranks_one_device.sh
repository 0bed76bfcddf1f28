#!/bin/bash
# two ranks of one RCCL communicator on ONE device (both LOCAL_RANK=0): does RCCL accept it?
cd $GRAFT_REPO_ROOT
D=$(mktemp -d)
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 SVIHMM_TEST_TAG=t2on1 SVIHMM_RENDEZVOUS_TIMEOUT=40 HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout 120 python tests/_rccl_worker.py $D tests/golden/metaobs_K4_D2_L10_mask.npz > $D/out$r.log 2>&1 &
done
wait
for r in 0 1; do echo "== rank $r"; tail -5 $D/out$r.log; done
ls $D
