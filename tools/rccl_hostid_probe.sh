#!/bin/bash
# Can two RCCL ranks share ONE GPU when each pretends to be a different host (NCCL_HOSTID)?  Transport then
# falls back to the socket/net path over loopback.  Used only to exercise the RCCL code path on a 1-GPU box.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2
export NCCL_SOCKET_IFNAME=lo NCCL_IB_DISABLE=1 NCCL_P2P_DISABLE=1 NCCL_SHM_DISABLE=1 NCCL_DEBUG=WARN
for r in 0 1; do
  RANK=$r NCCL_HOSTID=fakehost$r timeout -k 5 150 python tools/nccl_dup_probe.py nccl > gpurun_out/rccl_hostid_$r.log 2>&1 &
  pids[$r]=$!
done
rc=0
for r in 0 1; do wait ${pids[$r]} || rc=1; done
cat gpurun_out/rccl_hostid_0.log gpurun_out/rccl_hostid_1.log
exit $rc
