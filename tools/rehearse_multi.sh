#!/bin/bash
# bench.py --gpus N exactly as the driver starts it (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment),
# with all N ranks on the ONE GPU of a gpurun box and tests/mock_rccl standing in for RCCL: a rehearsal of the code
# path at the default (cfg 3) size, NOT a multi-GPU measurement (the stand-in is synchronous and goes through the host).
# usage: tools/rehearse_multi.sh N [bench args]
set -o pipefail
N=${1:-2}; shift
cd "$(dirname "$0")/.."
mkdir -p build gpurun_out
[ build/libmock_rccl.so -nt tests/mock_rccl/mock_rccl.cpp ] || hipcc -O2 -fPIC -shared --offload-arch=gfx950 -o build/libmock_rccl.so tests/mock_rccl/mock_rccl.cpp -lrt -lpthread
export MFX_K1_RES=levels MFX_TESTING=1 MFX_RCCL_LIB=$PWD/build/libmock_rccl.so MFX_BENCH_DEVICE=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29$((RANDOM % 900 + 100)) WORLD_SIZE=$N
pids=()
for ((r = 0; r < N; r++)); do
  RANK=$r LOCAL_RANK=$r python bench.py --gpus $N "$@" > gpurun_out/rehearse_rank$r.out 2> gpurun_out/rehearse_rank$r.err &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
grep -h "^{" gpurun_out/rehearse_rank0.out
tail -n 3 gpurun_out/rehearse_rank*.err | grep -v "^$" | tail -n 12
exit $rc
