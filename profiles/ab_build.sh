#!/bin/bash
# A/B helper: builds infinisst_amd/libinfinisst_hip_B.so from the CURRENT sources with extra -D flags (the A build is the normal
# library).  On the GPU box:  python bench.py ...; cp infinisst_amd/libinfinisst_hip_B.so infinisst_amd/libinfinisst_hip.so; python bench.py ...
# -- same box, same process state, only the library differs.   usage: profiles/ab_build.sh -DSOME_SWITCH=0
set -e
cd "$(dirname "$0")/../infinisst_amd/csrc"
mkdir -p /tmp/isst_ab
for f in gemm gemm_tiled gemm_mid rowops enc_attn llm_attn sample beam engine; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result "$@" -c $f.hip -o /tmp/isst_ab/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libinfinisst_hip_B.so /tmp/isst_ab/*.o
ls -la ../libinfinisst_hip_B.so
