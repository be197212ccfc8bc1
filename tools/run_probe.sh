#!/bin/bash
# builds the ablation variants of the igemm main loop and prints time per launch for R = 1, 3, 5
# (D l3 shape: 128 x 16 x 16 x 128 -> 128).  usage: tools/run_probe.sh build | run
cd "$(dirname "$0")"
declare -A V=(
 [full]=""
 [noglobal]="-DMMDGAN_ABLATE_GLOBAL"
 [nostore]="-DMMDGAN_ABLATE_STORE"
 [noglobal_nostore]="-DMMDGAN_ABLATE_GLOBAL -DMMDGAN_ABLATE_STORE"
 [nobarrier]="-DMMDGAN_ABLATE_BARRIER"
 [nofrag_full]="-DMMDGAN_ABLATE_FRAG"
 [mfmaonly]="-DMMDGAN_ABLATE_GLOBAL -DMMDGAN_ABLATE_STORE -DMMDGAN_ABLATE_BARRIER -DMMDGAN_ABLATE_FRAG"
 [noepi]="-DMMDGAN_ABLATE_EPILOGUE"
)
if [ "$1" = build ]; then
  for k in "${!V[@]}"; do
    hipcc -w -O3 --offload-arch=gfx950 -I../include ${V[$k]} -DVARIANT="\"$k\"" igemm_probe.hip -o probe_$k.bin &
  done
  wait
else
  for k in full noglobal nostore noglobal_nostore nobarrier nofrag_full mfmaonly noepi; do
    for R in 1 3 5; do ./probe_$k.bin $R; done
  done
fi
