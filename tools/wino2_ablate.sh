#!/bin/bash
# tools/wino2_ablate.sh build   (here: hipcc cross-compiles the variants into tools/scratch/w2a/, which travels with gpurun)
# tools/wino2_ablate.sh run     (on the GPU box: gpurun -- 'bash tools/wino2_ablate.sh run' -> gpurun_out/wino2_ablation.txt)
cd "$(dirname "$0")"
D=scratch/w2a
MASKS="0 1 2 4 8 16 3 31 32 64 96 128 129 130 136 159"
if [ "$1" = build ]; then
  mkdir -p $D
  for m in $MASKS; do
    hipcc -w -O3 --offload-arch=gfx950 -I../include -I../mmd-gan_amd/csrc -DW2_ABLATE=$m wino2_ablate.hip -o $D/w2a_$m.bin &
    [ $(jobs -r | wc -l) -ge 8 ] && wait -n
  done
  wait; ls $D
else
  mkdir -p ../gpurun_out; L=../gpurun_out/wino2_ablation.txt; : > $L
  for rep in 1 2; do for m in $MASKS; do timeout 60 $D/w2a_$m.bin 512 >> $L 2>&1; done; done
  for m in 0 128 129 130 159; do timeout 60 $D/w2a_$m.bin 256 >> $L 2>&1; done
  for m in 0 128 159; do timeout 60 $D/w2a_$m.bin 512 128 32 64 128 0 >> $L 2>&1; timeout 60 $D/w2a_$m.bin 512 192 16 128 256 1 >> $L 2>&1; done
  cat $L
fi
