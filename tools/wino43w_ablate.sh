#!/bin/bash
# what each part of the F(4x4,3x3) weight-gradient kernel costs (csrc/conv_wino43w.hip, W43W_ABLATE bit mask): one library per
# mask built HERE (hipcc cross-compiles), then on the GPU box the 3x3 layers are timed with each (tools/wino43w_bench.py).
#   tools/wino43w_ablate.sh build [masks]   (build container; variants -> tools/scratch/w43wlibs/, git-ignored, travels with gpurun)
#   tools/wino43w_ablate.sh run [configs]   (GPU box)  -> gpurun_out/wino43w_ablation.txt
# bits: 1 no patch / dY loads, 2 no transforms / operand stores, 4 no MFMAs, 8 no operand reads (LDS), 16 no epilogue
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = build ]; then
    MASKS=${2:-"0 1 2 3 4 12 16 19 27"}
    mkdir -p $R/tools/scratch/w43wlibs && cd $R/mmd-gan_amd && python build_ext.py > /dev/null || exit 1
    rm -f $R/tools/scratch/w43wlibs/*.so
    for m in $MASKS; do
        ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-variable -DW43W_ABLATE=$m $W43W_DEFS -c csrc/conv_wino43w.hip -o /tmp/w43w_$m.o &&
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/scratch/w43wlibs/lib_$m.so $(ls build/*.o | grep -v conv_wino43w.o) /tmp/w43w_$m.o || echo "mask $m failed" ) &
        [ $(jobs -r | wc -l) -ge 6 ] && wait -n
    done
    wait; ls $R/tools/scratch/w43wlibs
    exit 0
fi
shift
export BENCH_REPS=50
mkdir -p $R/gpurun_out
cp $R/mmd-gan_amd/lib/libmmdgan_hip.so /tmp/lib_keep.so
for f in $(ls $R/tools/scratch/w43wlibs/lib_*.so | sort -t_ -k2 -n); do
    m=$(basename $f .so); m=${m#lib_}
    echo "W43W_ABLATE=$m (1 loads, 2 transforms + operand stores, 4 MFMAs, 8 operand reads, 16 epilogue)"
    cp $f $R/mmd-gan_amd/lib/libmmdgan_hip.so
    timeout 120 python $R/tools/wino43w_bench.py ${@:-cifar} 2>&1 | grep -v "amdgpu.ids\|^config"
done > $R/gpurun_out/wino43w_ablation.txt
cp /tmp/lib_keep.so $R/mmd-gan_amd/lib/libmmdgan_hip.so
cat $R/gpurun_out/wino43w_ablation.txt
