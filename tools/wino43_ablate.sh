#!/bin/bash
# what each part of the F(4x4,3x3) kernel costs (csrc/conv_wino43.hip, W43_ABLATE bit mask): one library per mask built HERE
# (hipcc cross-compiles), then on the GPU box the 3x3 layers are timed with each (tools/wino43_bench.py, graph replay).
#   tools/wino43_ablate.sh build [masks]   (build container; variants -> tools/scratch/w43libs/, git-ignored, travels with gpurun)
#   tools/wino43_ablate.sh run [configs]   (GPU box)  -> gpurun_out/wino43_ablation.txt
# bits: 1 no patch loads, 2 no input transform / V stores, 4 no MFMAs, 8 no B-fragment loads, 16 no accumulator exchange /
# output transform, 32 no output stores
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = build ]; then
    MASKS=${2:-"0 1 3 4 8 16 48 51 59"}
    mkdir -p $R/tools/scratch/w43libs && cd $R/mmd-gan_amd && python build_ext.py > /dev/null || exit 1
    rm -f $R/tools/scratch/w43libs/*.so
    for m in $MASKS; do
        ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -DW43_ABLATE=$m -c csrc/conv_wino43.hip -o /tmp/w43_$m.o &&
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/scratch/w43libs/lib_$m.so $(ls build/*.o | grep -v conv_wino43.o) /tmp/w43_$m.o || echo "mask $m failed" ) &
        [ $(jobs -r | wc -l) -ge 6 ] && wait -n
    done
    wait; ls $R/tools/scratch/w43libs
    exit 0
fi
shift
export BENCH_REPS=50
mkdir -p $R/gpurun_out
cp $R/mmd-gan_amd/lib/libmmdgan_hip.so /tmp/lib_keep.so
for f in $(ls $R/tools/scratch/w43libs/lib_*.so | sort -t_ -k2 -n); do
    m=$(basename $f .so); m=${m#lib_}
    echo "W43_ABLATE=$m (1 patch loads, 2 transform + V stores, 4 MFMAs, 8 B loads, 16 exchange + output transform, 32 stores)"
    cp $f $R/mmd-gan_amd/lib/libmmdgan_hip.so
    timeout 120 python $R/tools/wino43_bench.py ${@:-cifar} 2>&1 | grep -v "amdgpu.ids\|^config"
done > $R/gpurun_out/wino43_ablation.txt
cp /tmp/lib_keep.so $R/mmd-gan_amd/lib/libmmdgan_hip.so
cat $R/gpurun_out/wino43_ablation.txt
