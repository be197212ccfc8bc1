#!/bin/bash
# what each part of the thin-layer n2w kernel costs (csrc/conv_thin_mfma.hip, THIN_ABLATE bit mask): builds one library per
# mask HERE (hipcc cross-compiles), then on the GPU box times the thin layers with each (tools/bench_conv.py, graph replay).
#   tools/thin_ablate.sh build          (in the build container; variants -> tools/scratch/thinlibs/, git-ignored, travels with gpurun)
#   tools/thin_ablate.sh run            (on the GPU box)  -> gpurun_out/thin_ablation.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MASKS="0 1 2 4 3 7"
if [ "$1" = build ]; then
    mkdir -p $R/tools/scratch/thinlibs && cd $R/mmd-gan_amd && python build_ext.py > /dev/null || exit 1
    for m in $MASKS; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DTHIN_ABLATE=$m -c csrc/conv_thin_mfma.hip -o /tmp/thin_$m.o &&
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/scratch/thinlibs/lib_$m.so $(ls build/*.o | grep -v conv_thin_mfma.o) /tmp/thin_$m.o || exit 1
    done
    exit 0
fi
export BENCH_GRAPH=1 BENCH_WARM=200 BENCH_REPS=50
mkdir -p $R/gpurun_out
cp $R/mmd-gan_amd/lib/libmmdgan_hip.so /tmp/lib_keep.so
for m in $MASKS; do
    echo "THIN_ABLATE=$m (1 stores, 2 gathers, 4 MFMAs)"
    cp $R/tools/scratch/thinlibs/lib_$m.so $R/mmd-gan_amd/lib/libmmdgan_hip.so
    timeout 120 python $R/tools/bench_conv.py 64 thin 2>&1 | grep "layer\|thin\|rror"
done > $R/gpurun_out/thin_ablation.txt
cp /tmp/lib_keep.so $R/mmd-gan_amd/lib/libmmdgan_hip.so
cat $R/gpurun_out/thin_ablation.txt
