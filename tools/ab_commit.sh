#!/bin/bash
# Interleaved A/B of the working tree against another commit of this repository, on one GPU box (box-to-box spread is
# ~1 %, the changes being measured are often smaller).
#   here (no GPU):   tools/ab_commit.sh prepare [commit]      -> tools/scratch/base: that commit's package, built, with its
#                                                               bench.py / oracle / fixture (git-ignored; travels with gpurun)
#   on the GPU box:  tools/ab_commit.sh run [config] [steps]  -> ms/step of base / new, eager and plan, two rounds
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/tools/scratch/base
case "${1:-run}" in
prepare)
  c=${2:-HEAD}
  rm -rf /tmp/ab_base "$B"
  git -C "$R" worktree add -f /tmp/ab_base "$c" > /dev/null
  (cd /tmp/ab_base && python mmd-gan_amd/build_ext.py > /dev/null)
  mkdir -p "$B/tests/golden"
  cp -r /tmp/ab_base/mmd-gan_amd /tmp/ab_base/oracle /tmp/ab_base/bench.py "$B/"
  rm -rf "$B/mmd-gan_amd/build"
  cp /tmp/ab_base/tests/shipped_step.py /tmp/ab_base/tests/helpers.py "$B/tests/"
  cp /tmp/ab_base/tests/golden/production_kernels.json "$B/tests/golden/"
  git -C "$R" worktree remove --force /tmp/ab_base
  echo "base = $(git -C "$R" rev-parse --short "$c") under $B"
  ;;
run)
  cfg=${2:-cifar}; steps=${3:-200}
  cd "$R"
  for rnd in 1 2; do
    for w in base new; do
      if [ $w = base ]; then d=tools/scratch/base; else d=.; fi
      for mode in eager plan; do
        python $d/bench.py --config $cfg --no-cpu-baseline --steps $steps --repeats 3 --launch-mode $mode 2>/dev/null | tail -1 | python -c "
import json, sys
o = json.loads(sys.stdin.read())
print('%-4s %-5s round $rnd  %.4f ms/step' % ('$w', '$mode', o['ms_per_step']), o['ms_per_step_regions'])"
      done
    done
  done
  ;;
esac
