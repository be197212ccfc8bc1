#!/bin/bash
# ablation variants of the Winograd kernel.  usage: tools/run_wprobe.sh build | run [N H C K]
cd "$(dirname "$0")"
declare -A V=( [full]="" [novstore]="-DWINO_ABLATE_VSTORE" [noxload]="-DWINO_ABLATE_XLOAD" [noustore]="-DWINO_ABLATE_USTORE"
 [nouload]="-DWINO_ABLATE_ULOAD" [noepi]="-DWINO_ABLATE_EPILOGUE"
 [mfma_nobar]="-DWINO_ABLATE_VSTORE -DWINO_ABLATE_XLOAD -DWINO_ABLATE_USTORE -DWINO_ABLATE_ULOAD -DWINO_ABLATE_EPILOGUE -DWINO_ABLATE_BARRIER"
 [mfma_nofrag]="-DWINO_ABLATE_VSTORE -DWINO_ABLATE_XLOAD -DWINO_ABLATE_USTORE -DWINO_ABLATE_ULOAD -DWINO_ABLATE_EPILOGUE -DWINO_ABLATE_BARRIER -DWINO_ABLATE_FRAG"
 [mfmaonly]="-DWINO_ABLATE_VSTORE -DWINO_ABLATE_XLOAD -DWINO_ABLATE_USTORE -DWINO_ABLATE_ULOAD -DWINO_ABLATE_EPILOGUE" )
if [ "$1" = build ]; then
  for k in "${!V[@]}"; do hipcc -w -O3 --offload-arch=gfx950 -I../include ${V[$k]} -DVARIANT="\"$k\"" wino_probe.hip -o wprobe_$k.bin & done; wait
else
  shift
  for k in full novstore noxload noustore nouload noepi mfmaonly mfma_nobar mfma_nofrag; do ./wprobe_$k.bin "$@"; done
fi
