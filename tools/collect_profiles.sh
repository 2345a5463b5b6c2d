#!/bin/bash
# gpurun merges only gpurun_out/ back: copy the round's evidence from gpurun_out/<tag>/profiles/ into profiles/ (tracked)
TAG=${1:-r04}
O=gpurun_out/$TAG/profiles
cp $O/* profiles/
cp $O/${TAG}_pmc.json profiles/pmc.json
[ -f $O/valu_mix.json ] && cp $O/valu_mix.json profiles/valu_mix.json
ls -la profiles | grep $TAG
