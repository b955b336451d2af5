#!/bin/bash
# dev helper (authoring container): build libgnnb from the csrc of a git revision (or the working tree: rev = WORK) -> tools/ablate/<name>.so
# for same-box A/B runs on the GPU box (`tools/run.sh ab tools/ablate/a.so tools/ablate/b.so`; GNNB_LIB skips the source-hash check)
#   tools/mklib.sh <rev|WORK> <name> [extra hipcc flags]
set -e
REV=$1; NAME=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd); D=$(mktemp -d)
mkdir -p $D/gnn_branching_amd/csrc $D/include
if [ "$REV" = WORK ]; then cp $R/gnn_branching_amd/csrc/*.h $R/gnn_branching_amd/csrc/*.hip $D/gnn_branching_amd/csrc/; cp $R/include/gnnb.h $D/include/
else git -C $R archive $REV gnn_branching_amd/csrc include | tar -x -C $D; fi
mkdir -p $R/tools/ablate
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -DGNNB_SRC_HASH="\"ab-$NAME\"" "$@" -o $R/tools/ablate/$NAME.so $D/gnn_branching_amd/csrc/gnnb.hip
rm -rf $D; ls -la $R/tools/ablate/$NAME.so
