#!/bin/bash
# Builds the REAL reference (CornellHPC/HySortK) from the sources where they lie under
# /root/reference, directly with g++ (the reference's own Makefile is NOT run), into
# oracle/_ref/<variant>/.  Test infrastructure only: the result is used (a) to pin the C
# restatement in oracle/hsk_oracle.c against the reference, (b) to generate tests/golden/,
# (c) optionally as bench.py's cpu_baseline "reference" leg.  Nothing under oracle/ is on the
# product path.
#
# Needs: g++, and the MPICH 3.3.2 that this image ships under /opt/conda (mpi.h + libmpi.so).
# No reference source is copied into the repo; objects/binaries land in oracle/_ref/ (git-ignored).
#
#   usage: oracle/build_ref.sh <variant> K M L U EXT [SORT] [LOG]
#   e.g.   oracle/build_ref.sh k31 31 17 1 65535 0 2
set -euo pipefail
REF=${HSK_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
V=$1; K=$2; M=$3; L=$4; U=$5; EXT=$6; SORT=${7:-2}; LOG=${8:-1}
OUT="$HERE/_ref/$V"
COMMON="$HERE/_ref/common"
MPI_INC=${HSK_MPI_INC:-/opt/conda/include}
MPI_LIB=${HSK_MPI_LIB:-/opt/conda/lib/libmpi.so}
[ -d "$REF/src" ] || { echo "reference not present at $REF (GPU box?) - nothing to build"; exit 0; }
[ -f "$MPI_INC/mpi.h" ] || { echo "mpi.h not found at $MPI_INC: reference unbuildable here"; exit 0; }
mkdir -p "$OUT" "$COMMON"
DEFS="-DKMER_SIZE=$K -DMINIMIZER_SIZE=$M -DLOWER_KMER_FREQ=$L -DUPPER_KMER_FREQ=$U -DLOG_LEVEL=$LOG -DDEBUG=0 \
 -DTHREAD_PER_WORKER=4 -DMAX_SEND_BATCH=80000 -DMAX_THREAD_MEMORY_BOUNDED=16 -DSORT=$SORT -DAVG_TASK_PER_WORKER=3 \
 -DDISPATCH_UPPER_COE=1.5 -DDISPATCH_STEP=0.05 -DUNBALANCED_RATIO=2.3 -DPLAIN_CLASSIFIER=0 -DPLAIN_DISPATCHER=0 -DEXTENSION=$EXT"
FLAGS="-O3 -pthread -m64 -mavx2 -DTHREADED -fopenmp -std=c++17 -I$REF/include -I$REF/src -I$REF/dependency/Raduls -I$REF/dependency/Paradis -I$MPI_INC"
echo "$DEFS" > "$OUT/defs.txt"
pids=()
for f in logger dnaseq dnabuffer fastaindex hashfuncs kmerops memcheck hysortk; do
  if [ ! -f "$OUT/$f.o" ]; then g++ $FLAGS $DEFS -c -o "$OUT/$f.o" "$REF/src/$f.cpp" & pids+=($!); fi
done
# RADULS sorting networks do not depend on the -D macros: compile once (its own Makefile uses -O1)
if [ ! -f "$COMMON/sorting_network.o" ]; then
  g++ -O1 -m64 -mavx2 -std=c++17 -I"$REF/dependency/Raduls" -c -o "$COMMON/sorting_network.o" "$REF/dependency/Raduls/sorting_network.cpp" & pids+=($!)
fi
if [ ! -f "$OUT/standalone.o" ]; then g++ $FLAGS $DEFS -c -o "$OUT/standalone.o" "$REF/standalone/main.cpp" & pids+=($!); fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
OBJS=""; for f in logger dnaseq dnabuffer fastaindex hashfuncs kmerops memcheck hysortk; do OBJS="$OBJS $OUT/$f.o"; done
g++ -O3 -fopenmp -pthread -o "$OUT/hysortk_ref" "$OUT/standalone.o" $OBJS "$COMMON/sorting_network.o" "$MPI_LIB"
# stage-level harness (our own code, includes the reference headers at build time only)
if [ -f "$HERE/ref_harness.cpp" ]; then
  g++ $FLAGS $DEFS -o "$OUT/ref_harness" "$HERE/ref_harness.cpp" $OBJS "$COMMON/sorting_network.o" "$MPI_LIB"
fi
echo "built $OUT/hysortk_ref"
