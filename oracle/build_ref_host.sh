#!/bin/bash
# oracle/build_ref_host.sh -- builds oracle/_ref/libref_host_<variant>_<DTYPE>.so from the
# REFERENCE'S OWN host loops, read in place from /root/reference (TEST INFRASTRUCTURE ONLY).
#
# The files that hold those loops (spmm_mul_csr.c, spmm_mul_coo.c, spmv_mul_coo.c, ops.hpp, spmm.h)
# include the UPMEM SDK's dpu.h for their device code, so the FILES do not compile here and no
# stand-in for dpu.h is written.  The host-oracle FUNCTIONS in them need nothing but the reference's
# own support/common.h and support/matrix.h.  So every translation unit is assembled at build time
# and piped to the compiler on stdin -- nothing of the reference is written to disk or committed:
#     <libc headers>  +  #include of the reference's support/common.h, support/matrix.h (in place)
#   + the text of the named functions / structs, cut out of the reference file BY NAME (awk below)
#   + oracle/ref_host_glue.inc (ours: flat-array entry points that fill the reference's structs)
# Compiled like the oracle itself (-fwrapv -ffp-contract=off; no -ffast-math), -O2, one library per
# val_dt (support/common.h:39-60) and per backend variant, because the variants reuse symbol names:
#   default : spmm_default/spmm_mul_csr.c  add_2D :41-50, matrix_add :55-60, memadd_2D :65-73,
#             memcpy_2D :78-86, spmm_host_csr :100-113; spmm.h memadd :106-110;
#             spmm_default/spmm_mul_coo.c spmm_host_coo :40-51;
#             spmm_default/ops.hpp spmm_host_csr_group :42-62, spmm_host_coo_group :97-118 (C++, as
#             the reference compiles it: ops.hpp is included by pytorch_api.cpp)
#   grande  : spmm_grande/spmm_mul_csr.c spmm_host_csr :119-136 (the VALUED loop), add_2D, memcpy_2D
#   spmv    : spmv_sparseP/spmv_mul_coo.c memadd :46-49, memadd_2D :54-62, memcpy_2D :67-75,
#             spmm_host :92-103, add_2D :107-115, spmm_host_group :128-148
# usage: build_ref_host.sh [reference backend_pim dir] [output dir]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${1:-/root/reference/backend_pim}"
OUT="${2:-$HERE/_ref}"
CC="${CC:-gcc}"
CXX="${CXX:-g++}"
FLAGS="-O2 -fPIC -fwrapv -ffp-contract=off -w -DNR_TASKLETS=16"
[ -d "$REF/spmm_default" ] || { echo "reference tree absent: keeping prebuilt oracle/_ref (if any)"; exit 0; }
mkdir -p "$OUT"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT

# definition of function NAME (first line matches `... NAME(` at column 0 and is not a prototype),
# up to the closing brace at column 0
fn() { awk -v name="$2" '
    !on && $0 ~ ("^(static |inline )*[A-Za-z_][A-Za-z0-9_ \\*]*[ \\*]" name "\\(") && $0 !~ /;[ \t]*$/ { on = 1 }
    on { print }
    on && /^}/ { exit }' "$1"; }
# definition of `struct NAME { ... };`
st() { awk -v name="$2" '
    !on && $0 ~ ("^struct " name "[ \t]*\\{") { on = 1 }
    on { print }
    on && /^};/ { exit }' "$1"; }
need() { [ -n "$2" ] || { echo "build_ref_host: $1 not found in the reference" >&2; exit 1; }; }
hdr() { printf '#include <stdint.h>\n#include <stdio.h>\n#include <stdlib.h>\n#include <string.h>\n#include "support/common.h"\n#include "support/matrix.h"\n'; }

for DT in INT8 INT16 INT32 INT64 FLT32 DBL64; do
  # ---------------- spmm_default: C loops + merge helpers, C++ group drivers ----------------
  D="$REF/spmm_default"
  { hdr
    t="$(fn "$D/spmm.h" memadd)"; need memadd "$t"; echo "$t"
    echo 'extern inline void memadd(val_dt*, val_dt*, uint32_t);'   # C99: emit the external definition
    for f in add_2D matrix_add memadd_2D memcpy_2D spmm_host_csr; do
      t="$(fn "$D/spmm_mul_csr.c" $f)"; need $f "$t"; echo "$t"; done
    t="$(fn "$D/spmm_mul_coo.c" spmm_host_coo)"; need spmm_host_coo "$t"; echo "$t"
  } | $CC $FLAGS -D$DT=1 -I"$D" -x c -c - -o "$TMP/default_c_$DT.o"
  { printf '#include <stdint.h>\n#include <stdio.h>\n#include <stdlib.h>\n#include <string.h>\nextern "C" {\n#include "support/common.h"\n#include "support/matrix.h"\n'
    for s in csr_info coo_info dense_info csr_info_group coo_info_group dense_info_group; do
      t="$(st "$D/spmm.h" $s)"; need "struct $s" "$t"; echo "$t"; done
    echo 'void spmm_host_csr(val_dt*, struct CSRMatrix*, val_dt*, uint32_t);'
    echo 'void spmm_host_coo(val_dt*, struct COOMatrix*, val_dt*, uint32_t);'
    echo 'void add_2D(val_dt*, val_dt*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);'
    echo '}'
    for f in spmm_host_csr_group spmm_host_coo_group; do
      t="$(fn "$D/ops.hpp" $f)"; need $f "$t"; echo "$t"; done
    echo '#define REF_GLUE_DEFAULT 1'
    cat "$HERE/ref_host_glue.inc"
  } | $CXX $FLAGS -D$DT=1 -I"$D" -x c++ -c - -o "$TMP/default_cpp_$DT.o"
  $CXX -shared "$TMP/default_c_$DT.o" "$TMP/default_cpp_$DT.o" -o "$OUT/libref_host_default_$DT.so"

  # ---------------- spmm_grande: the valued CSR loop ----------------
  D="$REF/spmm_grande"
  { hdr
    for f in add_2D memcpy_2D spmm_host_csr; do
      t="$(fn "$D/spmm_mul_csr.c" $f)"; need "grande $f" "$t"; echo "$t"; done
    echo '#define REF_GLUE_GRANDE 1'
    cat "$HERE/ref_host_glue.inc"
  } | $CC $FLAGS -D$DT=1 -I"$D" -x c -shared - -o "$OUT/libref_host_grande_$DT.so"

  # ---------------- spmv_sparseP: COO loop, merge helpers, group driver (all C) ----------------
  D="$REF/spmv_sparseP"
  { hdr
    for s in coo_info dense_info coo_info_group dense_info_group; do
      t="$(st "$D/spmm.h" $s)"; need "spmv struct $s" "$t"; echo "$t"; done
    t="$(fn "$D/spmv_mul_coo.c" memadd)"; need "spmv memadd" "$t"; echo "$t"
    echo 'extern inline void memadd(val_dt*, val_dt*, uint32_t);'
    for f in memadd_2D memcpy_2D spmm_host add_2D spmm_host_group; do
      t="$(fn "$D/spmv_mul_coo.c" $f)"; need "spmv $f" "$t"; echo "$t"; done
    echo '#define REF_GLUE_SPMV 1'
    cat "$HERE/ref_host_glue.inc"
  } | $CC $FLAGS -D$DT=1 -I"$D" -x c -shared - -o "$OUT/libref_host_spmv_$DT.so"
done
echo "built oracle/_ref/libref_host_{default,grande,spmv}_{INT8,INT16,INT32,INT64,FLT32,DBL64}.so from $REF (functions read in place)"
