#!/bin/bash
# round 6: same-box A/B of library builds through the LM step rate (scripts/dbg_fused.py rate): LIBS="a.so b.so" CFGS="cfg2"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/${OUT:-r6_ab.txt}; : > $out
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in $LIBS; do
    echo "== $lib" >> $out
    VXBA_LIB=$PWD/$lib timeout 600 python scripts/dbg_fused.py rate ${CFGS:-cfg2} 2>&1 | grep -v amdgpu.ids | grep "fused=1" >> $out
  done
done
cat $out
