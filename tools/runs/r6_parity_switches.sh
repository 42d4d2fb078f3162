#!/bin/bash
# Round 6: parity A/B - which of our rounding choices stands between the outputs and the literal 1e-4 (tools/runs/r6_parity_ab.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r6_h}
mkdir -p $O
{
timeout 900 python tools/runs/r6_parity_ab.py oracle /tmp/r6_parity
echo "== default (Winograd, fp32 pooling atomics, v_exp / v_rcp sigmoid)"; timeout 300 python tools/runs/r6_parity_ab.py gpu /tmp/r6_parity
echo "== pooling sums in fixed point"; timeout 300 python tools/runs/r6_parity_ab.py gpu /tmp/r6_parity fixed
echo "== libm sigmoid in the GRU gates"; FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_libm.so timeout 300 python tools/runs/r6_parity_ab.py gpu /tmp/r6_parity
echo "== both"; FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_libm.so timeout 300 python tools/runs/r6_parity_ab.py gpu /tmp/r6_parity fixed
echo "== direct form instead of Winograd"; FIERY_CONV_WINOGRAD=0 timeout 300 python tools/runs/r6_parity_ab.py gpu /tmp/r6_parity
echo "== direct form, fixed-point pooling, libm sigmoid"; FIERY_CONV_WINOGRAD=0 FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_libm.so timeout 300 python tools/runs/r6_parity_ab.py gpu /tmp/r6_parity fixed
} 2>&1 | grep -v amdgpu.ids | tee $O/parity_ab.txt
