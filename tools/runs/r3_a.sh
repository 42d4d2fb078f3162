#!/bin/bash
# Round 3, first GPU call: hardware / host-library semantics the kernels rely on, the warp tests, the guard-page hunt for the
# round-2 memory fault (training from images), then the whole GPU suite.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_a
mkdir -p $O
lscpu | grep -E "Model name|^CPU\(s\)" > $O/host.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/probe/buffer_oob_probe.hip -o /tmp/buffer_oob_probe && /tmp/buffer_oob_probe > $O/buffer_oob_probe.txt 2>&1
cat $O/buffer_oob_probe.txt
python tools/probe/aten_warp_probe.py > $O/aten_warp_probe.txt 2>&1; grep -c True $O/aten_warp_probe.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "warp" -x 2>&1 | tail -15
# guard-page hunt: every tensor against an unmapped page, launches serialised, library calls traced
export AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1
timeout 900 python tools/guard_alloc/run_guarded.py --trace call tests.test_train_graph:_full_size_training_steps_reduce_the_loss > $O/guard_train_after.out 2> $O/guard_train_after.err
echo "guard(after) train rc=$?"; tail -5 $O/guard_train_after.out; grep -v "^\[fiery\]" $O/guard_train_after.err | tail -8; grep "^\[fiery\]" $O/guard_train_after.err | tail -3
timeout 900 python tools/guard_alloc/run_guarded.py --trace --before call tests.test_train_graph:_full_size_training_steps_reduce_the_loss > $O/guard_train_before.out 2> $O/guard_train_before.err
echo "guard(before) train rc=$?"; grep -v "^\[fiery\]" $O/guard_train_before.err | tail -8; grep "^\[fiery\]" $O/guard_train_before.err | tail -3
unset AMD_SERIALIZE_KERNEL HIP_LAUNCH_BLOCKING
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
cp $R/gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
