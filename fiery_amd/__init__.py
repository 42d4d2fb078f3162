"""fiery_amd: MI355X-native (gfx950) implementation of the FIERY camera-to-BEV hot path.

`Fiery` mirrors `fiery.models.fiery.Fiery` of wayveai/fiery; its BEV path runs on the hand-written HIP
kernels in `fiery_amd/csrc` (libfiery_hip.so, C ABI in include/fiery_hip.h).
"""
import os as _os



def exclude_miopen_nhwc_bwd_solver():
    """MIOpen's assembly implicit-GEMM backward-data kernels for pixel-major tensors (igemm_bwd_gtcx35_nhwc_fp32_*: what
    PyTorch-ROCm's convolutions of the image trunk get in training on gfx950) read past the end of their operand - measured
    with GPU guard pages on ROCm 7.0.2 / MIOpen of torch 2.10 (tools/guard_alloc): harmless when the
    bytes behind the tensor are mapped, "Memory access fault by GPU" when the caching allocator put the tensor at the end of a
    segment.  Called only where the library itself sends convolutions through MIOpen - the training graph built with the
    PyTorch-ROCm trunk (`FIERY_HIP_TRUNK=0` / `hip_trunk=False`) - because the variable is process-wide: it excludes the
    solver for every model in the process, and only takes effect if MIOpen has not evaluated it yet.  Logged once."""
    if _os.environ.get('MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC') is None:
        _os.environ['MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC'] = '0'
        import logging
        logging.getLogger('fiery_amd').warning(
            'fiery_amd: MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 set for this process (the PyTorch-ROCm image '
            'trunk in training: that MIOpen solver reads past its operand on gfx950); set the variable yourself to override')


from .config import get_cfg, get_parser, get_preset_cfg, CfgNode   # noqa: F401


def __getattr__(name):
    if name == 'Fiery':
        from .model import Fiery
        return Fiery
    raise AttributeError(name)
