"""linr_pcgc_amd: MI355X-native (gfx950) coding network for LINR-PCGC.

Host side is Python on PyTorch-ROCm and mirrors the reference's operator surface (models/model_core.py,
models/upsample.py, models/resnet.py, models/module_utils.py, models/function_utils.py); all arithmetic of the hot path
runs in hand-written HIP kernels reached through the C-ABI in include/linr_hip.h (liblinr_hip.so, built in-tree by
csrc/build.sh).  There is no CPU or PyTorch fallback: a missing library raises at first use.
"""
__version__ = '0.1.0'
