"""nesti_net_amd -- MI355X-native Nesti-Net inference hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
every kernel lives in ``libnesti_hip.so`` behind the C-ABI declared in
``include/nesti_hip.h``.  There is no CPU fallback: importing :mod:`._lib`
raises if the library has not been built (``python -c "import __graft_entry__ as g; g.build()"``).
"""
from .config import NestiConfig, DTYPES  # noqa: F401
from . import synth  # noqa: F401

__all__ = ["NestiConfig", "DTYPES", "synth"]
