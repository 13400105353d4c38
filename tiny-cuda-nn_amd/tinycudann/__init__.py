"""tinycudann -- MI355X-native drop-in for the `tinycudann` PyTorch extension of tiny-cuda-nn.

Public surface (same as the reference's bindings/torch/tinycudann/__init__.py:9-11):
    Encoding, Network, NetworkWithInputEncoding, free_temporary_memory
plus `native.create_from_config` (the C++ user API of the reference, reachable from Python here).
"""
from .modules import Encoding, Module, Network, NetworkWithInputEncoding, free_temporary_memory  # noqa: F401
from .native import Trainer, create_from_config  # noqa: F401
