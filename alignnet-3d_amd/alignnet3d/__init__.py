"""MI355X-native engine for the AlignNet-3D `tp8` network: Python host side above the
C ABI of libalignnet_hip.so (include/alignnet_hip.h).  There is NO CPU fallback: importing
works anywhere, but creating an Engine without the built library or without a GPU raises."""
from .engine import Engine, EngineError, default_model_config, OUTPUT_NAMES  # noqa: F401
from ._capi import load_library, library_path  # noqa: F401
