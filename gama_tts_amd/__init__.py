"""MI355X-native batched vocal-tract-model synthesizer (GamaTTS VTM hot path).

The product is the HIP/C++ library under gama_tts_amd/csrc (C ABI: include/gama_vtm.h,
GamaTTS plugin: libgama_vtm_plugin.so).  This Python package is only the ctypes binding
used by the tests, bench.py and __graft_entry__.py; it never computes audio itself and
raises when the native library is missing.
"""
from .capi import (  # noqa: F401
    GvtmError,
    PinnedArray,
    Plan,
    Stream,
    TrackConfig,
    config5_from_dict,
    config_from_dict,
    device_count,
    library_path,
    load_library,
    read_config_file,
)
