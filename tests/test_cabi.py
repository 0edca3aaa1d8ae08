"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/c4_engine.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build_engine()
    from connect4_amd import _lib
    return _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "c4_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(c4_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 20
    L = lib.load()
    for n in names:
        assert hasattr(L, n), "libc4engine.so does not export %s" % n
    assert sorted(lib.SIGNATURES) == names, "python binding and header disagree"
    assert L.c4_abi_version() == lib.ABI_VERSION


def test_struct_sizes_match_header(lib):
    # c4_game_record / c4_root_result cross the ABI by value: sizes must match the C layout
    assert ctypes.sizeof(lib.GameRecord) == 8 + 4 + 4 + 42 * 8 * 2 + 42 * 4 + 42 * 8 + 42 * 7 * 8
    assert ctypes.sizeof(lib.Stats) == 17 * 8
    assert ctypes.sizeof(lib.ExportBuffers) == 9 * 8
    assert ctypes.sizeof(lib.Config) % 8 == 0


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from connect4_amd.engine import Engine, board_wins
    with pytest.raises(lib.EngineError) as ei:
        Engine(8, 10)
    assert ei.value.code == lib.EDEVICE
    with pytest.raises(lib.EngineError):
        board_wins([15])


def test_bad_config_rejected(lib):
    from connect4_amd.engine import Engine
    with pytest.raises(lib.EngineError):
        Engine(0, 10)
    with pytest.raises(lib.EngineError):
        Engine(4, 10, eval_mode=7)
