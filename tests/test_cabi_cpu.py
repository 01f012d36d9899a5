"""CPU: the C-ABI shared library loads and exports exactly the symbols include/spider_hip.h declares
(no compute calls without a GPU), the ctypes table mirrors the header, and argument validation reports errors."""
import ctypes
import os
import re

from spider_amd import lib as slib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "spider_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"(?:int|const char\s*\*)\s+(spider_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_exports_every_declared_symbol():
    decls = _header_decls()
    assert len(decls) >= 28
    lib = ctypes.CDLL(slib.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in spider_hip.h but missing from libspider_hip.so"


def test_ctypes_table_mirrors_header():
    decls = _header_decls()
    assert set(decls) == set(slib.SIGNATURES), set(decls) ^ set(slib.SIGNATURES)
    for name, n in decls.items():
        assert len(slib.SIGNATURES[name][1]) == n, name


def test_identity_and_argument_validation():
    lib = slib.load()
    assert lib.spider_abi_version() == 4
    assert lib.spider_target_arch() == b"gfx950"
    assert lib.spider_lm_head_nparts(152064) == 2048 and lib.spider_groupnorm_nchunk(4096) == 128
    # validation happens on the host before any launch: bad shapes return -1 with a message, no GPU needed
    assert lib.spider_gemv_bf16(None, None, None, None, None, None, 0.0, 17, 16, 64, None) == -1
    assert b"batch" in lib.spider_last_error()
    assert lib.spider_gemm_bf16(None, None, None, None, None, None, None, 0, 4, 4, 7, 8, 4, 0, 1.0, 0, None, None, None, 0, None) == -1
    assert lib.spider_gemm_f16(None, None, None, None, None, None, None, 0, 4, 4, 7, 8, 4, 0, 1.0, 0, None, None, None, 0, None) == -1
    assert lib.spider_attn_bf16(None, None, None, None, *([0] * 12), 1, 8, 8, 16, 16, 200, 1.0, 0, 0, None, None, 0, 0, None) == -1
    assert b"head_dim" in lib.spider_last_error()
    try:
        slib.call("spider_rmsnorm_bf16", None, None, None, None, None, 1, 12, 1e-6, None)
        raise AssertionError("expected SpiderHipError")
    except slib.SpiderHipError as e:
        assert "multiple of 8" in str(e)
