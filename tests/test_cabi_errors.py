"""Error behaviour of the C ABI (include/mmae_hip.h): every entry point validates its arguments on the host BEFORE any HIP
call and returns MMAE_ERR_ARG (-1) -- so these run without a GPU.  The Python binding turns a non-zero code into
MmaeLibraryError; host tensors are refused (there is no CPU path)."""
import ctypes

import pytest
import torch

from incomplete_multimodal_fusion_amd import _lib

QUERIES = {"mmae_abi_version", "mmae_last_hip_error", "mmae_modattn_bwd_nsplit",
           "mmae_add_ln_bwd_ws_floats", "mmae_hardneg_ws_floats", "mmae_mha_bwd_ws_floats",
           "mmae_gemm_nt_supported", "mmae_gemm_geglu_supported", "mmae_gemm_tn_supported", "mmae_gemm_tn_ws_floats",
           "mmae_mha_bwd_fused_supported", "mmae_mha_bwd_fused_ws_floats"}           # setters / size / shape queries: no pointers to validate


def test_every_entry_point_rejects_null_pointers():
    l = _lib.lib()
    for name, (_, argt) in _lib.parse_header().items():
        if name in QUERIES:
            continue
        args = [None if a is ctypes.c_void_p else a(0) for a in argt]
        assert getattr(l, name)(*args) == -1, name


def test_attention_argument_checks():
    l = _lib.lib()
    buf = (ctypes.c_char * 4096)()
    a = ctypes.addressof(buf)
    a16 = (a + 15) // 16 * 16
    P = ctypes.c_void_p

    def fwd(dtype=1, dh=64, B=1, H=8, nseg=4, q=a16, stride=1536, seg=a16, rows=64):
        return l.mmae_mha_fwd(dtype, dh, B, H, nseg, P(q), P(a16), P(a16), P(a16), P(a16), stride, stride, stride, 512, rows,
                              P(seg), P(a16), P(a16), P(a16), 64, 64, 0.125, 0, None)
    assert fwd(dh=48) == -1                 # head_dim 32 / 64 only
    assert fwd(dtype=7) == -1               # fp32 / bf16 only
    assert fwd(q=a16 + 2) == -1             # operands must be 16-byte aligned
    assert fwd(stride=1531) == -1           # row strides in multiples of 8 elements
    assert fwd(B=0) == -1 and fwd(H=0) == -1 and fwd(nseg=0) == -1
    assert fwd(seg=0) == -1                 # missing segment table
    assert fwd(rows=0) == -1


def test_row_kernel_argument_checks():
    l = _lib.lib()
    buf = (ctypes.c_char * 4096)()
    a16 = (ctypes.addressof(buf) + 15) // 16 * 16
    P = ctypes.c_void_p
    assert l.mmae_geglu_fwd(9, 16, 2048, P(a16), P(a16), None) == -1          # dtype
    assert l.mmae_adamw_step(6, P(a16), P(a16), P(a16), P(a16), None, 1e-3, 0.9, 0.95, 1e-8, 0.05, 1, 1.0, None) == -1  # n % 4
    assert l.mmae_adamw_step(8, P(a16), P(a16), P(a16), P(a16), None, 1e-3, 0.9, 0.95, 1e-8, 0.05, 0, 1.0, None) == -1  # step >= 1
    assert l.mmae_splitk_sum(0, 64, P(a16), P(a16), None) == -1               # S >= 1
    assert l.mmae_splitk_sum(2, 60, P(a16), P(a16), None) == -1               # n % 8
    off = (ctypes.c_long * 15)()
    assert l.mmae_descriptor_layout(2, 3, 16, 24, off) == off[14] > 0
    assert l.mmae_descriptor_layout(-1, 3, 16, 24, off) == -1


def test_gemm_argument_checks():
    l = _lib.lib()
    buf = (ctypes.c_char * 4096)()
    a16 = (ctypes.addressof(buf) + 15) // 16 * 16
    P = ctypes.c_void_p
    ok = lambda *a: l.mmae_gemm_nt_supported(*a)
    assert ok(163840, 4096, 768, 768, 768, 4096) == 1 and ok(1000, 256, 384, 384, 384, 256) == 1
    assert ok(1024, 300, 768, 768, 768, 300) == 0 and ok(1024, 256, 320, 320, 320, 256) == 0 and ok(1024, 256, 768, 760, 768, 256) == 0
    assert l.mmae_gemm_nt(1024, 256, 768, P(a16), 768, P(a16), 768, P(a16 + 4), 256, None) == -1      # C must be 8-byte aligned
    assert l.mmae_gemm_nt(1024, 256, 768, P(a16 + 8), 768, P(a16), 768, P(a16), 256, None) == -1      # A must be 16-byte aligned
    assert l.mmae_gemm_nt(1024, 250, 768, P(a16), 768, P(a16), 768, P(a16), 250, None) == -1
    assert l.mmae_gemm_geglu_supported(4096, 2048, 768, 768, 768, 4096, 2048) == 1
    assert l.mmae_gemm_geglu_supported(4096, 2000, 768, 768, 768, 4000, 2000) == 0                    # F % 128
    assert l.mmae_gemm_geglu(4096, 2048, 768, P(a16), 768, P(a16), 768, P(a16), 4096, None, 2048, None) == -1
    assert l.mmae_mha_bwd_ws_floats(8, 1000) == 24000 and l.mmae_mha_bwd_ws_floats(0, 5) == -1


def test_binding_raises_and_refuses_host_tensors():
    with pytest.raises(_lib.MmaeLibraryError, match="invalid argument"):
        _lib.call("mmae_shadow_bf16", 6, None, None, None)
    with pytest.raises(_lib.MmaeLibraryError, match="no CPU path"):
        _lib.ptr(torch.zeros(4))
    with pytest.raises(_lib.MmaeLibraryError, match="unsupported dtype"):
        _lib.dt(torch.float16)
