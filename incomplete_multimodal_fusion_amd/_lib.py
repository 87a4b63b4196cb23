"""ctypes binding of libmmae_hip.so (the C ABI declared in include/mmae_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel call fails, this module raises.
`include/mmae_hip.h` is the single source of truth: prototypes are parsed from it so argtypes can never drift.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMAE_HIP_LIB") or os.path.join(_HERE, "csrc", "libmmae_hip.so")   # env: A/B runs of another build
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mmae_hip.h")
INTERNAL_HEADER_PATH = os.path.join(_HERE, "csrc", "mmae_internal.h")      # test / tuning entry points, not the product ABI

F32, BF16 = 0, 1

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(mmae_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace("const", "").split()[0]
                    argtypes.append(_CT[base])
        protos[name] = (_CT[ret], argtypes)
    return protos


class MmaeLibraryError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise MmaeLibraryError(
                "libmmae_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C incomplete_multimodal_fusion_amd/csrc`. There is no CPU / eager fallback." % LIB_PATH)
        # torch ships its own libamdhip64 (torch/lib); it must be in the process BEFORE this library resolves its
        # libamdhip64.so.7 dependency, or the loader adds /opt/rocm's copy as a SECOND HIP runtime that does not know torch's
        # allocations (hipMemsetAsync on a torch pointer then fails while plain kernel launches still "work").
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        protos = dict(parse_header())
        protos.update(parse_header(INTERNAL_HEADER_PATH))
        for name, (ret, argtypes) in protos.items():
            fn = getattr(l, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = argtypes
        want = int(re.search(r"#define\s+MMAE_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read()).group(1))
        if l.mmae_abi_version() != want:            # a stale .so against a newer header (or the reverse): rebuild
            raise MmaeLibraryError("ABI version mismatch: library %d, include/mmae_hip.h %d" % (l.mmae_abi_version(), want))
        _lib = l
    return _lib


_ERR = {-1: "invalid argument (MMAE_ERR_ARG)", -2: "HIP launch failed (MMAE_ERR_LAUNCH)"}


_FN = {}          # entry point name -> ctypes function object (a getattr on the CDLL per launch costs more than the launch)


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        detail = ""
        if rc == -2:
            code = lib().mmae_last_hip_error()
            fn = lib().mmae_hip_error_name
            fn.restype, fn.argtypes = ctypes.c_char_p, [ctypes.c_int]
            detail = " [hipError %d %s]" % (code, (fn(code) or b"?").decode())
        raise MmaeLibraryError("%s failed: %s%s" % (name, _ERR.get(rc, rc), detail))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MmaeLibraryError("mmae HIP kernels need device tensors (got %s); there is no CPU path" % t.device)
    return ctypes.c_void_p(t.data_ptr())


_RAW = None


def raw_stream() -> int:
    """hipStream_t of torch's current stream on the current device, as an integer (0 = the default stream).  Through the two C
    accessors: torch.cuda.current_stream() builds a Stream object behind three Python layers (~10 us -- there are ~500 launches
    per step, and the small configurations are host-bound)."""
    global _RAW
    if _RAW is None:
        import torch
        torch.cuda.current_stream()                       # lazy CUDA init, once
        _RAW = (torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice)
    return _RAW[0](_RAW[1]())


def stream():
    return ctypes.c_void_p(raw_stream())


def dt(t_or_dtype):
    import torch
    d = t_or_dtype.dtype if hasattr(t_or_dtype, "dtype") else t_or_dtype
    if d == torch.float32:
        return F32
    if d == torch.bfloat16:
        return BF16
    raise MmaeLibraryError("unsupported dtype %s (fp32 / bf16 only)" % d)
