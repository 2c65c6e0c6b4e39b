"""ctypes binding of libcmr_hip.so.  Prototypes are parsed from include/cmr_hip.h (the single
source of truth for the C ABI), so a symbol that the header declares and the library lacks --
or the other way round -- is an import-time error.  There is NO fallback: if the library is
missing the first op call raises."""
import ctypes
import os
import re

import torch  # noqa: F401  -- MUST be imported before libcmr_hip.so is loaded: both link libamdhip64, and the HIP
#                            runtime copy that torch ships has to be the one in the process (loading ours first made
#                            every later kernel launch fail with hipErrorInvalidDeviceFunction-class errors)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libcmr_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "cmr_hip.h")

_SCALARS = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "hipStream_t": ctypes.c_void_p}
_RET = {"int": ctypes.c_int, "int64_t": ctypes.c_int64}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every `cmr_*` prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t)\s+(cmr_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes, argnames = [], []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            toks = a.replace("*", " * ").split()
            argnames.append(toks[-1])
            if "*" in toks:
                argtypes.append(ctypes.c_void_p)
            else:
                base = [t for t in toks[:-1] if t != "const"][0]
                argtypes.append(_SCALARS[base])
        protos[name] = (_RET[ret], argtypes, argnames)
    return protos


class CmrError(RuntimeError):
    pass


_lib = None
_protos = None


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CmrError("HIP library %s is missing -- build it with `make` (or __graft_entry__.build()); "
                       "cmr_agent_amd has no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (ret, argtypes, _) in _protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise CmrError("libcmr_hip.so does not export %s (declared in include/cmr_hip.h)" % name)
        fn.restype = ret
        fn.argtypes = argtypes
    _lib = lib
    if os.environ.get("CMR_B16_MM") == "0":             # A/B measurements: the two-team bf16 convolution for every layer
        lib.cmr_set_conv_bf16_variant(0, 0)
    return lib


def prototypes():
    load()
    return _protos


_ERR = {-1: "invalid argument (shape / alignment / null pointer)", -2: "kernel launch failed",
        -3: "not served by this entry point"}


UNSUPPORTED = -3


def call(name, *args, allow_unsupported=False, work_extra=None):
    """Invoke an int-returning entry point; raises CmrError on a non-zero status
    (returns UNSUPPORTED instead of raising when the caller has a fallback entry point).
    work_extra: sizes the arguments do not carry (rows behind a CSR), for utils/workmodel.CallTimer only."""
    rc = getattr(load(), name)(*args)
    if rc == UNSUPPORTED and allow_unsupported:
        return rc
    if rc != 0:
        raise CmrError("%s failed: %s (rc=%d)" % (name, _ERR.get(rc, "?"), rc))
