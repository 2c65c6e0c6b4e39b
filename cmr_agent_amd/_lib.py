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
# the A/B build (-DCMR_AB_SWITCHES): the same entry points + the kernel-variant switches of include/cmr_hip_ab.h.  Tests that compare two
# kernels bit for bit and tools/*_bench.py load it through ab(); the product never does.
AB_LIB_PATH = os.path.join(_PKG, "lib", "libcmr_hip_ab.so")
AB_HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "cmr_hip_ab.h")

_SCALARS = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "hipStream_t": ctypes.c_void_p}
_RET = {"int": ctypes.c_int, "int64_t": ctypes.c_int64}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every `cmr_*` prototype."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r'#include\s+"cmr_hip.h"', "", text)
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
_ab_lib = None


def _bind(path, protos):
    if not os.path.exists(path):
        raise CmrError("HIP library %s is missing -- build it with `make` (or __graft_entry__.build()); "
                       "cmr_agent_amd has no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    for name, (ret, argtypes, _) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise CmrError("%s does not export %s (declared in its header)" % (os.path.basename(path), name))
        fn.restype = ret
        fn.argtypes = argtypes
    return lib


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    _protos = parse_header()
    _lib = _bind(LIB_PATH, _protos)
    return _lib


def load_ab():
    """The A/B library (every entry point of the product + the cmr_set_* switches)."""
    global _ab_lib
    if _ab_lib is None:
        load()
        protos = dict(_protos)
        protos.update(parse_header(AB_HEADER_PATH))
        _ab_lib = _bind(AB_LIB_PATH, protos)
    return _ab_lib


def use_ab():
    """tools/*_bench.py: switch this process to the A/B library for good (every later ops.* call and load() return it)."""
    global _lib
    _lib = load_ab()
    return _lib


class ab:
    """with _lib.ab() as lib: every ops.* call inside goes to libcmr_hip_ab.so, whose switches `lib.cmr_set_*` select kernel variants
    (tests / tools only).  Same kernels, same results as the product library under the default settings."""

    def __enter__(self):
        global _lib
        self.old = load()
        _lib = load_ab()
        return _lib

    def __exit__(self, *a):
        global _lib
        _lib = self.old


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
