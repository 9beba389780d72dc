"""pytest plugin of tools/sanitize_cpu.sh: points the product's loader and the oracle's at the sanitizer-instrumented scratch
builds under $HSK_SAN_DIR (the test suite itself is unchanged; without the variable the plugin does nothing)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SAN = os.environ.get("HSK_SAN_DIR")
if SAN:
    import housescan_amd._lib as _lib
    from oracle import oracle as _oracle
    # (importing the package has already bound the shipped library: every call goes through _lib.load(), so dropping its
    #  cache and binding again is enough)
    _lib.LIB_PATH = os.path.join(SAN, "libhskinfu.so")
    _lib._lib = None
    _lib.load()

    def _lib_path(omp=False, _orig=_oracle.lib):
        # oracle.lib(False / True): the instrumented plain / OpenMP builds; every other form (native, literal, variants) as it is
        if omp in (False, True):
            name = "libkinfu_oracle_omp.so" if omp else "libkinfu_oracle.so"
            if name not in _oracle._libs:
                here, _oracle._HERE = _oracle._HERE, SAN
                try:
                    return _orig(omp)
                finally:
                    _oracle._HERE = here
        return _orig(omp)

    _oracle.lib = _lib_path
