# sourced by every script here that rebuilds the library with other than the default flags, or copies another library
# over it: whatever happens, the default build of THIS tree is back in place when the script exits (bench.py and the tests
# refuse a "+exp" or stale library).  The library is removed first: a copied-in .so is newer than the objects, and make
# would call it up to date.
restore_default_build() {
  rm -f "${GRAFT_REPO_ROOT:-.}/housescan_amd/libhskinfu.so"
  make -s -C "${GRAFT_REPO_ROOT:-.}/housescan_amd/csrc" 2>&1 | grep -E "error"; true
}
trap restore_default_build EXIT
