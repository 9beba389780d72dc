# sourced by every script here that rebuilds the library with other than the default flags: whatever happens,
# the default build is back in place when the script exits (bench.py and the tests refuse a "+exp" library)
restore_default_build() { make -s -C "${GRAFT_REPO_ROOT:-.}/housescan_amd/csrc" 2>&1 | grep -E "error"; true; }
trap restore_default_build EXIT
