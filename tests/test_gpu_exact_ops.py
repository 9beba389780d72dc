"""The kernels replace the specification's correctly rounded 1/x, sqrt(x) and a/n (10-14 instruction sequences) by the
hardware approximation plus ONE fused correction step (hsk_dev.h).  That is only legitimate if the bits are the same for
every input the kernels can meet -- which is checked here exhaustively, on the GPU under test: all 2^32 binary32 values
for 1/x and sqrt, all (a, n) pairs with |a| < 512, n = 1..129 for the running-mean division."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_exact_shortcuts_hold_for_every_input(hsk):
    counts = (C.c_uint64 * 8)()
    rc = hsk._lib.load().hsk_selftest_exact_ops(0, counts)
    assert rc == 0
    c = [int(v) for v in counts]
    # domains: normal x with a normal reciprocal; x >= 2^-102; 2^-100 <= |a| < 512 or a = +0, times 129 divisors
    assert c[0] > 4.2e9 and c[3] > 1.9e9 and c[6] == 129 * (2 * (135 - 27 + 1) * 2 ** 23 + 1)
    assert c[1] == 0, f"hsk_rcp_exact differs from 1/x on {c[1]} values"
    assert c[4] == 0, f"hsk_sqrt_exact differs from sqrtf on {c[4]} values"
    assert c[7] == 0, f"hsk_div_small_exact differs from a/n on {c[7]} pairs"
    # the comparison bites: the bare instructions are wrong on about a tenth of the values
    assert c[2] > 1e8 and c[5] > 1e8
