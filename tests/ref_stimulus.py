"""The stimulus of the reference's own component test, restated (test infrastructure).

``gen_psk`` follows reference tests/test_psk_soft.py:98-117: ideal constellation points, a
rectangular pulse and 1e-4 uniform noise on the real part only, driven by Python's Mersenne
Twister seeded with 100 (``random.seed(100)``, tests/test_psk_soft.py:41).  The reference test is
Python 2; its ``random.choice(seq)`` is ``seq[int(random() * len(seq))]`` there, which is what
``_py2_choice`` does so that the symbol stream is the one Python 2 draws."""
import math
import random as _random

import numpy as np


def _py2_choice(rng, seq):
    return seq[int(rng.random() * len(seq))]


def gen_psk(num_symbols, samp_per_baud=8, num_syms=4, differential=False, rng=None):
    """Return (interleaved float32 I/Q, list of transmitted complex symbols)."""
    if rng is None:
        rng = _random.Random(100)
    syms = list(range(num_syms))
    phase = [2 * math.pi * x / num_syms for x in syms]
    cx = [complex(math.cos(x), math.sin(x)) for x in phase]
    out = np.empty(2 * num_symbols * samp_per_baud, dtype=np.float64)
    input_symbols = []
    last = 1
    pos = 0
    for _ in range(num_symbols):
        x = _py2_choice(rng, syms)
        x_cx = cx[x]
        input_symbols.append(x_cx)
        if differential:
            val = x_cx * last
            last = val
        else:
            val = x_cx
        for _ in range(samp_per_baud):
            v = val + 0.0001 * rng.random()
            out[pos] = v.real
            out[pos + 1] = v.imag
            pos += 2
    return out.astype(np.float32), input_symbols
